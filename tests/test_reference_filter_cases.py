"""The reference's filter / multi-index / update tests as data (tests/golden/reference_filter_cases.json: src/tests/filter.rs,
multi_index.rs, update_docs.rs, replace_doc_on_insert.rs, bugs.rs, commit.rs, and — second half of round 6 — fulltext_search.rs
(`indexes`, property names per index, commit / reload), omc_test.rs (multipliers under commit / delete / replace), delete_doc.rs)
through the CPU restatement: the host-side filter
materialiser (oramacore_amd/filter.py <-> index/filter.rs:33-392) produces the document set, the oracle scores what passes.
tests/test_reference_filter_cases_gpu.py runs the same cases through the kernels."""
import pytest

import filtercases
import refcases
import util
from oramacore_amd import filter as flt

DOC = util.load_json("reference_filter_cases.json")
CASES = DOC["cases"]


def test_every_case_cites_the_reference():
    assert len(CASES) >= 27 and sum(1 for c in CASES for s in c["steps"] if s["op"] == "search") >= 75
    files = {c["reference"].split(":")[0] for c in CASES}
    assert {"src/tests/filter.rs", "src/tests/multi_index.rs", "src/tests/update_docs.rs", "src/tests/replace_doc_on_insert.rs",
            "src/tests/bugs.rs", "src/tests/commit.rs", "src/tests/fulltext_search.rs", "src/tests/omc_test.rs",
            "src/tests/delete_doc.rs"} <= files
    for c in CASES:
        assert c["constrains"] and c["indexes"] and c["steps"], c["name"]


@pytest.mark.parametrize("case", CASES, ids=lambda c: c["name"])
def test_oracle_and_filter_mirror_satisfy_the_reference_case(case):
    def search(col, p):  # the CPU run has one implementation: the oracle checks itself against the reference's numbers
        return col.oracle_search(p)

    filtercases.run_case(case, lambda c: filtercases.Collection(c, refcases.HostIndex), search)


def test_filter_tree_rules():
    """The rules of calculate_filter that no reference test spells out one by one (filter.rs:176-291)."""
    idx = refcases.HostIndex()
    idx.document_ids = {1, 2, 3, 4}
    idx.bool_fields["b"] = {1: True, 2: False, 3: True, 4: [True, False]}
    idx.number_fields["n"] = {1: 1, 2: 2.5, 3: [3, 30], 4: 4}
    idx.string_filter_fields["s"] = {1: "a", 2: "b", 3: ["a", "b"], 4: "x" * 26}
    fc = flt.FilterContext(idx)
    assert fc.allowed_set(None) is None and fc.allowed_set({}) is None           # no filter, no deletes: no predicate
    assert fc.allowed_set({"b": True}) == {1, 3, 4} and fc.allowed_set({"b": False}) == {2, 4}
    assert fc.allowed_set({"n": {"between": [2.5, 4]}}) == {2, 3, 4} and fc.allowed_set({"n": {"gt": 4}}) == {3}
    assert fc.allowed_set({"s": "a"}) == {1, 3} and fc.allowed_set({"s": "x" * 26}) == set()   # over 25 bytes: not a filter value
    assert fc.allowed_set({"b": {"eq": 1}}) == set() and fc.allowed_set({"n": True}) == set()  # wrong filter type: empty
    idx.date_fields["d"] = {1: flt.parse_date("2024-02-29T12:00:00Z"), 2: flt.parse_date("2024-03-01T00:00:00+01:00"), 3: [0, 10**13]}
    fc = flt.FilterContext(idx)
    assert fc.allowed_set({"d": {"gte": "2024-02-29T23:00:00Z"}}) == {2, 3} and fc.allowed_set({"d": {"eq": "2024-02-29T12:00:00Z"}}) == {1}
    assert fc.allowed_set({"d": {"gt": 5}}) == set() and fc.allowed_set({"n": {"gt": "2024-01-01T00:00:00Z"}}) == set()  # number <-> date: empty
    assert fc.allowed_set({"d": {"lt": "not a date"}}) == set() and flt.parse_date("machine learning") is None
    del idx.date_fields["d"]
    fc = flt.FilterContext(idx)
    assert fc.allowed_set({"nope": True}) == set()                                # unknown key: the whole level is empty
    assert fc.allowed_set({"or": []}) == set() and fc.allowed_set({"and": []}) == set()
    assert fc.allowed_set({"b": True, "n": {"lt": 4}}) == {1, 3}                  # entries of a level are AND-ed
    assert fc.allowed_set({"not": {"b": True}}) == {2} and fc.allowed_set({"not": {"nope": 1 == 1}}) == {1, 2, 3, 4}
    assert fc.allowed_set({"or": [{"s": "a"}, {"n": {"eq": 4}}], "not": {"b": False}}) == {1, 3}
    idx.uncommitted_deleted_documents = {3}
    idx.document_ids.discard(3)
    fc = flt.FilterContext(idx)
    assert fc.allowed_set(None) == {1, 2, 4} and fc.allowed_set({"b": True}) == {1, 4}   # NOT(deleted), alone or AND-ed
    bm = fc.execute_filter({"b": True})
    assert [d for d in range(bm.n_bits) if bm.contains(d)] == [1, 4]
    assert flt.all_keys({"a": 1, "and": [{"b": 2}, {"or": [{"c": 3}]}], "not": {"d": 4}}) == ["a", "b", "c", "d"]
    with pytest.raises(flt.FilterFieldNotFound):
        flt.check_filter_fields([idx], {"and": [{"b": True}, {"zzz": True}]})


def test_property_names_resolve_per_index_and_missing_facets_fail():
    """Host logic of the mirror that needs no device: calculate_string_properties (token_score.rs:154-177) and the collection-level
    facet check (search.rs:452-463)."""
    from types import SimpleNamespace

    from oramacore_amd.token_score import FacetFieldNotFound, StringFieldStorage, TokenScoreContext, check_facet_results

    idx = SimpleNamespace(string_fields={0: StringFieldStorage(), 2: StringFieldStorage()},
                          path_to_field_id_map={"title": (0, "string"), "body": (2, "string"), "price": (1, "number")})
    tsc = TokenScoreContext(idx)
    assert tsc.calculate_string_properties(None) == [0, 2]                       # Properties::None | Star
    assert tsc.calculate_string_properties(["body", "title"]) == [0, 2]          # canonical order: ascending FieldId
    assert tsc.calculate_string_properties(["body", "nope", "price"]) == [2]     # unknown names and non-string fields are skipped
    assert tsc.calculate_string_properties(["nope"]) == []                       # nothing left: the index searches nothing
    assert tsc.calculate_string_properties([2, 7, "title"]) == [0, 2]            # field ids still work (7: not a field of this index)
    check_facet_results({"a": "bool"}, {"a": {"count": 0, "values": {}}})
    with pytest.raises(FacetFieldNotFound) as e:
        check_facet_results({"a": "bool", "b": "string", "c": "string"}, {"b": {}})
    assert list(e.value.args[0]) == ["a", "c"]
