"""The oracle against the cases the reference's own integration tests hold for the scoring path (tests/golden/
reference_cases.json — documents, queries and asserted relations of src/tests/boost_integration.rs, fulltext_search.rs,
omc_test.rs as data).  CPU only: the restatement must satisfy what the reference asserts before the HIP path is compared
with it (tests/test_reference_cases_gpu.py runs the same cases through the kernels)."""
import pytest

import refcases
import util
from oracle import oracle as orc

CASES = util.load_json("reference_cases.json")["cases"]


def test_every_case_cites_the_reference_and_names_what_it_constrains():
    from oramacore_amd.token_score import DEFAULT_EXACT_MATCH_BOOST

    # the checker's default IS the mirror's default, and it is a factor > 1 (boost_integration.rs:449-491 needs that)
    assert refcases.DEFAULT_EXACT_MATCH_BOOST == DEFAULT_EXACT_MATCH_BOOST > 1.0
    assert not any("host_params" in c for c in CASES)
    assert len(CASES) >= 19
    for c in CASES:
        assert c["reference"].startswith("src/tests/") and ":" in c["reference"], c["name"]
        assert c["constrains"] and c["fields"] and c["searches"], c["name"]
        assert ("documents" in c) != ("generate_documents" in c), c["name"]


@pytest.mark.parametrize("case", CASES, ids=lambda c: c["name"])
def test_oracle_satisfies_the_reference_case(case):
    idx = refcases.HostIndex()
    ids = refcases.fill(idx, case)

    def search(spec, exact_match_boost):
        p = spec["params"]
        docs, scores = refcases.oracle_search(idx, p, case["fields"], exact_match_boost)
        off = p.get("offset", 0)
        td, ts = orc.top_n(docs, scores, p.get("limit", 10) + off)  # search.rs:481-500: top (limit + offset), skip(offset)
        return [(int(d), float(s)) for d, s in zip(td[off:], ts[off:])], len(docs), ids

    refcases.check_case(case, search)


# ---------------------------------------------------------------- facets (src/tests/facets.rs as data)
FACET_CASES = util.load_json("reference_facet_cases.json")["cases"]


def test_every_facet_case_cites_the_reference():
    assert len(FACET_CASES) >= 8
    for c in FACET_CASES:
        assert c["reference"].startswith("src/tests/facets.rs:") and c["constrains"] and c["facets"] and c["expect"], c["name"]
        for spec in c["filter_fields"].values():
            assert spec["kind"] in ("bool", "number", "string")


@pytest.mark.parametrize("case", FACET_CASES, ids=lambda c: c["name"])
def test_oracle_counts_the_facets_the_reference_asserts(case):
    """The restatement of facet.rs (orc_facet_count_buckets / orc_facet_count_ranges over the restatement's score map) against
    the facet results the reference's own tests assert — number ranges inclusive at both ends, bool / string values, facets of
    the matched documents only, the `where` filter left out of the facet map, values added over the indexes of a collection."""
    import numpy as np

    total = {}
    defs = refcases.facet_definitions(case)
    for _ in range(case.get("indexes", 1)):
        idx = refcases.HostIndex()
        ids = refcases.fill(idx, case)
        refcases.fill_filters(idx, case, ids)
        docs, scores = refcases.oracle_search(idx, {"term": case["search"]["term"]}, case["fields"])
        keep = refcases.where_docs(idx, case["search"].get("where"))
        if "expect_hits_count" in case:  # the hits keep the filter ...
            assert sum(1 for d in docs if keep is None or int(d) in keep) == case["expect_hits_count"]
        # ... the facets do not (search.rs:347-396)
        refcases.add_facets(total, refcases.oracle_facets(idx, np.asarray(docs, dtype=np.uint64), defs))
    for _ in range(case.get("empty_indexes", 0)):
        refcases.add_facets(total, refcases.oracle_facets(refcases.HostIndex(), np.zeros(0, dtype=np.uint64), defs))
    refcases.check_facets(case, total)


# ---------------------------------------------------------------- groups (src/tests/groupby.rs as data)
GROUP_CASES = util.load_json("reference_group_cases.json")["cases"]


@pytest.mark.parametrize("case", GROUP_CASES, ids=lambda c: c["name"])
def test_oracle_forms_the_groups_the_reference_asserts(case):
    """The restatement of group.rs / sort.rs's score-ordered groups (orc_group_top over the restatement's score map) against
    what the reference's own tests assert about them."""
    assert case["reference"].startswith("src/tests/groupby.rs:") and case["constrains"]
    idx = refcases.HostIndex()
    ids = refcases.fill(idx, case)
    refcases.fill_filters(idx, case, ids)
    for spec in case["searches"]:
        docs, scores = refcases.oracle_search(idx, {"term": spec["term"]}, case["fields"])
        g = spec["group_by"]
        mr = 1 if g["max_results"] is None else g["max_results"]  # default_group_by_max_results, types.rs:1473-1475
        groups = refcases.oracle_groups(idx, docs, scores, g["properties"], mr)
        td, ts = orc.top_n(docs, scores, 10)
        refcases.check_groups(spec, groups, ids, hits=list(zip(td.tolist(), ts.tolist())), count=len(docs))
