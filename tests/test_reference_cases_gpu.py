"""The reference's own integration-test cases for the scoring path, as DATA (tests/golden/reference_cases.json: their
documents, their queries, the relations they assert — src/tests/boost_integration.rs, fulltext_search.rs, omc_test.rs),
driven through the HIP path: TokenScoreContext -> resident postings -> K3 / K3r -> K4.  VERDICT r04 next #7: these are
among the few reference-held constraints on the declared ntf / boost / exact-match assumptions of DESIGN.md §3.

Every search is additionally compared with the oracle bit for bit, so a case cannot pass on the relations alone."""
import numpy as np
import pytest

import oramacore_amd as oa
import refcases
import util
from oracle import oracle as orc
from oramacore_amd.token_score import FulltextMode, Index, TokenScoreContext, TokenScoreParams

pytestmark = pytest.mark.gpu
F = np.float32
CASES = util.load_json("reference_cases.json")["cases"]


@pytest.mark.parametrize("case", CASES, ids=lambda c: c["name"])
def test_reference_case_through_the_hip_path(case):
    with oa.Context(0) as ctx:
        idx = Index(ctx)
        ids = refcases.fill(idx, case)
        idx.commit()
        fields = {name: fi for fi, name in enumerate(case["fields"])}

        def search(spec, exact_match_boost):
            p = spec["params"]
            tsc = TokenScoreContext(idx, exact_match_boost=exact_match_boost)
            mode = FulltextMode(p["term"], threshold=p.get("threshold"), exact=p.get("exact", False), tolerance=p.get("tolerance"))
            params = TokenScoreParams(mode=mode, limit=p.get("limit", 10), offset=p.get("offset", 0),
                                      properties=None if "properties" not in p else [fields[n] for n in p["properties"]],
                                      boost={fields[n]: float(v) for n, v in p.get("boost", {}).items()})
            hits, count = tsc.execute(params)
            od, os_ = refcases.oracle_search(idx, p, case["fields"], exact_match_boost)
            td, ts = orc.top_n(od, os_, params.limit + params.offset)
            td, ts = td[params.offset:], ts[params.offset:]
            assert count == len(od), (spec, count, len(od))
            assert [h[0] for h in hits] == td.tolist(), spec
            assert np.array_equal(np.array([h[1] for h in hits], dtype=F).view(np.uint32), ts.view(np.uint32)), spec
            return hits, count, ids

        refcases.check_case(case, search)


# ---------------------------------------------------------------- facets (src/tests/facets.rs as data)
FACET_CASES = util.load_json("reference_facet_cases.json")["cases"]


@pytest.mark.parametrize("case", FACET_CASES, ids=lambda c: c["name"])
def test_reference_facet_case_through_the_hip_path(case):
    """The reference's facet tests through the resident score map (K3r in score-map mode -> orama_facet_count /
    orama_facet_count_ranges, K7): the facet results the reference asserts, and the oracle's restatement of every index's
    facets bit for bit (counts are integers: equal or wrong)."""
    from oramacore_amd.token_score import facets_and_groups

    defs = refcases.facet_definitions(case)
    with oa.Context(0) as ctx:
        total, total_oracle = {}, {}
        for _ in range(case.get("indexes", 1)):
            idx = Index(ctx)
            ids = refcases.fill(idx, case)
            refcases.fill_filters(idx, case, ids)
            idx.commit()
            keep = refcases.where_docs(idx, case["search"].get("where"))
            allow = None if keep is None else oa.AllowBitmap(max(ids.values()) + 1, np.asarray(sorted(keep), dtype=np.uint64))
            tsc = TokenScoreContext(idx)
            params = TokenScoreParams(mode=FulltextMode(case["search"]["term"]), limit=10, filtered_doc_ids=allow)
            hits, count, facets, _ = facets_and_groups(tsc, params, facets=defs, has_where_filter=keep is not None, not_deleted=None)
            if "expect_hits_count" in case:
                assert count == case["expect_hits_count"] and all(h[0] in keep for h in hits)
            od, _ = refcases.oracle_search(idx, {"term": case["search"]["term"]}, case["fields"])
            mine = refcases.oracle_facets(idx, np.asarray(od, dtype=np.uint64), defs)
            assert facets == mine, (case["name"], facets, mine)
            refcases.add_facets(total, facets)
        # (an index without documents holds no filter field: facet.rs:159-163 skips it — nothing to launch)
        refcases.check_facets(case, total)


# ---------------------------------------------------------------- groups (src/tests/groupby.rs as data)
GROUP_CASES = util.load_json("reference_group_cases.json")["cases"]


@pytest.mark.parametrize("case", GROUP_CASES, ids=lambda c: c["name"])
def test_reference_group_case_through_the_hip_path(case):
    """The reference's score-ordered group tests through the resident score map (orama_group_top, K7): what the reference
    asserts, and every group equal to the oracle's — documents and score bits."""
    from oramacore_amd.token_score import facets_and_groups

    with oa.Context(0) as ctx:
        idx = Index(ctx)
        ids = refcases.fill(idx, case)
        refcases.fill_filters(idx, case, ids)
        idx.commit()
        tsc = TokenScoreContext(idx)
        for spec in case["searches"]:
            g = spec["group_by"]
            mr = 1 if g["max_results"] is None else g["max_results"]  # default_group_by_max_results, types.rs:1473-1475
            params = TokenScoreParams(mode=FulltextMode(spec["term"]), limit=10)
            hits, count, _, groups = facets_and_groups(tsc, params, group_by=(g["properties"], mr))
            refcases.check_groups(spec, groups, ids, hits=hits, count=count)
            od, os_ = refcases.oracle_search(idx, {"term": spec["term"]}, case["fields"])
            mine = refcases.oracle_groups(idx, od, os_, g["properties"], mr)
            assert set(groups) == set(mine)
            for key, got in groups.items():
                assert [d for d, _ in got] == [d for d, _ in mine[key]], (case["name"], key)
                assert np.array_equal(np.array([x for _, x in got], dtype=F).view(np.uint32),
                                      np.array([x for _, x in mine[key]], dtype=F).view(np.uint32)), (case["name"], key)
