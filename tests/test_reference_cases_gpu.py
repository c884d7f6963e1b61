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
            mode = FulltextMode(p["term"], threshold=p.get("threshold"), exact=p.get("exact", False))
            params = TokenScoreParams(mode=mode, limit=p.get("limit", 10),
                                      properties=None if "properties" not in p else [fields[n] for n in p["properties"]],
                                      boost={fields[n]: float(v) for n, v in p.get("boost", {}).items()})
            hits, count = tsc.execute(params)
            od, os_ = refcases.oracle_search(idx, p, case["fields"], exact_match_boost)
            td, ts = orc.top_n(od, os_, params.limit)
            assert count == len(od), (spec, count, len(od))
            assert [h[0] for h in hits] == td.tolist(), spec
            assert np.array_equal(np.array([h[1] for h in hits], dtype=F).view(np.uint32), ts.view(np.uint32)), spec
            return hits, count, ids

        refcases.check_case(case, search)
