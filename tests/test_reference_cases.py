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
    assert len(CASES) >= 7
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
        td, ts = orc.top_n(docs, scores, p.get("limit", 10))
        return [(int(d), float(s)) for d, s in zip(td, ts)], len(docs), ids

    refcases.check_case(case, search)
