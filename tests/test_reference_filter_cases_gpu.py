"""The reference's filter / multi-index / update tests as DATA (tests/golden/reference_filter_cases.json: src/tests/filter.rs,
multi_index.rs, update_docs.rs, replace_doc_on_insert.rs, bugs.rs, commit.rs) through the HIP path: the host-side filter
materialiser -> AllowBitmap -> resident postings -> K3 / K3r -> K4, several indexes through `search_on_indexes`.

Each case runs TWICE: mutations as the reference applies them between commits — inserts as delta posting lists
(orama_post_append), deletes through the NOT-deleted bitmap every later search carries (filter.rs:352-390) — and with a commit
(rebuild) after every mutation.  Both must satisfy what the reference asserts and equal the oracle's answer: ids in order, scores
bit for bit, count — which also makes them equal to each other (src/tests/commit.rs: the same answers before and after compact)."""
import pytest

import filtercases
import oramacore_amd as oa
import util
from oramacore_amd.token_score import FulltextMode, Index, TokenScoreContext, TokenScoreParams, search_on_indexes

pytestmark = pytest.mark.gpu
CASES = util.load_json("reference_filter_cases.json")["cases"]


@pytest.mark.parametrize("live", [True, False], ids=["delta-lists+bitmap", "commit-after-every-mutation"])
@pytest.mark.parametrize("case", CASES, ids=lambda c: c["name"])
def test_reference_filter_case_through_the_hip_path(case, live):
    with oa.Context(0) as ctx:
        def on_insert(idx, ids):
            if live and idx._post is not None:
                idx.append_documents(ids)
            else:
                idx.commit()

        def make(c):
            col = filtercases.Collection(c, lambda: Index(ctx), on_insert=on_insert, on_commit=lambda idx: idx.commit())
            if not live:  # deletes too are followed by a rebuild: wrap the model's apply
                plain = col.apply

                def apply(step):
                    plain(step)
                    for idx in col.indexes:
                        idx.commit()
                col.apply = apply
            return col

        def search(col, p):
            tscs = [TokenScoreContext(col.indexes[ii]) for ii in col.searched_indexes(p)]
            # (`properties` by NAME: every index resolves them against its own fields — token_score.rs:154-177)
            params = TokenScoreParams(mode=FulltextMode(p["term"]), limit=p.get("limit", 10), offset=p.get("offset", 0),
                                      properties=p.get("properties"), where_filter=p.get("where"))
            return search_on_indexes(tscs, params)

        col = filtercases.run_case(case, make, search)
        if live:  # nothing of the uncommitted state may be lost by the commit that follows
            last = [s for s in case["steps"] if s["op"] == "search" and "error" not in s["expect"]]
            before = [search(col, s["params"]) for s in last]
            for idx in col.indexes:
                idx.commit()
                assert not idx.uncommitted_deleted_documents
            assert [search(col, s["params"]) for s in last] == before
