import os, sys
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np
import oramacore_amd as oa
import util
from oracle import oracle as orc
ctx = oa.Context(0)
n, d, nq, k = 207, 129, 10, 7
bad = 0
for it in range(300):
    seed = 207611474 + it
    rng = np.random.default_rng(seed)
    corpus = util.gaussian_rows(n, d, seed=seed % 100003)
    doc_ids = np.arange(n, dtype=np.uint64) * 2 + 3
    st = oa.EmbeddingFieldStorage(ctx, dimensions=d)
    st.insert_rows(doc_ids, corpus)
    dead = np.zeros(n, dtype=bool)
    for r in rng.choice(n, size=3, replace=False):
        st.delete(int(doc_ids[r])); dead[r] = True
    queries = util.gaussian_rows(nq, d, seed=(seed + 1) % 100003)
    ids, dist, cnt = st.storage_search(queries, k)
    info = st.info()
    for qi in range(nq):
        full = orc.distances(corpus, queries[qi]).astype(np.float64); full[dead] = np.nan
        exp = np.argsort(np.where(np.isnan(full), np.inf, full), kind="stable")[:k]
        got = ((ids[qi, :cnt[qi]] - 3) // 2).astype(np.int64)
        if sorted(got.tolist()) != sorted(exp.tolist()):
            bad += 1
            print("iter", it, "q", qi, "pending", info["pending_ops"], "got", got.tolist(), "exp", exp.tolist(), "dead rows", np.nonzero(dead)[0].tolist(), flush=True)
            # which rows does a k=n search return?
            ids2, dist2, cnt2 = st.storage_search(queries[qi], n)
            present = set(((ids2[0, :cnt2[0]] - 3) // 2).tolist())
            print("   rows missing from a full-k search:", sorted(set(range(n)) - present)[:40], "count", cnt2[0])
            break
    st.close()
print("bad iterations:", bad, "NO_VMM" if os.environ.get("ORAMA_NO_VMM") else "VMM")
