import sys
sys.path.insert(0, '.')
import oramacore_amd as oa
ctx = oa.Context(0)
for n in (1_000_000, 5_000_000, 10_000_016):
    try:
        st = oa.EmbeddingFieldStorage(ctx, dimensions=768, reserve_rows=n)
        print(n, "ok", st.info()["hbm_bytes"])
        st.close()
    except Exception as e:
        print(n, "FAILED", e)
