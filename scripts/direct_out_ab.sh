# the answers written to the pinned host block by the selection's final launch (default) against a staging launch behind it: C4 (K3r batch, single calls, hybrid) and C2 (lone vector query)
# (round 6: sweep / A-B variables are read by the COMPARISON flavour only — liborama_hip_cmp.so, built and loaded with this set)
export ORAMA_COMPARISON_KERNELS=1
for M in 1 0 1 0; do
  echo "== ORAMA_DIRECT_OUT=$M"
  ORAMA_DIRECT_OUT=$M python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-two-stage --configs c4 --no-pmc 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d['configs']['c4']; b=c['bm25_only']
print('bm25 batch', round(b['value']), 'best', round(b['runs']['best']), 'single', round(b['single_query_calls']['value']), 'wrapper', round(b['through_python_wrapper']['value']), '| hybrid', round(c['value'],2), 'p50', round(c['latency_ms_p50'],4), 'p95', round(c['latency_ms_p95'],4))"
done
for M in 1 0 1 0; do
  echo "== ORAMA_DIRECT_OUT=$M (c2)"
  ORAMA_DIRECT_OUT=$M python bench.py --workload c2 --steps 200 --warmup 10 --no-cpu-baseline --no-two-stage --configs none --no-pmc 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('c2 host-API p50', round(d['latency_ms_p50_host_api'],4), 'session p50', round(d['latency_ms_p50'],4), 'qps', round(d['value'],1))"
done
