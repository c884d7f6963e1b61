for M in 0 1 0 1; do
# (round 6: sweep / A-B variables are read by the COMPARISON flavour only — liborama_hip_cmp.so, built and loaded with this set)
export ORAMA_COMPARISON_KERNELS=1
  echo "== ORAMA_HYBRID_DEVICE_TAIL=$M"
  ORAMA_HYBRID_DEVICE_TAIL=$M python bench.py --steps 10 --warmup 3 --no-cpu-baseline --configs c4 --no-pmc 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
c=d['configs']['c4']
print('hybrid', round(c['value'],2), 'p50', round(c['latency_ms_p50'],4), 'p95', round(c['latency_ms_p95'],4), 'shadow', round(c['shadow_store']['value'],2), 'p50', round(c['shadow_store']['latency_ms_p50'],4))"
done
