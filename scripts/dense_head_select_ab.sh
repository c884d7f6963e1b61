# the dense heads of the fp16 scans through pairs_reduce_wide_kernel (default) against round 4's form (ORAMA_SELECT_WIDE=3: wide only for a few long lists)
# (round 6: sweep / A-B variables are read by the COMPARISON flavour only — liborama_hip_cmp.so, built and loaded with this set)
export ORAMA_COMPARISON_KERNELS=1
for M in 1 3 1 3; do
  echo "== ORAMA_SELECT_WIDE=$M"
  ORAMA_SELECT_WIDE=$M python bench.py --workload c3 --steps 40 --warmup 3 --no-cpu-baseline --configs none --no-pmc 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('c3', round(d['value'],1), 'ms_per_step', round(d['ms_per_step'],4), 'p50', round(d['latency_ms_p50'],4), d['step_breakdown_us']['select'])"
  ORAMA_SELECT_WIDE=$M python bench.py --workload c5 --rows 10000000 --steps 30 --warmup 3 --no-cpu-baseline --configs none --no-pmc 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('c5', round(d['value'],1), 'ms_per_step', round(d['ms_per_step'],4), 'p50', round(d['latency_ms_p50'],4), d['step_breakdown_us']['select'])"
done
