# the scan's completion event on its dispatch (default) against a record packet behind it (ORAMA_SCAN_DONE_EVENT=record)
# (round 6: sweep / A-B variables are read by the COMPARISON flavour only — liborama_hip_cmp.so, built and loaded with this set)
export ORAMA_COMPARISON_KERNELS=1
for M in dispatch record dispatch record; do
  echo "== ORAMA_SCAN_DONE_EVENT=$M"
  for W in c2 ns; do
    ORAMA_SCAN_DONE_EVENT=$M python bench.py --workload $W --steps $([ $W = c2 ] && echo 400 || echo 40) --warmup 10 --no-cpu-baseline --no-two-stage --configs none --no-pmc 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$W', round(d['value'],2), 'ms_per_step', round(d['ms_per_step'],4), 'p50', round(d['latency_ms_p50'],4), 'median scan', d['roofline'].get('median_scan_ms_per_step'))"
  done
done
