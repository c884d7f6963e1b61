export ORAMA_K3R_STATS=0
# (round 6: sweep / A-B variables are read by the COMPARISON flavour only — liborama_hip_cmp.so, built and loaded with this set)
export ORAMA_COMPARISON_KERNELS=1
mkdir -p gpurun_out
(
for M in kernel dma; do
  echo "== ORAMA_STAGE_COPY=$M: BM25 batch before / after one hybrid call (scripts/k3r_bench_gap_probe2.py hy:100:10)"
  ORAMA_STAGE_COPY=$M python scripts/k3r_bench_gap_probe2.py hy:100:10 2>&1 | grep -E "queries/s"
  echo "== ORAMA_STAGE_COPY=$M: C2 (1M x 384 fp32, lone query)"
  ORAMA_STAGE_COPY=$M python bench.py --workload c2 --steps 200 --warmup 10 --no-cpu-baseline --no-two-stage --configs none --no-pmc 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print({k:d.get(k) for k in ('value','ms_per_step','latency_ms_p50','latency_ms_p95','latency_ms_p50_host_api')}, d['roofline']['frac'], d.get('step_breakdown_us'))"
  echo "== ORAMA_STAGE_COPY=$M: C4"
  ORAMA_STAGE_COPY=$M python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-two-stage --configs c4 --no-pmc 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
c=d['configs']['c4']; b=c['bm25_only']
print('hybrid', c['value'], 'p50', c['latency_ms_p50'], 'shadow', c['shadow_store']['value'], 'p50', c['shadow_store']['latency_ms_p50'])
print('bm25 batch', b['value'], b.get('runs'), 'wrapper', b['through_python_wrapper']['value'], 'single', b['single_query_calls']['value'])"
done
) > gpurun_out/r05_stage_copy_ab.log 2>&1
cat gpurun_out/r05_stage_copy_ab.log
