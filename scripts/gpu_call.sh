#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/meas
for h in 131072 65536 32768 16384; do
  echo "== head $h"; ORAMA_F16_HEAD_ROWS=$h timeout 300 python bench.py --workload c3 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('c3', round(d['value']), round(d['ms_per_step'],3), round(d['roofline']['frac'],3), round(d['roofline']['topk_select_ms_per_step'],3))"
  ORAMA_F16_HEAD_ROWS=$h QB=64,256 timeout 300 python scripts/two_stage_breakdown.py 2>&1 | tail -2
done | tee gpurun_out/meas/head_rows.log
