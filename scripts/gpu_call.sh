#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/meas
for w in 2 4 2 4; do
  echo "== ORAMA_F16_WIDE=$w"; ORAMA_F16_WIDE=$w QB=256 timeout 300 python scripts/two_stage_breakdown.py 2>&1 | tail -1
  ORAMA_F16_WIDE=$w timeout 200 python bench.py --workload c5 --rows 10000000 --steps 20 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('c5 shard', round(d['value']), round(d['ms_per_step'],3), round(d['roofline']['frac'],3))"
done | tee gpurun_out/meas/pcb4.log
