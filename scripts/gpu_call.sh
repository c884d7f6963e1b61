#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/meas
for g in default 0 1; do
  if [ $g = default ]; then unset ORAMA_F16_CHUNK_GROW; else export ORAMA_F16_CHUNK_GROW=$g; fi
  echo "== growth $g"; QB=100,128,200 timeout 300 python scripts/two_stage_breakdown.py 2>&1 | tail -3
done | tee gpurun_out/meas/two_stage_128.log
