#!/bin/bash
set -x
mkdir -p gpurun_out/r02c2
cd $GRAFT_REPO_ROOT
timeout 600 python scripts/k2d_probe.py > gpurun_out/r02c2/k2d_probe.log 2>&1
cat gpurun_out/r02c2/k2d_probe.log
for m in 4 5 7; do
  ( ORAMA_F16_WIDE=$m timeout 300 python -m pytest tests/test_vector_f16_gpu.py -m gpu -x -q ) > gpurun_out/r02c2/pytest_f16_mode$m.log 2>&1
  tail -3 gpurun_out/r02c2/pytest_f16_mode$m.log
done
