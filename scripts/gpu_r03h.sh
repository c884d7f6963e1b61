#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03h
mkdir -p $O
timeout 600 python -m pytest tests/test_vector_f16_gpu.py -q -m gpu -p no:cacheprovider -x -k "wide or l2" 2>&1 | tail -8 | tee $O/pytest_f16.log
MODES=2,4,5 NQ=256,200 timeout 300 python scripts/k2d_probe.py 2>&1 | tee $O/k2h_probe.log
for D in 9 1 42 34 40 32; do
  DBG=$D MODES=4 NQ=256 timeout 200 python scripts/k2d_probe.py 2>&1 | grep mode | sed "s/^/DBG=$D /" | tee -a $O/k2h_ablation.log
done
