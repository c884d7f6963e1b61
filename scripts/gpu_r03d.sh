#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03d
mkdir -p $O
timeout 600 python -m pytest tests/test_vector_f16_gpu.py -q -m gpu -p no:cacheprovider -x -k "wide or l2" 2>&1 | tail -5 | tee $O/pytest_f16.log
MODES=2,4 NQ=256,200 timeout 300 python scripts/k2d_probe.py 2>&1 | tee $O/k2q_probe.log
for D in 9 2 1 32; do
  DBG=$D MODES=4 NQ=256 timeout 200 python scripts/k2d_probe.py 2>&1 | grep mode | sed "s/^/DBG=$D /" | tee -a $O/k2q_ablation.log
done
