#!/bin/bash
# round 3, call C: K2q (queries stationary in registers) — parity, then timing vs K2d and the ablation builds
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03c
mkdir -p $O
timeout 900 python -m pytest tests/test_vector_f16_gpu.py tests/test_two_stage_gpu.py tests/test_batcher_gpu.py -q -m gpu -p no:cacheprovider -x 2>&1 | tail -15 > $O/pytest_f16.log
tail -6 $O/pytest_f16.log
echo "== probe: K2d (2) vs K2q (4), 256 / 128 queries"
MODES=2,4 NQ=256,128,200 timeout 300 python scripts/k2d_probe.py 2>&1 | tee $O/k2q_probe.log
for D in 9 2 1 32; do
  echo "== ablation DBG=$D (K2q)"
  DBG=$D MODES=4 NQ=256 timeout 200 python scripts/k2d_probe.py 2>&1 | grep mode | tee -a $O/k2q_ablation.log
done
echo "== full-size C5 shard test"
timeout 600 python -m pytest tests/test_full_size_gpu.py -q -m gpu -p no:cacheprovider -x -k "c5 or c3 or two_stage" 2>&1 | tail -5 | tee $O/pytest_full_c5.log
