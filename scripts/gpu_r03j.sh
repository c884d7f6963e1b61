#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03j
mkdir -p $O
timeout 900 python -m pytest tests/test_vector_f16_gpu.py tests/test_two_stage_gpu.py -q -m gpu -p no:cacheprovider -x 2>&1 | tail -5 | tee $O/pytest_f16.log
MODES=2,4,5 NQ=256 timeout 300 python scripts/k2d_probe.py 2>&1 | tee $O/probe.log
timeout 600 python -m pytest tests/test_full_size_gpu.py -q -m gpu -p no:cacheprovider -x -k "c5" 2>&1 | tail -3 | tee $O/pytest_c5.log
