#!/bin/bash
# K2 (<= 64 queries) after the zero-scratch rewrite: parity suites, the C3 shape, the batch-width table on the C5 shape
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03k
mkdir -p $O
timeout 1200 python -m pytest tests/test_vector_f16_gpu.py tests/test_two_stage_gpu.py tests/test_batcher_gpu.py tests/test_vector_gpu.py -q -m gpu -p no:cacheprovider -x 2>&1 | tail -5 | tee $O/pytest_f16.log
ROWS=1000000 MODES=4 NQ=1,8,16,32,64 timeout 300 python scripts/k2d_probe.py 2>&1 | tee $O/probe_c3_shape.log
MODES=4 NQ=1,4,8,9,16,32,64,128,256 timeout 600 python scripts/k2d_probe.py 2>&1 | tee $O/probe_c5_shape.log
timeout 600 python bench.py --no-pmc 2>&1 | tail -1 | tee $O/bench.json
