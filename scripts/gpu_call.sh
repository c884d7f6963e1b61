#!/bin/bash
O=gpurun_out/r02pb
mkdir -p $O
cd $GRAFT_REPO_ROOT
( time timeout 600 python -m pytest tests/test_batcher_gpu.py -m gpu -x -q -p no:cacheprovider ) > $O/pytest.log 2>&1; tail -4 $O/pytest.log
timeout 300 scripts/native/bench_serving vec 10000000 100 8,64,256,512 > $O/serving_vec.log 2>&1; cat $O/serving_vec.log
