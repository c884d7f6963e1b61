#!/bin/bash
set -x
mkdir -p gpurun_out/r02c7
cd $GRAFT_REPO_ROOT
O=gpurun_out/r02c7
( time timeout 900 python -m pytest tests/test_shard_group_gpu.py tests/test_sharded_fulltext_gpu.py tests/test_vector_f16_gpu.py -m gpu -x -q ) > $O/pytest_new.log 2>&1; tail -15 $O/pytest_new.log
NQ=128,256 MODES=1,2,3 ROWS=10000000 timeout 300 python scripts/k2d_probe.py > $O/k2d_probe_nq.log 2>&1; cat $O/k2d_probe_nq.log
