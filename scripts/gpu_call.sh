#!/bin/bash
O=gpurun_out/r02filt
mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 600 python scripts/bench_filtered.py > $O/filtered.log 2>&1; tail -9 $O/filtered.log
( time timeout 900 python -m pytest tests/test_vector_f16_gpu.py tests/test_two_stage_gpu.py tests/test_random_gpu.py tests/test_batcher_gpu.py tests/test_facets_gpu.py tests/test_shard_group_gpu.py -m gpu -x -q -p no:cacheprovider ) > $O/pytest.log 2>&1; grep "passed\|failed\|Error" $O/pytest.log | head -3
