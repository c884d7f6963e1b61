#!/bin/bash
O=gpurun_out/r02ts2
mkdir -p $O
cd $GRAFT_REPO_ROOT
( time timeout 900 python -m pytest tests/test_two_stage_gpu.py tests/test_fulltext_gpu.py tests/test_token_score_gpu.py tests/test_facets_gpu.py tests/test_sharded_fulltext_gpu.py tests/test_shard_group_gpu.py -m gpu -x -q -p no:cacheprovider ) > $O/pytest.log 2>&1; tail -4 $O/pytest.log
timeout 400 scripts/native/bench_serving hybrid 10000000 60 1,32,128 shadow > $O/serving_hybrid_shadow.log 2>&1; cat $O/serving_hybrid_shadow.log
