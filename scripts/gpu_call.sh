#!/bin/bash
O=gpurun_out/r02filt
mkdir -p $O
cd $GRAFT_REPO_ROOT
( time timeout 900 python -m pytest tests/test_bm25_ranges_gpu.py tests/test_fulltext_gpu.py tests/test_random_gpu.py tests/test_sharded_fulltext_gpu.py tests/test_token_score_gpu.py -m gpu -x -q -p no:cacheprovider ) > $O/pytest.log 2>&1; grep "passed\|failed\|Error" $O/pytest.log | head -3
timeout 600 python scripts/bench_bm25_filtered.py > $O/bm25_filtered.log 2>&1; tail -2 $O/bm25_filtered.log
