#!/bin/bash
set -x
mkdir -p gpurun_out/r02c16
cd $GRAFT_REPO_ROOT
( timeout 300 python -m pytest tests/test_fulltext_gpu.py tests/test_batcher_gpu.py tests/test_token_score_gpu.py -m gpu -x -q -p no:cacheprovider ) > gpurun_out/r02c16/pytest.log 2>&1; tail -4 gpurun_out/r02c16/pytest.log
timeout 150 python scripts/bench_bm25_threads.py > gpurun_out/r02c16/bm25_threads.log 2>&1; cat gpurun_out/r02c16/bm25_threads.log
