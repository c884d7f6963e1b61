#!/bin/bash
cd $GRAFT_REPO_ROOT
( timeout 900 python -m pytest tests/test_vector_gpu.py tests/test_batcher_gpu.py tests/test_token_score_gpu.py tests/test_fulltext_gpu.py tests/test_two_stage_gpu.py tests/test_post_append_gpu.py -m gpu -x -q -p no:cacheprovider ) 2>&1 | grep "passed\|failed\|rror" | head -3
