#!/bin/bash
set -x
mkdir -p gpurun_out/r02c9
cd $GRAFT_REPO_ROOT
( time timeout 900 python -m pytest tests/test_facets_gpu.py tests/test_fulltext_gpu.py tests/test_token_score_gpu.py -m gpu -x -q ) > gpurun_out/r02c9/pytest.log 2>&1; tail -25 gpurun_out/r02c9/pytest.log
