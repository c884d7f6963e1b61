#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_vector_f16_gpu.py tests/test_two_stage_gpu.py -q -m gpu -p no:cacheprovider -x 2>&1 | grep -E "passed|failed|error" | tail -1
