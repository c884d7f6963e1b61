#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/final
timeout 1500 python -m pytest tests/ -q -m gpu -p no:cacheprovider -x 2>&1 | grep -E "passed|failed|error" | tail -3 | tee gpurun_out/final/pytest_gpu_summary.log
