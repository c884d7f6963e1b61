#!/bin/bash
set -x
mkdir -p gpurun_out/r02c10
cd $GRAFT_REPO_ROOT
( time timeout 1200 python -m pytest tests -m gpu -x -q ) > gpurun_out/r02c10/pytest_gpu.log 2>&1; tail -15 gpurun_out/r02c10/pytest_gpu.log
