#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03b
mkdir -p $O
timeout 1200 python -m pytest tests/ -q -m gpu -p no:cacheprovider 2>&1 | tail -120 > $O/pytest_gpu.log
tail -60 $O/pytest_gpu.log
