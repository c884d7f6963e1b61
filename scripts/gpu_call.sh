#!/bin/bash
O=gpurun_out/r02rand
mkdir -p $O
cd $GRAFT_REPO_ROOT
( time timeout 1200 python -m pytest tests/test_random_gpu.py -m gpu -x -q -p no:cacheprovider ) > $O/pytest.log 2>&1; tail -15 $O/pytest.log
