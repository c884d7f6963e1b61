#!/bin/bash
O=gpurun_out/r02ts3
mkdir -p $O
cd $GRAFT_REPO_ROOT
( time timeout 900 python -m pytest tests/test_two_stage_gpu.py -m gpu -x -q -p no:cacheprovider ) > $O/pytest.log 2>&1; tail -12 $O/pytest.log
