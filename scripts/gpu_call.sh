#!/bin/bash
cd $GRAFT_REPO_ROOT
( time timeout 300 python -m pytest tests/test_stress_gpu.py -m gpu -x -q -p no:cacheprovider ) 2>&1 | tail -25 | grep -v "^  File"
