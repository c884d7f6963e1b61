#!/bin/bash
set -x
mkdir -p gpurun_out/r02c6
cd $GRAFT_REPO_ROOT
O=gpurun_out/r02c6
timeout 300 python scripts/k2d_trace.py > $O/k2d_trace.log 2>&1; cat $O/k2d_trace.log
