#!/bin/bash
O=gpurun_out/r02pb
mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 400 scripts/native/bench_serving hybrid 10000000 30 1,8,32,64 > $O/serving_hybrid.log 2>&1; cat $O/serving_hybrid.log
