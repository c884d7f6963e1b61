#!/bin/bash
O=gpurun_out/r02hyb
mkdir -p $O
cd $GRAFT_REPO_ROOT
( time timeout 1500 python -m pytest tests -m gpu -x -q -p no:cacheprovider ) > $O/pytest.log 2>&1; grep "passed\|failed\|rror" $O/pytest.log | head -3
timeout 400 scripts/native/bench_serving hybrid 10000000 40 128,512 shadow > $O/serving_hybrid_shadow2.log 2>&1; tail -4 $O/serving_hybrid_shadow2.log
