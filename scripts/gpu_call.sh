#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/meas
export LD_LIBRARY_PATH=$PWD/oramacore_amd/csrc
echo "== vec f16"; timeout 300 scripts/native/bench_serving vec 10000000 150 8,64,256,512 f16 2>&1 | tee gpurun_out/meas/serving_vec.log | tail -7
echo "== vec shadow"; timeout 300 scripts/native/bench_serving vec 10000000 150 1,64,256,512 shadow 2>&1 | tee gpurun_out/meas/serving_vec_shadow.log | tail -7
echo "== hybrid shadow"; timeout 400 scripts/native/bench_serving hybrid 10000000 100 1,32,128 shadow 2>&1 | tee gpurun_out/meas/serving_hybrid_shadow.log | tail -8
