#!/bin/bash
# round 2, GPU call 1: parity suite (incl. the new full-size C5 / C4 tests) + K2c ablations + two micro probes
set -x
mkdir -p gpurun_out/r02c1
cd $GRAFT_REPO_ROOT
( time timeout 900 python -m pytest tests -m gpu -x -q ) > gpurun_out/r02c1/pytest_gpu.log 2>&1
tail -5 gpurun_out/r02c1/pytest_gpu.log
timeout 120 scripts/micro/ldsdma_order_probe > gpurun_out/r02c1/ldsdma_order_probe.log 2>&1
cat gpurun_out/r02c1/ldsdma_order_probe.log
timeout 120 scripts/micro/vmm_probe > gpurun_out/r02c1/vmm_probe.log 2>&1
cat gpurun_out/r02c1/vmm_probe.log
timeout 300 python scripts/k2c_ablate.py > gpurun_out/r02c1/k2c_ablate.log 2>&1
cat gpurun_out/r02c1/k2c_ablate.log
