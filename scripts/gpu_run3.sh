#!/bin/bash
# RCCL plumbing test on one GPU: torch.distributed.run with 1 rank + forced all-gather/merge path.
set +e
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
echo "== torchrun 1 rank, forced exchange"
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --steps 30 --warmup 5 --force-exchange --no-cpu-baseline > gpurun_out/bench_exchange.log 2>&1; echo "rc=$?"; tail -3 gpurun_out/bench_exchange.log | cut -c1-900
echo "== plain"
timeout 600 python bench.py --steps 100 --warmup 10 > gpurun_out/bench_ns.log 2>&1; echo "rc=$?"; tail -1 gpurun_out/bench_ns.log | cut -c1-1800
