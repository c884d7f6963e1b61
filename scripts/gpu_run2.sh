#!/bin/bash
# GPU pass 2: full GPU parity suite (vector + top-n + BM25F + hybrid), then the default bench.
set +e
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== pytest gpu"; timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/pytest_gpu.log 2>&1; echo "rc=$?"; tail -40 gpurun_out/pytest_gpu.log
echo "== bench ns"; timeout 900 python bench.py --steps 50 --warmup 5 --no-cpu-baseline > gpurun_out/bench_ns.log 2>&1; echo "rc=$?"; tail -1 gpurun_out/bench_ns.log | cut -c1-600
