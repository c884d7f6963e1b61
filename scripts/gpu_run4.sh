#!/bin/bash
set +e
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== pytest f16"; timeout 1500 python -m pytest tests/test_vector_f16_gpu.py -m gpu -q -x > gpurun_out/pytest_f16.log 2>&1; echo "rc=$?"; tail -30 gpurun_out/pytest_f16.log
echo "== bench c3"; timeout 900 python bench.py --workload c3 --steps 20 --warmup 3 > gpurun_out/bench_c3.log 2>&1; echo "rc=$?"; tail -2 gpurun_out/bench_c3.log | cut -c1-1500
