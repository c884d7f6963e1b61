#!/bin/bash
# First GPU pass: smoke, GPU parity tests, K1 launch-geometry sweep, bench, rocprof kernel stats.
set +e
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== smoke"; timeout 600 python __graft_entry__.py --smoke > gpurun_out/smoke.log 2>&1; echo "rc=$?"; tail -3 gpurun_out/smoke.log
echo "== pytest gpu"; timeout 1200 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; echo "rc=$?"; tail -15 gpurun_out/pytest_gpu.log
echo "== sweep"; timeout 600 python scripts/sweep_scan.py --out gpurun_out/sweep_ns.json > gpurun_out/sweep_ns.log 2>&1; echo "rc=$?"; tail -4 gpurun_out/sweep_ns.log
echo "== bench ns"; timeout 900 python bench.py --steps 50 --warmup 5 > gpurun_out/bench_ns.log 2>&1; echo "rc=$?"; tail -2 gpurun_out/bench_ns.log
echo "== bench c2"; timeout 600 python bench.py --workload c2 --steps 200 --warmup 10 --no-cpu-baseline > gpurun_out/bench_c2.log 2>&1; echo "rc=$?"; tail -2 gpurun_out/bench_c2.log
echo "== rocprof"; (cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d "$OLDPWD/gpurun_out/prof_ns" -o ns -- python "$OLDPWD/bench.py" --steps 20 --warmup 3 --no-cpu-baseline > "$OLDPWD/gpurun_out/rocprof_ns.log" 2>&1); echo "rc=$?"; tail -2 gpurun_out/rocprof_ns.log
find gpurun_out/prof_ns -name "*stats*" | head; 
