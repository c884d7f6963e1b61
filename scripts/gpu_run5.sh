#!/bin/bash
set +e
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== pytest gpu (all)"; timeout 1700 python -m pytest tests -m gpu -q > gpurun_out/pytest_gpu.log 2>&1; echo "rc=$?"; tail -8 gpurun_out/pytest_gpu.log
echo "== bench c3"; timeout 900 python bench.py --workload c3 --steps 30 --warmup 3 > gpurun_out/bench_c3.log 2>&1; echo "rc=$?"; tail -1 gpurun_out/bench_c3.log | cut -c1-1500
echo "== rocprof c3"; (cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d "$OLDPWD/gpurun_out/prof_c3" -o c3 -- python "$OLDPWD/bench.py" --workload c3 --steps 10 --warmup 2 > "$OLDPWD/gpurun_out/rocprof_c3.log" 2>&1); echo "rc=$?"
python scripts/rocpd_summary.py gpurun_out/prof_c3/c3_results.db 2>/dev/null | cut -c1-260 | head -16
