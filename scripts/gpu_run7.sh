#!/bin/bash
set +e
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
timeout 1700 python -m pytest tests -m gpu -q > gpurun_out/pytest_gpu.log 2>&1; echo "rc=$?"; tail -30 gpurun_out/pytest_gpu.log
