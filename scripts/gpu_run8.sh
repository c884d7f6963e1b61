#!/bin/bash
set +e
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_token_score_gpu.py tests/test_fulltext_gpu.py -m gpu -q 2>&1 | tail -6
timeout 900 python scripts/bench_hybrid.py --steps 50 --warmup 5 > gpurun_out/bench_c4.log 2>&1; echo rc=$?; tail -1 gpurun_out/bench_c4.log | python -c "
import sys,json
d=json.loads(sys.stdin.read())
print('hybrid fused QPS', d['value'], 'ms', d['ms_per_step']); print('two-call', d['hybrid_two_call_path']['value']); print('bm25', d['bm25_only']); print(d['hybrid_breakdown_ms_per_query']); print(d['parity_check'])"
