#!/bin/bash
O=gpurun_out/r02fused
mkdir -p $O
cd $GRAFT_REPO_ROOT
( time timeout 1200 python -m pytest tests/test_full_size_gpu.py tests/test_vector_gpu.py tests/test_two_stage_gpu.py tests/test_shard_group_gpu.py tests/test_token_score_gpu.py -m gpu -x -q -p no:cacheprovider ) > $O/pytest.log 2>&1; grep "passed\|failed\|rror" $O/pytest.log | head -5
timeout 400 python bench.py --steps 50 --warmup 5 > $O/bench_ns.json 2> $O/bench_ns.err; tail -2 $O/bench_ns.err; python -c "
import json; d=json.loads(open('$O/bench_ns.json').read().strip().splitlines()[-1]); print('ns', round(d['value'],1), 'QPS', 'frac', round(d['roofline']['frac'],3), 'scan ms', round(d['roofline']['avg_launch_ms'],3), 'p50', round(d['latency_ms_p50_host_api'],3), 'two_stage', round(d['two_stage_exact']['value'],1), round(d['two_stage_exact']['plain_fp32_same_api'],1))"
timeout 300 python scripts/bench_hybrid.py --steps 100 --warmup 5 --no-two-stage > $O/bench_c4.json 2>$O/bench_c4.err; python -c "
import json; d=json.load(open('$O/bench_c4.json')); print('c4 hybrid QPS', round(d['value'],1), 'frac', round(d['roofline']['frac'],3))"
