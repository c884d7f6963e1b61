#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/meas
timeout 900 python -m pytest tests/test_vector_f16_gpu.py tests/test_two_stage_gpu.py tests/test_random_gpu.py -m gpu -x -q -p no:cacheprovider 2>&1 | tail -3
timeout 500 python scripts/bench_two_stage.py --batches 1,8,64,256 2>&1 | tee gpurun_out/meas/two_stage2.log | grep "^batch\|fallbacks" | cut -c1-200
timeout 300 python bench.py --workload c3 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('c3', d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['topk_select_ms_per_step'])"
