#!/bin/bash
O=gpurun_out/r02ts
mkdir -p $O
cd $GRAFT_REPO_ROOT
( time timeout 900 python -m pytest tests/test_full_size_gpu.py::test_ns_full_size_two_stage_equals_the_fp32_scan tests/test_two_stage_gpu.py tests/test_vector_gpu.py tests/test_vector_f16_gpu.py tests/test_batcher_gpu.py -m gpu -x -q -p no:cacheprovider ) > $O/pytest.log 2>&1; tail -6 $O/pytest.log
timeout 400 scripts/native/bench_serving vec 10000000 100 1,8,64,256,512 shadow > $O/serving_shadow.log 2>&1; cat $O/serving_shadow.log
timeout 400 python bench.py --steps 30 --warmup 3 > $O/bench_ns.json 2> $O/bench_ns.err; tail -2 $O/bench_ns.err; python -c "
import json; d=json.loads(open('$O/bench_ns.json').read().strip().splitlines()[-1]); print(d['value'], d['roofline']['frac']); print(json.dumps(d.get('two_stage_exact'), indent=1))"
