#!/bin/bash
O=gpurun_out/r02ts4
mkdir -p $O
cd $GRAFT_REPO_ROOT
( time timeout 900 python -m pytest tests/test_two_stage_gpu.py tests/test_full_size_gpu.py::test_ns_full_size_two_stage_equals_the_fp32_scan tests/test_abi.py -x -q -p no:cacheprovider ) > $O/pytest.log 2>&1; grep "passed\|failed\|Error" $O/pytest.log | head
timeout 200 python bench.py --workload c2 --steps 200 --no-cpu-baseline 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('c2', round(d['value']), d.get('two_stage_exact'))"
