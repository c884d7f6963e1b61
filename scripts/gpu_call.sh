#!/bin/bash
# final check: the whole parity suite twice, smoke, serving benches
O=gpurun_out/r02final2
mkdir -p $O
cd $GRAFT_REPO_ROOT
for i in 1 2; do ( time timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider ) > $O/pytest_$i.log 2>&1; grep "passed\|failed" $O/pytest_$i.log; done
python -c "import __graft_entry__ as g; g.build(); g.smoke()" 2>&1 | tail -2
timeout 300 scripts/native/bench_serving bm25 10000000 400 1,8,32,128,512 > $O/serving_bm25.log 2>&1; cat $O/serving_bm25.log
timeout 300 python bench.py --workload c2 --steps 200 --no-cpu-baseline > $O/bench_c2.json 2>/dev/null; python -c "
import json; d=json.loads(open('$O/bench_c2.json').read().strip().splitlines()[-1]); print('c2', round(d['value']), d['two_stage_exact']['value'], d['two_stage_exact']['batch64_queries_per_s'])"
