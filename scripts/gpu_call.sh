#!/bin/bash
O=gpurun_out/r02fin3
mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 600 python scripts/bench_two_stage.py > $O/bench_two_stage.log 2>&1; grep -v "^{" $O/bench_two_stage.log | tail -5
timeout 400 python bench.py --steps 50 --warmup 5 > $O/bench_ns.json 2> $O/bench_ns.err; python -c "
import json; d=json.loads(open('$O/bench_ns.json').read().strip().splitlines()[-1]); print('ns', round(d['value'],1), 'frac', round(d['roofline']['frac'],3), 'two_stage', round(d['two_stage_exact']['value'],1), 'b64', round(d['two_stage_exact']['batch64_queries_per_s']), 'fallbacks', d['two_stage_exact']['fallbacks'])"
timeout 400 python scripts/bench_hybrid.py --steps 100 --warmup 5 > $O/bench_c4.json 2>$O/bench_c4.err; python -c "
import json; d=json.load(open('$O/bench_c4.json')); print('c4 hybrid', round(d['value'],1), 'two-stage', round(d['hybrid_two_stage_exact']['value'],1), 'bm25 batch', round(d['bm25_only']['value']))"
timeout 300 scripts/native/bench_serving vec 10000000 100 1,64,512 shadow > $O/serving_shadow.log 2>&1; tail -7 $O/serving_shadow.log
