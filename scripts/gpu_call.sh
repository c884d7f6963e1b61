#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/final
( time timeout 1500 python -m pytest tests/ -q -m gpu -p no:cacheprovider -x ) 2>&1 | tail -8 | tee gpurun_out/final/pytest_gpu.log
timeout 120 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 | tee -a gpurun_out/final/pytest_gpu.log
timeout 300 python bench.py 2>&1 | tail -1 > gpurun_out/final/bench_default.json; python -c "
import json; d=json.loads(open('gpurun_out/final/bench_default.json').read()); print('bench', round(d['value'],1), d['ms_per_step'], round(d['roofline']['frac'],4), d['two_stage_exact']['value'], d['cpu_baseline']['value'])"
