#!/bin/bash
# round 3, call A: full GPU suite (incl. the multi-rank jobs on the mock transport), the all-config bench line,
# a 2-rank self-launched bench on the loopback transport.
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03a
mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/ -q -m gpu -p no:cacheprovider -x 2>&1 | tail -40 > $O/pytest_gpu.log
tail -5 $O/pytest_gpu.log
echo "== bench (all configs, live PMC)"
( time timeout 600 python bench.py --steps 20 --warmup 5 ) > $O/bench_all.json 2> $O/bench_all.err
tail -c 1500 $O/bench_all.err
python - <<'PY'
import json
try:
    d=json.loads(open('gpurun_out/r03a/bench_all.json').read().strip().splitlines()[-1])
    print('NS', d['value'], d['roofline']['frac'], d['roofline'].get('traffic'), d['roofline'].get('traffic_source','')[:40])
    for k,v in d.get('configs',{}).items():
        print(k, round(v['value'],1), round(v['ms_per_step'],3), round(v['roofline']['frac'],3), v.get('bm25_only',{}).get('value'), v.get('bm25_only',{}).get('roofline',{}).get('frac'))
except Exception as e:
    print('bench parse failed', e)
PY
echo "== 2 ranks, self-launched, loopback transport on one GPU"
( time ORAMA_RCCL_LIB=$PWD/tests/mock_rccl/libmock_rccl.so timeout 300 python bench.py --gpus 2 --rows 2000000 --steps 10 --warmup 2 ) > $O/bench_2rank_mock.json 2> $O/bench_2rank_mock.err
tail -c 600 $O/bench_2rank_mock.json; tail -c 800 $O/bench_2rank_mock.err
