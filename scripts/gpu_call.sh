#!/bin/bash
cd $GRAFT_REPO_ROOT
for rows in 1250000 2500000 5000000; do for f in 0 1; do
ORAMA_FUSED_TOPK=$f timeout 200 python bench.py --rows $rows --steps 200 --warmup 10 --no-cpu-baseline --no-two-stage 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('rows $rows fused $f:', round(d['value'],1), 'QPS', round(d['ms_per_step'],4), 'ms/step | scan ms', round(d['roofline']['avg_launch_ms'],4), 'frac', round(d['roofline']['frac'],3))"
done; done
