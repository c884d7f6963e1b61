#!/bin/bash
set -x
mkdir -p gpurun_out/r02c17
cd $GRAFT_REPO_ROOT
O=gpurun_out/r02c17
for i in 1 2 3; do NCCL_DEBUG=WARN timeout 90 python bench.py --rows 1250000 --force-exchange --steps 200 --no-cpu-baseline > $O/bench_exchange_$i.json 2> $O/bench_exchange_$i.err; echo "rc=$?"; tail -c 200 $O/bench_exchange_$i.json; tail -3 $O/bench_exchange_$i.err; done
timeout 200 python bench.py --workload c5 --rows 10000000 --steps 20 --warmup 3 --no-cpu-baseline > $O/bench_c5_shard.json 2> $O/bench_c5.err; python -c "
import json; d=json.load(open('$O/bench_c5_shard.json')); r=d['roofline']; print('c5 shard QPS %.0f ms/step %.3f scan %.0f GB/s avg launch %.3f ms'%(d['value'], d['ms_per_step'], r['achieved'], r['avg_launch_ms']))"
timeout 200 python bench.py --workload c3 --steps 30 --warmup 3 --no-cpu-baseline > $O/bench_c3.json 2> $O/bench_c3.err; python -c "
import json; d=json.load(open('$O/bench_c3.json')); r=d['roofline']; print('c3 QPS %.0f ms/step %.3f scan %.0f GB/s'%(d['value'], d['ms_per_step'], r['achieved']))"
