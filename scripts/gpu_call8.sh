#!/bin/bash
set -x
mkdir -p gpurun_out/r02c8
cd $GRAFT_REPO_ROOT
O=gpurun_out/r02c8
( time timeout 600 python bench.py --steps 20 --warmup 5 ) > $O/bench_ns.json 2> $O/bench_ns.err; tail -c 1500 $O/bench_ns.json; tail -5 $O/bench_ns.err
timeout 300 python bench.py --rows 1250000 --force-exchange --steps 200 --no-cpu-baseline > $O/bench_exchange_shard.json 2> $O/bench_exchange.err; tail -c 600 $O/bench_exchange_shard.json; tail -3 $O/bench_exchange.err
timeout 300 python bench.py --workload c5 --rows 10000000 --steps 10 --warmup 2 --no-cpu-baseline > $O/bench_c5_shard.json 2> $O/bench_c5.err; tail -c 900 $O/bench_c5_shard.json; tail -3 $O/bench_c5.err
timeout 300 python bench.py --workload c3 --steps 30 --warmup 3 --no-cpu-baseline > $O/bench_c3.json 2> $O/bench_c3.err; tail -c 700 $O/bench_c3.json; tail -3 $O/bench_c3.err
