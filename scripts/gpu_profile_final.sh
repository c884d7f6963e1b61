#!/bin/bash
# Round-end measurement set: bench lines (NS / C2 / C3), rocprofv3 kernel traces of the same commands,
# PMC traffic passes for C2 and C3 (FETCH_SIZE / WRITE_SIZE in separate runs, kernel-trace only), full GPU suite.
set +e
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R=$PWD
O=$R/gpurun_out/final
rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
echo "== pytest -m gpu"; timeout 1500 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; tail -3 $O/pytest_gpu.log
echo "== bench ns"; timeout 600 python bench.py > $O/bench_ns.json 2> $O/bench_ns.err; tail -c 400 $O/bench_ns.json; echo
echo "== bench c2"; timeout 300 python bench.py --workload c2 --steps 200 --no-cpu-baseline > $O/bench_c2.json 2> $O/bench_c2.err
echo "== bench c3"; timeout 300 python bench.py --workload c3 --steps 30 --warmup 3 --no-cpu-baseline > $O/bench_c3.json 2> $O/bench_c3.err
echo "== bench ns exchange (1 rank, RCCL path)"; timeout 300 python bench.py --rows 1250000 --force-exchange --steps 200 --no-cpu-baseline > $O/bench_exchange_shard.json 2> $O/bench_exchange.err
cd /tmp
echo "== rocprof ns"; timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof_ns -o ns -- python $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline > $O/rocprof_ns.log 2>&1; echo rc=$?
echo "== rocprof c3"; timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof_c3 -o c3 -- python $R/bench.py --workload c3 --steps 10 --warmup 2 --no-cpu-baseline > $O/rocprof_c3.log 2>&1; echo rc=$?
echo "== rocprof c4"; timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof_c4 -o c4 -- python $R/scripts/bench_hybrid.py --steps 20 --warmup 3 --no-check > $O/rocprof_c4.log 2>&1; echo rc=$?
for W in c2 c3; do
  for C in FETCH_SIZE WRITE_SIZE; do
    echo "== pmc $W $C"
    timeout 600 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $O/pmc_$W/$C -o p -- python $R/bench.py --workload $W --steps 5 --warmup 2 --no-cpu-baseline > $O/pmc_${W}_$C.log 2>&1; echo rc=$?
  done
done
cd $R
for W in ns c3 c4; do python scripts/rocpd_summary.py $(find $O/prof_$W -name "*results.db" | head -1) > $O/${W}_kernel_stats.md 2>$O/${W}_kernel_stats.err; done
python scripts/pmc_summary.py $O/pmc_c2 vec_scan_f32_kernel 1536000000 > $O/pmc_c2_vec_scan.json 2>$O/pmc_c2.err
python scripts/pmc_summary.py $O/pmc_c3 vec_scan_f16_kernel 7680000000 mean > $O/pmc_c3_vec_scan.json 2>$O/pmc_c3.err
find $O -name "*.csv" -size +2M -delete; find $O -name "*.db" -delete
du -sh $O; head -8 $O/ns_kernel_stats.md | cut -c1-200; cat $O/pmc_c2_vec_scan.json | head -30
echo "== c5 per-GPU shard probe (10M x 768 fp16, Q = 64 / 128 / 256)"; timeout 300 python scripts/k2c_probe.py > $O/k2c_probe.log 2>&1; tail -3 $O/k2c_probe.log
echo "== bench c5 on one GPU"; timeout 600 python bench.py --workload c5 --steps 10 --warmup 2 --no-cpu-baseline > $O/bench_c5_1gpu.json 2> $O/bench_c5.err; tail -c 300 $O/bench_c5_1gpu.json; echo
echo "== batcher"; timeout 300 python scripts/bench_batcher.py --dtype f16 --threads 128 > $O/bench_batcher_f16.json 2>/dev/null; timeout 300 python scripts/bench_batcher.py --dtype f16 --threads 512 --per-thread 8 --max-batch 256 > $O/bench_batcher_f16_b256.json 2>/dev/null; timeout 300 python scripts/bench_batcher.py --dtype f32 --threads 32 --per-thread 10 --max-batch 16 > $O/bench_batcher_f32.json 2>/dev/null
echo "== c4 / c1"; timeout 300 python scripts/bench_hybrid.py --steps 50 --warmup 5 > $O/bench_c4.json 2>/dev/null; timeout 200 python scripts/bench_c1.py > $O/bench_c1.json 2>/dev/null
cd /tmp; echo "== rocprof c5 shard"; NQ=256 timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof_c5 -o c5 -- python $R/scripts/k2c_probe.py > $O/rocprof_c5.log 2>&1; cd $R
python scripts/rocpd_summary.py $(find $O/prof_c5 -name "*results.db" | head -1) > $O/c5_kernel_stats.md 2>/dev/null; find $O -name "*.db" -delete
for f in bench_batcher_f16 bench_batcher_f16_b256 bench_batcher_f32; do python -c "
import json,sys
d=json.load(open('$O/$f.json')); print('$f', 'direct %.0f QPS'%d['direct_calls']['qps'], 'batched %.0f QPS'%d['through_batcher']['qps'], 'mean batch %.1f'%d['through_batcher']['mean_batch'], d['batched_equals_solo'])"; done
