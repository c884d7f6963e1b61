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
