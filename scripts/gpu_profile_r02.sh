#!/bin/bash
# Round-2 measurement set: bench lines (NS / C2 / C3 / C5 per-GPU shard / exchange), rocprofv3 kernel traces of the same
# commands, PMC traffic passes (FETCH_SIZE / WRITE_SIZE / TCC hits in separate runs, kernel-trace only), C4 hybrid.
set +e
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R=$PWD
O=$R/gpurun_out/r02final
rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
echo "== bench ns"; timeout 300 python bench.py --steps 50 --warmup 5 > $O/bench_ns.json 2> $O/bench_ns.err; tail -c 300 $O/bench_ns.json; echo
echo "== bench c2"; timeout 200 python bench.py --workload c2 --steps 200 --no-cpu-baseline > $O/bench_c2.json 2> $O/bench_c2.err
echo "== bench c3"; timeout 200 python bench.py --workload c3 --steps 30 --warmup 3 --no-cpu-baseline > $O/bench_c3.json 2> $O/bench_c3.err
echo "== bench c5 per-GPU shard"; timeout 200 python bench.py --workload c5 --rows 10000000 --steps 20 --warmup 3 --no-cpu-baseline > $O/bench_c5_shard.json 2> $O/bench_c5.err
echo "== bench exchange (1 rank, RCCL inside the library)"; timeout 200 python bench.py --rows 1250000 --force-exchange --steps 200 --no-cpu-baseline > $O/bench_exchange_shard.json 2> $O/bench_exchange.err
echo "== c4"; timeout 200 python scripts/bench_hybrid.py --steps 100 --warmup 5 > $O/bench_c4.json 2>$O/bench_c4.err
cd /tmp
echo "== rocprof ns"; timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof_ns -o ns -- python $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-two-stage > $O/rocprof_ns.log 2>&1; echo rc=$?
echo "== rocprof c3"; timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof_c3 -o c3 -- python $R/bench.py --workload c3 --steps 10 --warmup 2 --no-cpu-baseline > $O/rocprof_c3.log 2>&1; echo rc=$?
echo "== rocprof c5 shard"; timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof_c5 -o c5 -- python $R/bench.py --workload c5 --rows 10000000 --steps 10 --warmup 2 --no-cpu-baseline > $O/rocprof_c5.log 2>&1; echo rc=$?
echo "== rocprof c4"; timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof_c4 -o c4 -- python $R/scripts/bench_hybrid.py --steps 20 --warmup 3 --no-check --no-two-stage > $O/rocprof_c4.log 2>&1; echo rc=$?
for C in FETCH_SIZE WRITE_SIZE; do
  echo "== pmc ns $C"; timeout 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $O/pmc_ns/$C -o p -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-two-stage > $O/pmc_ns_$C.log 2>&1
  echo "== pmc c3 $C"; timeout 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $O/pmc_c3/$C -o p -- python $R/bench.py --workload c3 --steps 5 --warmup 2 --no-cpu-baseline > $O/pmc_c3_$C.log 2>&1
  echo "== pmc c5 $C"; timeout 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $O/pmc_c5/$C -o p -- python $R/bench.py --workload c5 --rows 10000000 --steps 5 --warmup 2 --no-cpu-baseline > $O/pmc_c5_$C.log 2>&1
  echo "== pmc c4 $C"; timeout 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $O/pmc_c4/$C -o p -- python $R/scripts/bench_hybrid.py --steps 10 --warmup 2 --no-check --no-two-stage > $O/pmc_c4_$C.log 2>&1
done
echo "== pmc c5 L2 hit split"; timeout 300 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum --kernel-trace --output-format csv -d $O/pmc_c5/TCC -o p -- python $R/bench.py --workload c5 --rows 10000000 --steps 5 --warmup 2 --no-cpu-baseline > $O/pmc_c5_TCC.log 2>&1
cd $R
for W in ns c3 c5 c4; do python scripts/rocpd_summary.py $(find $O/prof_$W -name "*results.db" | head -1) > $O/${W}_kernel_stats.md 2>$O/${W}_kernel_stats.err; done
python scripts/pmc_summary.py $O/pmc_ns vec_scan_f32_kernel 30720000000 > $O/pmc_ns_vec_scan.json 2>$O/pmc_ns.err
python scripts/pmc_summary.py $O/pmc_c3 vec_scan_f16_kernel 7680000000 mean > $O/pmc_c3_vec_scan.json 2>$O/pmc_c3.err
python scripts/pmc_summary.py $O/pmc_c5 vec_scan_f16_pc_kernel 3072000000 mean > $O/pmc_c5_vec_scan.json 2>$O/pmc_c5.err
python scripts/pmc_summary.py $O/pmc_c4 bm25_accumulate_kernel 4900000 mean > $O/pmc_c4_bm25_accumulate.json 2>$O/pmc_c4a.err
python scripts/pmc_summary.py $O/pmc_c4 bm25_finalize_kernel 4900000 mean > $O/pmc_c4_bm25_finalize.json 2>$O/pmc_c4f.err
find $O -name "*.csv" -size +2M -delete; find $O -name "*.db" -delete
du -sh $O; head -6 $O/ns_kernel_stats.md | cut -c1-220; head -6 $O/c5_kernel_stats.md | cut -c1-220; cat $O/pmc_c5_vec_scan.json | head -30; cat $O/pmc_c4_bm25_accumulate.json | head -20
