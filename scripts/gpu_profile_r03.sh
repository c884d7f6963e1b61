#!/bin/bash
# Round-3 measurement set: the whole GPU suite, the driver's bench command (every config + live PMC traffic), rocprofv3
# kernel traces of the bench legs (stats tables), PMC passes for the wide fp16 kernel and the K3r launches.
set +e
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R=$PWD
O=$R/gpurun_out/r03final
rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
echo "== pytest -m gpu"; timeout 2400 python -m pytest tests -q -m gpu -p no:cacheprovider 2>&1 | tail -6 | tee $O/pytest_gpu.log
echo "== bench (driver command)"; timeout 900 python bench.py > $O/bench_all.json 2> $O/bench_all.err; tail -c 400 $O/bench_all.json; echo
cd /tmp
echo "== rocprof ns"; timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof_ns -o ns -- python $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-two-stage --configs none --no-pmc > $O/rocprof_ns.log 2>&1; echo rc=$?
echo "== rocprof c3"; timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof_c3 -o c3 -- python $R/bench.py --workload c3 --steps 10 --warmup 2 --no-cpu-baseline --configs none --no-pmc > $O/rocprof_c3.log 2>&1; echo rc=$?
echo "== rocprof c5 shard"; timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof_c5 -o c5 -- python $R/bench.py --workload c5 --rows 10000000 --steps 10 --warmup 2 --no-cpu-baseline --configs none --no-pmc > $O/rocprof_c5.log 2>&1; echo rc=$?
echo "== rocprof c4 (ns + c4 leg)"; timeout 400 rocprofv3 --kernel-trace --stats -d $O/prof_c4 -o c4 -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-two-stage --configs c4 --no-pmc > $O/rocprof_c4.log 2>&1; echo rc=$?
for C in FETCH_SIZE WRITE_SIZE; do
  echo "== pmc c5 $C"; timeout 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $O/pmc_c5/$C -o p -- python $R/bench.py --workload c5 --rows 10000000 --steps 5 --warmup 2 --no-cpu-baseline --configs none --no-pmc > $O/pmc_c5_$C.log 2>&1
  echo "== pmc c3 $C"; timeout 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $O/pmc_c3/$C -o p -- python $R/bench.py --workload c3 --steps 5 --warmup 2 --no-cpu-baseline --configs none --no-pmc > $O/pmc_c3_$C.log 2>&1
  echo "== pmc k3r $C"; timeout 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $O/pmc_k3r/$C -o p -- python $R/scripts/k3r_chunk_probe.py > $O/pmc_k3r_$C.log 2>&1
done
cd $R
for W in ns c3 c5 c4; do python scripts/rocpd_summary.py $(find $O/prof_$W -name "*results.db" | head -1) > $O/${W}_kernel_stats.md 2>$O/${W}_kernel_stats.err; done
python scripts/pmc_summary.py $O/pmc_c5 vec_scan_f16_qs_kernel 3072000000 mean > $O/pmc_c5_vec_scan.json 2>$O/pmc_c5.err
python scripts/pmc_summary.py $O/pmc_c3 vec_scan_f16_kernel 7680000000 mean > $O/pmc_c3_vec_scan.json 2>$O/pmc_c3.err
python scripts/pmc_summary.py $O/pmc_k3r range_score_kernel 155000000 mean > $O/pmc_k3r_range_score.json 2>$O/pmc_k3r.err
find $O -name "*.csv" -size +2M -delete; find $O -name "*.db" -delete
du -sh $O; head -5 $O/ns_kernel_stats.md | cut -c1-200; head -6 $O/c5_kernel_stats.md | cut -c1-200; head -8 $O/c4_kernel_stats.md | cut -c1-200; cat $O/pmc_c5_vec_scan.json | head -20
