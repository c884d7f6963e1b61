#!/bin/bash
# K3r evidence: rocprofv3 kernel stats and PMC traffic of the BM25 batch path (10 M docs, 12 tokens, top-100)
set +e
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R=$PWD
O=$R/gpurun_out/r02k3rprof
rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
CMD="python $R/scripts/bench_bm25_threads.py --threads 0 --batch-callers 1 --scorers k3r"
cd /tmp
echo "== rocprof"; timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof -o k3r -- $CMD > $O/rocprof.log 2>&1; echo rc=$?
for C in FETCH_SIZE WRITE_SIZE; do
  echo "== pmc $C"; timeout 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $O/pmc/$C -o p -- $CMD > $O/pmc_$C.log 2>&1
done
cd $R
python scripts/rocpd_summary.py $(find $O/prof -name "*results.db" | head -1) > $O/k3r_kernel_stats.md 2>$O/k3r_kernel_stats.err
AVG=$(grep "postings referenced" $O/rocprof.log | awk '{print $5}')
python scripts/pmc_summary.py $O/pmc range_score_kernel $((AVG * 32 * 16)) mean > $O/pmc_k3r_range_score.json 2>$O/pmc_s.err
python scripts/pmc_summary.py $O/pmc range_bounds_kernel $((AVG * 32 * 4)) mean > $O/pmc_k3r_range_bounds.json 2>$O/pmc_b.err
python scripts/pmc_summary.py $O/pmc keys_reduce_kernel $((AVG * 32 * 8)) mean > $O/pmc_k3r_keys_reduce.json 2>$O/pmc_k.err
find $O -name "*.csv" -size +2M -delete; find $O -name "*.db" -delete
grep -v "^{" $O/rocprof.log | tail -8; head -12 $O/k3r_kernel_stats.md | cut -c1-200; cat $O/pmc_k3r_range_score.json; cat $O/pmc_s.err | tail -3
echo "== c4"; timeout 300 python scripts/bench_hybrid.py --steps 100 --warmup 5 > $O/bench_c4.json 2>$O/bench_c4.err; tail -3 $O/bench_c4.err; python -c "
import json; d=json.load(open('$O/bench_c4.json')); print('hybrid QPS', round(d['value'],1)); print(json.dumps(d['bm25_only'], indent=1))"
