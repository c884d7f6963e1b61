#!/bin/bash
cd $GRAFT_REPO_ROOT
R=$PWD; O=$R/gpurun_out/meas; mkdir -p $O
export LD_LIBRARY_PATH=$PWD/oramacore_amd/csrc
echo "== serving bm25"; timeout 300 scripts/native/bench_serving bm25 10000000 1500 32,128,512 2>&1 | tee $O/serving_bm25.log | tail -8
cd /tmp; export TMPDIR=/tmp
QB=1 timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof_ts1 -o ts -- python $R/scripts/two_stage_breakdown.py > $O/rocprof_ts1.log 2>&1; echo rc=$?
cd $R
python scripts/rocpd_summary.py $(find $O/prof_ts1 -name "*results.db" | head -1) > $O/ts1_kernel_stats.md 2>$O/ts1.err
find $O -name "*.db" -delete
head -12 $O/ts1_kernel_stats.md | cut -c1-180
