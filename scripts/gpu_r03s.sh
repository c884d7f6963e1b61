#!/bin/bash
cd $GRAFT_REPO_ROOT
R=$PWD
O=$R/gpurun_out/r03s
mkdir -p $O
export TMPDIR=/tmp
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $O/trace -o k -- python $R/scripts/k3r_filter_probe.py > $O/run.log 2>&1
cd $R
python scripts/rocpd_summary.py $(find $O/trace -name "*results.db" | head -1) 2>&1 | head -30 | cut -c1-230 | tee $O/k3r_filter_kernel_stats.md
find $O -name "*.db" -delete
