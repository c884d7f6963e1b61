#!/bin/bash
cd $GRAFT_REPO_ROOT
R=$PWD
O=$R/gpurun_out/r03m
mkdir -p $O
export TMPDIR=/tmp
cd /tmp
timeout 300 rocprofv3 --kernel-trace -d $O/trace -o h -- python $R/scripts/hybrid_tail_run.py > $O/run.log 2>&1
cd $R
python scripts/hybrid_tail_run.py --report $O/trace 2>&1 | tee $O/hybrid_tail_timeline.log
find $O -name "*.db" -delete
