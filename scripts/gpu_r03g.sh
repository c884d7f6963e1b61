#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03g
mkdir -p $O
run() { echo "== $*"; env "$@" MODES=4 NQ=256 timeout 200 python scripts/k2d_probe.py 2>&1 | grep mode; }
for D in 32 34 40 42; do run DBG=$D | tee -a $O/exp.log; done
for D in 0 32 34 42; do run ORAMA_QS_LAG=0 DBG=$D | tee -a $O/exp.log; done
