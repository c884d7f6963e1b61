#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03e
mkdir -p $O
run() { echo "== $*"; env "$@" MODES=2,4 NQ=256 timeout 200 python scripts/k2d_probe.py 2>&1 | grep mode; }
run ORAMA_QS_LAG=0 | tee -a $O/exp.log
run ORAMA_QS_LAG=0 DBG=32 | tee -a $O/exp.log
run ORAMA_QS_LAG=0 ORAMA_F16_CHUNK_GROW=1 | tee -a $O/exp.log
run ORAMA_QS_LAG=1 ORAMA_F16_CHUNK_GROW=1 | tee -a $O/exp.log
run ORAMA_QS_LAG=0 ORAMA_K2_DBG=2 | tee -a $O/exp.log
