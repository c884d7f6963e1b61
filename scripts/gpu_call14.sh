#!/bin/bash
set -x
mkdir -p gpurun_out/r02c14
cd $GRAFT_REPO_ROOT
for i in 1 2; do ( time timeout 1200 python -m pytest tests -m gpu -q -p no:cacheprovider ) > gpurun_out/r02c14/full_$i.log 2>&1; tail -4 gpurun_out/r02c14/full_$i.log | grep -v "^$"; grep "^FAILED\|^E  " gpurun_out/r02c14/full_$i.log | head; done
