#!/bin/bash
# (round 6: sweep / A-B variables are read by the COMPARISON flavour only — liborama_hip_cmp.so, built and loaded with this set)
export ORAMA_COMPARISON_KERNELS=1
# K2 register ring (k-steps per chunk x chunks) on the C3 shape, 10 M x 768 fp16: scan ms per pass at 16 and 64 queries
cd $GRAFT_REPO_ROOT
for CFG in "8 2" "8 3" "8 4" "12 2" "12 3" "16 2"; do
  set -- $CFG
  echo "== KC=$1 NBUF=$2"
  ORAMA_F16_KC=$1 ORAMA_F16_NBUF=$2 MODES=4 NQ=16,64 timeout 200 python scripts/k2d_probe.py 2>&1 | tail -2
done
