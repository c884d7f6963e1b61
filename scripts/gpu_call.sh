#!/bin/bash
cd $GRAFT_REPO_ROOT
for cfg in "1 2" "1 4" "1 8" "2 2" "4 1" "4 2"; do set -- $cfg; echo "== TW=$1 BPC=$2"; ORAMA_F16_SOLO_TW=$1 ORAMA_F16_SOLO_BPC=$2 timeout 300 python scripts/two_stage_breakdown.py 2>&1 | tail -3 | head -1; done
