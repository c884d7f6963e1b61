#!/bin/bash
O=gpurun_out/r02k1h
mkdir -p $O
cd $GRAFT_REPO_ROOT
for cfg in "2 2" "2 3" "2 4" "1 0"; do set -- $cfg; echo "== ORAMA_F16_SOLO=$1 BPC=$2"; ORAMA_F16_SOLO=$1 ORAMA_F16_SOLO_BPC=$2 timeout 300 python scripts/two_stage_breakdown.py 2>&1 | tail -3 | head -1; done
ORAMA_F16_SOLO=2 timeout 600 python -m pytest tests/test_two_stage_gpu.py -m gpu -x -q -p no:cacheprovider 2>&1 | grep "passed\|failed\|Error" | head -3
