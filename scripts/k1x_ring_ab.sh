#!/bin/bash
# K1x register-ring depth, A/B in one lease (comparison build): ORAMA_K1X_RING = 4 / 6 (default) / 8 (32 queries only)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export ORAMA_COMPARISON_KERNELS=1
for V in 6 4 8 6; do
  echo "== ORAMA_K1X_RING=$V"
  ORAMA_K1X_RING=$V timeout 200 python scripts/k1m_probe.py --batches 32,64,32,64 --reps 10 2>&1 | tail -2 | cut -c1-260
done
