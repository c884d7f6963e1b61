#!/bin/bash
# tile-pad (channel skew) experiment for K2: same sweep under different ORAMA_F16_TILE_PAD
set +e
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
for PAD in 0 256 1024 4096 768; do
  echo "== pad $PAD"
  ORAMA_F16_TILE_PAD=$PAD timeout 300 python scripts/sweep_f16.py --iters 6 2>&1 | grep -E '"kc": (8|12), "nbuf": 3|BEST' | cut -c1-200
done
echo "== f16 tests with default pad"; timeout 900 python -m pytest tests/test_vector_f16_gpu.py -m gpu -q 2>&1 | tail -3
