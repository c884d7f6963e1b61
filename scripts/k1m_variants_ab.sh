#!/bin/bash
# K1m ring depth / load pairing / query-fragment buffering, A/B in ONE lease (comparison build): ORAMA_K1M_VARIANT = depth x 100 + 10 pair + 1 bdbl
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export ORAMA_COMPARISON_KERNELS=1
for V in ${@:-401 400 410 411 600 610 611 800 810 401}; do
  echo "== ORAMA_K1M_VARIANT=$V"
  ORAMA_K1M_VARIANT=$V timeout 120 python -m pytest tests/test_vector_f32_mfma_gpu.py -x -q -p no:cacheprovider -k "same_bits or filter_path" 2>&1 | tail -1
  ORAMA_K1M_VARIANT=$V timeout 200 python scripts/k1m_probe.py --batches 32,32 --reps 10 2>&1 | tail -2 | cut -c1-330
done
