#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/meas
timeout 900 python -m pytest tests/test_vector_f16_gpu.py tests/test_random_gpu.py -m gpu -x -q -p no:cacheprovider 2>&1 | tail -2
for d in 0 1 0; do echo "== ORAMA_K2_DBG=$d"; ORAMA_K2_DBG=$d timeout 300 python scripts/k2_epilogue_ablation.py 2>&1 | tail -2 | head -1; done | tee gpurun_out/meas/k2_prefetch.log
timeout 300 python bench.py --workload c3 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('c3', d['value'], d['ms_per_step'], d['roofline']['frac'])" | tee -a gpurun_out/meas/k2_prefetch.log
