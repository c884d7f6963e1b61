#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/k2e
timeout 900 python -m pytest tests/test_vector_f16_gpu.py tests/test_two_stage_gpu.py tests/test_random_gpu.py -m gpu -x -q -p no:cacheprovider 2>&1 | tail -5
for d in 0 32; do echo "== ORAMA_K2C_DBG=$d"; ORAMA_K2C_DBG=$d timeout 300 python scripts/k2_epilogue_ablation.py 2>&1 | tail -1; done | tee gpurun_out/k2e/k2d_epi2.log
echo "== K2_DBG=2 (no appends)"; ORAMA_K2_DBG=2 timeout 300 python scripts/k2_epilogue_ablation.py 2>&1 | tail -1 | tee -a gpurun_out/k2e/k2d_epi2.log
timeout 200 python bench.py --workload c5 --rows 10000000 --steps 20 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('c5 shard', d['value'], d['ms_per_step'], d['roofline']['frac'])" | tee -a gpurun_out/k2e/k2d_epi2.log
