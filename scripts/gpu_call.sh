#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/k2e
timeout 600 python -m pytest tests/test_vector_f16_gpu.py tests/test_two_stage_gpu.py tests/test_random_gpu.py -m gpu -x -q -p no:cacheprovider 2>&1 | tail -5
for d in 0 2 0 2; do echo "== ORAMA_K2_DBG=$d"; ORAMA_K2_DBG=$d timeout 300 python scripts/k2_epilogue_ablation.py 2>&1 | tail -2 | head -1; done | tee gpurun_out/k2e/ablation6.log
for f in 0 16; do echo "== grow $f";  if [ $f = 0 ]; then export ORAMA_F16_CHUNK_GROW=0; else export ORAMA_F16_CHUNK_GROW=1 ORAMA_F16_GROW_FACTOR=$f; fi; timeout 300 python bench.py --workload c3 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['frac'])"; done | tee gpurun_out/k2e/grow4.log
