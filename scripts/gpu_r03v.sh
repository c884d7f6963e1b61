#!/bin/bash
cd $GRAFT_REPO_ROOT
R=$PWD
O=$R/gpurun_out/r03v
mkdir -p $O
timeout 1500 python -m pytest tests/test_shard_group_gpu.py tests/test_multirank_gpu.py tests/test_sharded_fulltext_gpu.py tests/test_batcher_gpu.py tests/test_stress_gpu.py -q -m gpu -p no:cacheprovider -x 2>&1 | tail -12 | tee $O/pytest.log
cd scripts/native
for S in 1 4; do
  timeout 600 ./bench_serving vec 10000000 100 8,64,512 f16 $S 2>&1 | tee $O/serving_vec_f16_shards$S.log
done
for S in 1 4; do
  timeout 600 ./bench_serving vec 10000000 12 8,64 f32 $S 2>&1 | tee $O/serving_vec_f32_shards$S.log
done
