#!/bin/bash
cd $GRAFT_REPO_ROOT
R=$PWD
O=$R/gpurun_out/r03t
mkdir -p $O
timeout 1500 python -m pytest tests/test_bm25_ranges_gpu.py tests/test_token_score_gpu.py tests/test_fulltext_gpu.py tests/test_random_gpu.py tests/test_batcher_gpu.py tests/test_sharded_fulltext_gpu.py tests/test_post_append_gpu.py -q -m gpu -p no:cacheprovider -x 2>&1 | tail -15 | tee $O/pytest.log
timeout 900 python -m pytest tests/test_full_size_gpu.py -q -m gpu -p no:cacheprovider -x -k "c4 or bm25 or hybrid" 2>&1 | tail -5 | tee $O/pytest_c4.log
python scripts/k3r_filter_probe.py 2>&1 | tee $O/probe.log
ORAMA_K3R_DBG=8 python scripts/k3r_filter_probe.py 2>&1 | tee $O/probe_merge_tree.log
