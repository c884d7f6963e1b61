#!/bin/bash
O=gpurun_out/r02k3r7
mkdir -p $O
cd $GRAFT_REPO_ROOT
( time timeout 600 python -m pytest tests/test_bm25_ranges_gpu.py tests/test_fulltext_gpu.py tests/test_random_gpu.py tests/test_post_append_gpu.py tests/test_sharded_fulltext_gpu.py "tests/test_full_size_gpu.py::test_c4_full_size_bm25_bit_exact" -m gpu -x -q -p no:cacheprovider ) > $O/pytest.log 2>&1; grep "passed\|failed\|Error" $O/pytest.log | head -4
timeout 300 python scripts/bench_bm25_threads.py --threads 1,8 --scorers k3r > $O/bm25.log 2>&1; grep -v "^{" $O/bm25.log
