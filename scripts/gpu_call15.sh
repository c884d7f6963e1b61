#!/bin/bash
set -x
mkdir -p gpurun_out/r02c15
cd $GRAFT_REPO_ROOT
O=gpurun_out/r02c15
( time timeout 900 python -m pytest tests/test_fulltext_gpu.py tests/test_token_score_gpu.py tests/test_facets_gpu.py tests/test_random_gpu.py tests/test_post_append_gpu.py tests/test_batcher_gpu.py "tests/test_full_size_gpu.py::test_c4_full_size_bm25_bit_exact" "tests/test_full_size_gpu.py::test_c4_full_size_hybrid_bit_exact" -m gpu -x -q -p no:cacheprovider ) > $O/pytest.log 2>&1; tail -5 $O/pytest.log
timeout 300 python scripts/bench_hybrid.py --steps 200 --warmup 10 > $O/bench_c4.json 2> $O/bench_c4.err; python -c "
import json; d=json.load(open('$O/bench_c4.json')); print('hybrid QPS', round(d['value'],1), '| bm25_only', {k: (round(v,4) if isinstance(v,float) else v) for k,v in d['bm25_only'].items()})"; tail -3 $O/bench_c4.err
ORAMA_NO_FUSED_SELECT=1 timeout 300 python scripts/bench_hybrid.py --steps 200 --warmup 10 --no-check > $O/bench_c4_nofused.json 2>/dev/null; python -c "
import json; d=json.load(open('$O/bench_c4_nofused.json')); print('NO_FUSED bm25_only', {k: (round(v,4) if isinstance(v,float) else v) for k,v in d['bm25_only'].items()})"
