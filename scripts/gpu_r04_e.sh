#!/bin/bash
# Round 4, GPU call: filtered top-k of the range scorer (sample ranges first, the rest emit only keys above the sample's k-th).
set +e
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R=$PWD
O=$R/gpurun_out/r04e
rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
echo "== pytest K3r-facing + sharded tests"
timeout 1200 python -m pytest tests/test_bm25_ranges_gpu.py tests/test_fulltext_gpu.py tests/test_random_gpu.py tests/test_sharded_fulltext_gpu.py tests/test_batcher_gpu.py tests/test_full_size_gpu.py -q -m gpu -p no:cacheprovider -x 2>&1 | tail -15 | tee $O/pytest_k3r.log
echo "== k3r A/B, filter on"
timeout 600 python scripts/k3r_ab.py 2>&1 | tail -16 | tee $O/k3r_ab_filter_on.log
echo "== k3r A/B, filter off"
ORAMA_K3R_FILTER=0 timeout 600 python scripts/k3r_ab.py 2>&1 | head -6 | tee $O/k3r_ab_filter_off.log
echo "== chunk probe on/off"
timeout 200 python scripts/k3r_chunk_probe.py 2>&1 | tail -1 | tee $O/probe_on.log
ORAMA_K3R_FILTER=0 timeout 200 python scripts/k3r_chunk_probe.py 2>&1 | tail -1 | tee $O/probe_off.log
echo "== serving"
timeout 300 scripts/native/bench_serving bm25 2>&1 | tail -5 | tee $O/serving_bm25.log
du -sh $O
