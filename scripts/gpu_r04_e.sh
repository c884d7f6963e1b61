#!/bin/bash
# Round 4, GPU call 8: top-k without histogram rounds (bound from the lanes' best keys + ranks by counting) — parity, batch
# rate, and the single-query chain at several range sizes.
set +e
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R=$PWD
O=$R/gpurun_out/r04e
rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
echo "== pytest -m gpu (top-k users)"
timeout 1500 python -m pytest tests/test_topn_gpu.py tests/test_vector_gpu.py tests/test_bm25_ranges_gpu.py tests/test_fulltext_gpu.py \
    tests/test_token_score_gpu.py tests/test_random_gpu.py tests/test_vector_f16_gpu.py tests/test_two_stage_gpu.py -q -m gpu -p no:cacheprovider -x 2>&1 | tail -12 | tee $O/pytest_gpu.log
echo "== k3r A/B"
timeout 600 python scripts/k3r_ab.py 2>&1 | tail -16 | tee $O/k3r_ab.log
echo "== single query, ranges of 1536 / 1024 / 768 / 512 postings"
for t in 1536 1024 768 512; do
  echo -n "target $t: "; ORAMA_K3R_TARGET=$t timeout 300 python scripts/k3r_single_probe.py 2>&1 | tail -1
done | tee $O/single_targets.log
echo "== single query trace"
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $O/single_trace -o single -- python $R/scripts/k3r_single_probe.py > $O/single_trace.log 2>&1
tail -2 $O/single_trace.log
find $O/single_trace -name "*kernel_stats*" | head -1 | xargs -I{} head -12 {}
du -sh $O
