#!/bin/bash
# Round 4, GPU call 12: the key-list top-k with bounds from registers, ranks by counting and one atomic per wave — whole suite,
# A/B against the merge tree, serving, the single-query chain.
set +e
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R=$PWD
O=$R/gpurun_out/r04f
rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
echo "== pytest -m gpu (all)"
timeout 2400 python -m pytest tests -q -m gpu -p no:cacheprovider -x 2>&1 | tail -8 | tee $O/pytest_gpu.log
echo "== k3r A/B"
timeout 600 python scripts/k3r_ab.py 2>&1 | tail -16 | tee $O/k3r_ab.log
echo "== key-list kernels by phase"
scripts/micro/keys_reduce_probe 2>&1 | tee $O/keys_reduce_probe.log
echo "== native serving bm25"
timeout 300 scripts/native/bench_serving bm25 2>&1 | tail -9 | tee $O/serving_bm25.log
ORAMA_POST_CALL_TRACE=1 timeout 300 scripts/native/bench_serving bm25 10000000 6000 1 2>&1 | head -4 | tee $O/post_call_trace.log
echo "== single query trace"
cd /tmp
timeout 300 rocprofv3 --kernel-trace --memory-copy-trace -d $O/single_trace -o single -- python $R/scripts/k3r_single_probe.py > $O/single_trace.log 2>&1
cd $R
python scripts/rocpd_gaps.py $O/single_trace/single_results.db 2>&1 | tail -20 | tee $O/single_timeline.log
rm -rf $O/single_trace
du -sh $O
