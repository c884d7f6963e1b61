#!/bin/bash
# Round 4, GPU call: (value, index) lists and the dense heads of the fp16 scans through the register-streaming pairs_reduce —
# parity of everything that selects, then the fp16 legs of the bench.
set +e
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R=$PWD
O=$R/gpurun_out/r04g
rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
echo "== pytest -m gpu (selection users)"
timeout 1500 python -m pytest tests/test_topn_gpu.py tests/test_vector_gpu.py tests/test_vector_f16_gpu.py tests/test_two_stage_gpu.py tests/test_random_gpu.py \
    tests/test_bm25_ranges_gpu.py tests/test_limits_gpu.py tests/test_stress_gpu.py -q -m gpu -p no:cacheprovider -x 2>&1 | tail -8 | tee $O/pytest_gpu.log
echo "== key-list kernels by phase"
scripts/micro/keys_reduce_probe 2>&1 | tee $O/keys_reduce_probe.log
echo "== bench c3 / c5 shard"
timeout 600 python bench.py --workload c3 --steps 20 --warmup 5 --no-preflight 2>$O/c3.err | tail -1 > $O/c3.json
timeout 600 python bench.py --workload c5 --steps 20 --warmup 5 --no-preflight 2>$O/c5.err | tail -1 > $O/c5.json
python - <<'PY'
import json
for n in ("c3","c5"):
    try:
        d=json.loads(open(f"gpurun_out/r04g/{n}.json").read())
        print(n, round(d["value"],1), d["unit"], "ms/step", round(d["ms_per_step"],3), "topk ms/step", round(d["roofline"]["topk_select_ms_per_step"],3), "scan avg ms", round(d["roofline"]["avg_launch_ms"],3), "launches/step", d["roofline"]["scan_launches_per_step"])
    except Exception as e:
        print(n, "failed", e)
PY
tail -3 $O/c3.err $O/c5.err
