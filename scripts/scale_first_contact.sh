#!/bin/bash
# First contact with an 8-GPU MI355X node: everything the scaling record needs, in one command.
#
#   scripts/scale_first_contact.sh [--loopback] [--out DIR] [--worlds "1 2 4 8"] [--steps K]
#
#   0. preflight   topology (rocm-smi --showtopo), RCCL the library will dlopen, devices visible, bench.py's own preflight per N
#   1. NS          the north-star job (10 M x 768 fp32, strong scaling) at N = 1 / 2 / 4 / 8: the compact metric line of each
#                  + the details file; ranks_seen == N is required
#   2. C5          BASELINE configs[4] (80 M x 768 fp16 over 8 ranks, 256 queries per batch) at N = 8
#   3. identity    every rank's dump of the last step (--dump-result) holds the same global answer, bit for bit
#   4. table       QPS / p50 / scan / all-gather / merge per N beside the bounds profiles/r05_scale_dry_run.json projected
#                  (lower = tail chain fully exposed, upper = scan-bound) -> <out>/SCALE_table.md and SCALE.json
#   5. fallbacks   what to flip if a figure falls under its lower bound (printed with the table)
#
# --loopback: the same job over tests/mock_rccl's shared-memory transport with every rank on GPU 0 (this is what runs on the 1-GPU
# boxes of the build pool and in CI: it proves the plumbing — launcher, communicator, all-gather + K6, dumps, table — not xGMI).
set -u
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
R=$PWD
LOOPBACK=0; OUT=$R/gpurun_out/scale_first_contact; WORLDS="1 2 4 8"; STEPS=50
while [ $# -gt 0 ]; do
  case $1 in
    --loopback) LOOPBACK=1 ;;
    --out) OUT=$2; shift ;;
    --worlds) WORLDS=$2; shift ;;
    --steps) STEPS=$2; shift ;;
    *) echo "unknown argument $1"; exit 2 ;;
  esac; shift
done
mkdir -p "$OUT"; rm -f "$OUT"/*.json "$OUT"/*.npz "$OUT"/*.log "$OUT"/*.md 2>/dev/null
export HSA_ENABLE_IPC_MODE_LEGACY=0
for v in RANK LOCAL_RANK WORLD_SIZE MASTER_ADDR MASTER_PORT; do unset $v; done
if [ $LOOPBACK = 1 ]; then
  [ -f tests/mock_rccl/libmock_rccl.so ] || make -C tests/mock_rccl > /dev/null
  export ORAMA_RCCL_LIB=$R/tests/mock_rccl/libmock_rccl.so
fi

echo "== 0. preflight" | tee "$OUT/preflight.log"
{
  echo "-- devices"; python - <<'PY'
import oramacore_amd as oa
from oramacore_amd import _native as N
import ctypes as C
lib = N.load(); n = C.c_int()
print("orama_device_count:", lib.orama_device_count(C.byref(n)), n.value)
for d in range(n.value):
    with oa.Context(d) as c:
        i = c.device_info(); print(d, i["name"], f"{i['hbm_bytes'] / 2**30:.0f} GiB", c.pci_bus_id())
PY
  echo "-- topology (rocm-smi --showtopo)"; (rocm-smi --showtopo 2>&1 || echo "rocm-smi not usable here") | head -60
  echo "-- collective library"; echo "ORAMA_RCCL_LIB=${ORAMA_RCCL_LIB:-<unset: librccl.so from the loader path>}"
  (ls -l /opt/rocm/lib/librccl.so* 2>/dev/null | head -3; strings /opt/rocm/lib/librccl.so 2>/dev/null | grep -m1 "RCCL version" ) || true
} 2>&1 | tee -a "$OUT/preflight.log"

run_job() {  # tag, N, bench args...
  local TAG=$1 N=$2; shift 2
  echo "== $TAG at N = $N"
  timeout 1500 python bench.py --gpus $N --steps $STEPS --warmup 5 --no-cpu-baseline --no-pmc --configs none --no-two-stage \
      --details-file "$OUT/${TAG}_n$N.details.json" --dump-result "$OUT/${TAG}_n$N" "$@" > "$OUT/${TAG}_n$N.out" 2> "$OUT/${TAG}_n$N.err"
  local RC=$?
  tail -n 1 "$OUT/${TAG}_n$N.out" > "$OUT/${TAG}_n$N.line.json"
  echo "rc=$RC $(wc -c < "$OUT/${TAG}_n$N.line.json") bytes: $(cut -c1-220 "$OUT/${TAG}_n$N.line.json")"
  [ $RC = 0 ] || tail -5 "$OUT/${TAG}_n$N.err"
}
for N in $WORLDS; do run_job ns $N; done
case " $WORLDS " in *" 8 "*) run_job c5 8 --workload c5 --steps 10 ;; esac

python - "$OUT" "$LOOPBACK" <<'PY' | tee "$OUT/SCALE_table.md"
import glob, json, os, sys
import numpy as np
out, loopback = sys.argv[1], sys.argv[2] == "1"
bounds = {}
try:
    bounds = json.load(open("profiles/r05_scale_dry_run.json"))["projection"]
except Exception as e:  # noqa: BLE001
    print(f"(no projection file: {e})")
rows, record, ok = [], {"transport": "tests/mock_rccl loopback on ONE GPU" if loopback else "RCCL", "jobs": {}}, True
base = None
for path in sorted(glob.glob(os.path.join(out, "*.line.json")), key=lambda p: (os.path.basename(p).split("_n")[0], int(os.path.basename(p).split("_n")[1].split(".")[0]))):
    tag, n = os.path.basename(path).split("_n")[0], int(os.path.basename(path).split("_n")[1].split(".")[0])
    try:
        line = json.loads(open(path).read())
        assert len(open(path).read()) < 4096
        det = json.load(open(os.path.join(out, f"{tag}_n{n}.details.json")))
    except Exception as e:  # noqa: BLE001
        rows.append(f"| {tag} | {n} | FAILED: {e} | | | | | | | |"); ok = False; continue
    if "error" in line:
        rows.append(f"| {tag} | {n} | FAILED: {line['error'][:80]} | | | | | | | |"); ok = False; continue
    seen = line["config"]["ranks_seen"]
    # identity: every rank's dump holds the same global answer
    dumps = [np.load(f"{out}/{tag}_n{n}.rank{r}.npz") for r in range(n) if os.path.exists(f"{out}/{tag}_n{n}.rank{r}.npz")]
    same = len(dumps) == n and all(np.array_equal(d["ids"], dumps[0]["ids"]) and np.array_equal(d["dist"].view(np.uint32), dumps[0]["dist"].view(np.uint32))
                                   and np.array_equal(d["cnt"], dumps[0]["cnt"]) for d in dumps)
    ok = ok and seen == n and same
    bd = det.get("step_breakdown_us") or {}
    if tag == "ns" and n == 1:
        base = line["value"]
    b = bounds.get(str(n), {}) if tag == "ns" else {}
    lo, hi = b.get("qps_lower_bound_tail_exposed"), b.get("qps_upper_bound_scan_bound")
    verdict = "" if not lo or loopback else ("UNDER the lower bound" if line["value"] < 0.97 * lo else "within" if not hi or line["value"] <= 1.03 * hi else "above the upper bound")
    rows.append(f"| {tag} | {n} | {line['value']:.1f} | {line['value'] / base / n:.2f} | {line['latency_ms_p50']:.3f} | {bd.get('scan', 0) / 1e3:.3f} | "
                f"{(bd.get('all_gather') or 0):.0f} | {(bd.get('merge_k6') or 0):.0f} | {seen}/{n} ranks, dumps {'identical' if same else 'DIFFER'} | "
                f"{'' if not lo else f'{lo:.0f} .. {hi:.0f}'} {verdict} |" if base else f"| {tag} | {n} | {line['value']:.1f} | | {line['latency_ms_p50']:.3f} | {bd.get('scan', 0) / 1e3:.3f} | {(bd.get('all_gather') or 0):.0f} | {(bd.get('merge_k6') or 0):.0f} | {seen}/{n} ranks, dumps {'identical' if same else 'DIFFER'} | |")
    record["jobs"][f"{tag}_n{n}"] = {"line": line, "ranks_identical": bool(same), "step_breakdown_us": bd}
print(f"# SCALE — {'loopback transport, all ranks on one GPU (plumbing only)' if loopback else 'RCCL over xGMI'}\n")
print("| job | N | queries/s | efficiency vs N=1 | p50 ms | scan ms | all-gather us | merge us | ranks | projected bounds (r05 dry run) |")
print("|---|---|---|---|---|---|---|---|---|---|")
print("\n".join(rows))
print("""
Fallback matrix (what to flip when a figure falls UNDER its lower bound — each is one environment variable or option, no rebuild):
* the all-gather serialises behind a scan that fills every CU (all-gather us >> 100, step = scan + gather):
  the tail streams are created high-priority already — check `rocm-smi --showtopo` for a link that is not XGMI; then try
  ORAMA_SHARD_LANES=1 (one sharded call in flight per process) and `--streams 1` (no second tail stream competing for the SDMA/CP);
* ranks_seen < N or a rank's dump differs: a rank bound to the wrong device — HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES must not be set
  per rank (LOCAL_RANK picks the device); the preflight lists what the library sees;
* RCCL fails to initialise (hipIpcGetMemHandle: invalid argument): HSA_ENABLE_IPC_MODE_LEGACY=0 must be exported (this script does);
* scan ms at N ranks > (scan ms at N = 1) / N by more than 5 %: the shard is too small for the fused per-wave top-k's rule (>= 3 GB
  per rank): `Context.set_option("fused_topk", 1)` forces it; under 1.25 M rows per rank the launch is latency-bound — expected.
""")
json.dump(record, open(os.path.join(out, "SCALE.json"), "w"), indent=1)
print("RESULT:", "GREEN" if ok else "RED (see rows above)")
sys.exit(0 if ok else 1)
PY
