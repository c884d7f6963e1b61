#!/bin/bash
# Round-5 GPU calls, one script with a step list: scripts/gpu_r05.sh <out-name> step [step ...]
#   probe     scripts/sampler_probe.py (is the sampler on the right GPU, does it agree with rocm-smi)
#   bench     the driver's command (python bench.py --steps 20 --warmup 5)
#   scale     scripts/scale_dry_run.py (bench.py --gpus 2/4/8 over the loopback transport + uncontended rank steps)
#   tests     the whole GPU suite            tests:<expr>  pytest -k <expr>
#   prof      rocprofv3 kernel traces of every bench leg + the K3r chunk probe + PMC passes of the K3r launches
#   profall   rocprofv3 --kernel-trace --stats of `python bench.py --steps 20 --warmup 5 --no-pmc` (every leg in one trace)
#   pmc       FETCH_SIZE / WRITE_SIZE passes over the scan kernels of C2 / C3 / C5 shard / NS (total traffic per step)
#   py:<path> any script under scripts/ (arguments after a colon, comma separated)
set +e
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R=$PWD
NAME=$1; shift
O=$R/gpurun_out/$NAME
rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
prof() {  # name, bench args...
  local W=$1; shift
  echo "== rocprof $W"; (cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats -d $O/prof_$W -o $W -- python $R/bench.py "$@" > $O/rocprof_$W.log 2>&1); echo rc=$?
  python scripts/rocpd_summary.py $(find $O/prof_$W -name "*results.db" | head -1) > $O/${W}_kernel_stats.md 2>$O/${W}_kernel_stats.err
  python - "$O/rocprof_$W.log" >> $O/${W}_kernel_stats.md 2>/dev/null <<'PY'
import json, sys
for line in open(sys.argv[1]):
    line = line.strip()
    if line.startswith("{") and '"roofline"' in line:
        d = json.loads(line)
        rf = d["roofline"]
        c = rf.get("clocks_during_timed_region", {})
        c = {k: c.get(k) for k in ("available", "samples", "sclk_mhz_median", "power_w_mean", "power_w_from_energy_counter",
                                   "ppt_throttle_residency_pct", "pci_bus_id")}
        print(f"\nclocks during the timed region of this (profiled) run: {json.dumps(c)}; ms_per_step {d['ms_per_step']:.4f}, "
              f"HIP events: median scan per step {rf.get('median_scan_ms_per_step')} ms over {rf.get('steps_in_median')} steps, "
              f"average launch {rf['avg_launch_ms']:.4f} ms")
PY
}
for STEP in "$@"; do
  case $STEP in
    probe) echo "== sampler probe"; timeout 300 python scripts/sampler_probe.py > $O/sampler_probe.log 2>&1; tail -25 $O/sampler_probe.log ;;
    bench) echo "== bench (driver command)"; timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench_all.json 2> $O/bench_all.err; tail -c 400 $O/bench_all.json; echo; tail -3 $O/bench_all.err ;;
    scale) echo "== scale dry run"; timeout 1500 python scripts/scale_dry_run.py --out $O/scale_dry_run.json > $O/scale_dry_run.log 2>&1; tail -40 $O/scale_dry_run.log ;;
    tests) echo "== pytest -m gpu"; timeout 2400 python -m pytest tests -q -m gpu -p no:cacheprovider > $O/pytest_gpu_full.log 2>&1; grep -E "passed|failed|error" $O/pytest_gpu_full.log | tail -5 | tee $O/pytest_gpu.log ;;
    tests:*) K=${STEP#tests:}; echo "== pytest -m gpu -k $K"; timeout 1500 python -m pytest tests -q -m gpu -p no:cacheprovider -x -k "$K" > $O/pytest_gpu_k.log 2>&1; tail -15 $O/pytest_gpu_k.log ;;
    prof)
      prof ns --steps 50 --warmup 3 --no-cpu-baseline --no-two-stage --configs none --no-pmc
      prof c2 --workload c2 --steps 200 --warmup 10 --no-cpu-baseline --no-two-stage --configs none --no-pmc
      prof c3 --workload c3 --steps 50 --warmup 2 --no-cpu-baseline --configs none --no-pmc
      prof c5 --workload c5 --rows 10000000 --steps 50 --warmup 2 --no-cpu-baseline --configs none --no-pmc
      prof c4 --steps 5 --warmup 2 --no-cpu-baseline --no-two-stage --configs c4 --no-pmc
      echo "== rocprof k3r"; (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof_k3r -o k3r -- python $R/scripts/k3r_chunk_probe.py > $O/rocprof_k3r.log 2>&1); echo rc=$?
      python scripts/rocpd_summary.py $(find $O/prof_k3r -name "*results.db" | head -1) > $O/k3r_kernel_stats.md 2>$O/k3r_kernel_stats.err
      for C in FETCH_SIZE WRITE_SIZE; do
        echo "== pmc k3r $C"; (cd /tmp && timeout 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $O/pmc_k3r/$C -o p -- python $R/scripts/k3r_chunk_probe.py > $O/pmc_k3r_$C.log 2>&1)
      done
      python scripts/pmc_summary.py $O/pmc_k3r range_score_ 151000000 mean > $O/pmc_k3r_range_score.json 2>$O/pmc_k3r.err
      python scripts/pmc_summary.py $O/pmc_k3r keys_reduce_kernel 151000000 mean > $O/pmc_k3r_keys_reduce.json 2>>$O/pmc_k3r.err
      python scripts/pmc_summary.py $O/pmc_k3r keys_final_kernel 151000000 mean > $O/pmc_k3r_keys_final.json 2>>$O/pmc_k3r.err
      for W in ns c2 c3 c5 c4 k3r; do echo "-- $W"; head -8 $O/${W}_kernel_stats.md | cut -c1-180; tail -2 $O/${W}_kernel_stats.md | cut -c1-400; done ;;
    profall)  # the DRIVER'S command itself under the kernel trace (its nested --pmc passes off: rocprofv3 does not nest)
      prof driver_command --steps 20 --warmup 5 --no-pmc
      head -14 $O/driver_command_kernel_stats.md | cut -c1-200; tail -2 $O/driver_command_kernel_stats.md | cut -c1-400 ;;
    pmc)  # HBM traffic of the scan kernels of C2 / C3 / C5-shard / NS: total over a fixed number of steps (scripts/pmc_scan_probe.py)
      for W in c2 c3 c5 ns; do
        for C in FETCH_SIZE WRITE_SIZE; do
          (cd /tmp && timeout 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $O/pmc_$W/$C -o p -- python $R/scripts/pmc_scan_probe.py $W 6 > $O/pmc_${W}_$C.log 2>&1)
        done
        A=$(grep -o '"alg_bytes_per_step": [0-9]*' $O/pmc_${W}_FETCH_SIZE.log | grep -o '[0-9]*$')
        python scripts/pmc_total.py $O/pmc_$W vec_scan $A 6 > $O/pmc_${W}_vec_scan.json 2>> $O/pmc_scans.err
        echo "-- $W"; grep -E "traffic_over|launches|FETCH|WRITE" $O/pmc_${W}_vec_scan.json | tr -d '\n'; echo
      done ;;
    c2) echo "== bench c2"; timeout 600 python bench.py --workload c2 --steps 200 --warmup 10 --no-cpu-baseline --no-two-stage --configs none --no-pmc > $O/bench_c2.json 2> $O/bench_c2.err; python - $O/bench_c2.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print({k: d[k] for k in ("value", "ms_per_step", "latency_ms_p50", "latency_ms_p95", "latency_ms_p50_host_api")}, d["roofline"]["frac"], d["step_breakdown_us"])
PY
      ;;
    c4) for T in 1 0; do echo "== bench c4, ORAMA_HYBRID_DEVICE_TAIL=$T"; ORAMA_HYBRID_DEVICE_TAIL=$T timeout 600 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --configs c4 --no-pmc > $O/bench_c4_tail$T.json 2> $O/bench_c4_tail$T.err; python - $O/bench_c4_tail$T.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
c = d["configs"]["c4"]
print("ns", round(d["value"], 2), "| c4", round(c["value"], 2), "QPS, p50", round(c["latency_ms_p50"], 4), "ms | shadow", round(c["shadow_store"]["value"], 2),
      "QPS, p50", round(c["shadow_store"]["latency_ms_p50"], 4), "| two-stage host API", round(d["two_stage_exact"]["value"], 1), "session", round(d["two_stage_exact"]["session"]["value"], 1))
PY
      done ;;
    py:*) A=${STEP#py:}; S=${A%%:*}; ARGS=""; [ "$A" != "$S" ] && ARGS=$(echo "${A#*:}" | tr ',' ' '); B=$(basename $S .py)
      echo "== python $S $ARGS"; ORAMA_K3R_STATS=1 ORAMA_K3R_DBG=${K3R_DBG:-0} timeout 1200 python $S $ARGS > $O/$B.log 2>&1; echo rc=$?; tail -60 $O/$B.log ;;
  esac
done
find $O -name "*.csv" -size +2M -delete; find $O -name "*.db" -delete
du -sh $O
