#!/bin/bash
# Round-4 measurement set: whole GPU suite, the driver's bench command, rocprofv3 kernel traces of every bench leg (NS, C2, C3,
# C5 shard, C4) with the clocks the same runs sampled, PMC passes for the K3r launches, the 1-rank exchange path.
set +e
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R=$PWD
O=$R/gpurun_out/r04final
rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
echo "== pytest -m gpu"; timeout 2400 python -m pytest tests -q -m gpu -p no:cacheprovider > $O/pytest_gpu_full.log 2>&1; grep -E "passed|failed|error" $O/pytest_gpu_full.log | tail -3 | tee $O/pytest_gpu.log
echo "== bench (driver command)"; timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench_all.json 2> $O/bench_all.err; tail -c 300 $O/bench_all.json; echo
echo "== bench --force-exchange (the N-rank code path on one rank)"; timeout 300 python bench.py --gpus 1 --force-exchange --steps 20 --warmup 5 --no-cpu-baseline --no-two-stage --configs none --no-pmc > $O/bench_force_exchange.json 2>$O/bench_force_exchange.err; tail -c 200 $O/bench_force_exchange.json; echo
cd /tmp
prof() {  # name, bench args...
  local W=$1; shift
  echo "== rocprof $W"; timeout 400 rocprofv3 --kernel-trace --stats -d $O/prof_$W -o $W -- python $R/bench.py "$@" > $O/rocprof_$W.log 2>&1; echo rc=$?
}
prof ns --steps 20 --warmup 3 --no-cpu-baseline --no-two-stage --configs none --no-pmc
prof c2 --workload c2 --steps 200 --warmup 10 --no-cpu-baseline --no-two-stage --configs none --no-pmc
prof c3 --workload c3 --steps 10 --warmup 2 --no-cpu-baseline --configs none --no-pmc
prof c5 --workload c5 --rows 10000000 --steps 10 --warmup 2 --no-cpu-baseline --configs none --no-pmc
prof c4 --steps 5 --warmup 2 --no-cpu-baseline --no-two-stage --configs c4 --no-pmc
echo "== rocprof k3r"; timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof_k3r -o k3r -- python $R/scripts/k3r_chunk_probe.py > $O/rocprof_k3r.log 2>&1; echo rc=$?
for C in FETCH_SIZE WRITE_SIZE; do
  echo "== pmc k3r $C"; timeout 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $O/pmc_k3r/$C -o p -- python $R/scripts/k3r_chunk_probe.py > $O/pmc_k3r_$C.log 2>&1
done
cd $R
for W in ns c2 c3 c5 c4 k3r; do
  python scripts/rocpd_summary.py $(find $O/prof_$W -name "*results.db" | head -1) > $O/${W}_kernel_stats.md 2>$O/${W}_kernel_stats.err
  # the clocks / power the same run sampled during its timed region (bench.py prints them in its JSON line)
  python - "$O/rocprof_$W.log" >> $O/${W}_kernel_stats.md 2>/dev/null <<'PY'
import json, sys
for line in open(sys.argv[1]):
    line = line.strip()
    if line.startswith("{") and '"roofline"' in line:
        d = json.loads(line)
        c = d["roofline"].get("clocks_during_timed_region", {})
        print(f"\nclocks during the timed region of this (profiled) run: {json.dumps(c)}; ms_per_step {d['ms_per_step']:.4f}, "
              f"HIP-event average launch {d['roofline']['avg_launch_ms']:.4f} ms")
PY
done
python scripts/pmc_summary.py $O/pmc_k3r range_score_kernel 151000000 mean > $O/pmc_k3r_range_score.json 2>$O/pmc_k3r.err
python scripts/pmc_summary.py $O/pmc_k3r keys_reduce_kernel 151000000 mean > $O/pmc_k3r_keys_reduce.json 2>>$O/pmc_k3r.err
find $O -name "*.csv" -size +2M -delete; find $O -name "*.db" -delete
du -sh $O; for W in ns c2 c3 c5 c4 k3r; do echo "-- $W"; head -8 $O/${W}_kernel_stats.md | cut -c1-180; tail -2 $O/${W}_kernel_stats.md | cut -c1-300; done
