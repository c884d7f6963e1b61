#!/bin/bash
# Round-6 GPU calls, one script with a step list: scripts/gpu_r06.sh <out-name> step [step ...]
#   bench        the driver's command (python bench.py --steps 20 --warmup 5): compact line + bench_details.json
#   tests        the whole GPU suite            tests:<expr>  pytest -k <expr>
#   k1m          scripts/k1m_probe.py (fp32 MFMA batches on the north-star rows, clocks beside every line)
#   serving-f32  scripts/native/bench_serving vec on the plain fp32 store (direct calls and the request batcher)
#   serving      the full serving check (bm25 / hybrid / vec f32)
#   profall      rocprofv3 --kernel-trace --stats of the driver's command (nested PMC passes off)
#   py:<path>    any script under scripts/ (arguments after a colon, comma separated)
set +e
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R=$PWD
NAME=$1; shift
O=$R/gpurun_out/$NAME
rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
serving_build() {
  [ -x scripts/native/bench_serving ] || g++ -O2 -std=c++17 -I include scripts/native/bench_serving.cpp -L oramacore_amd/csrc -lorama_hip -pthread -Wl,-rpath,$R/oramacore_amd/csrc -o scripts/native/bench_serving
}
prof() {  # name, command...
  local W=$1; shift
  echo "== rocprof $W"; (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof_$W -o $W -- "$@" > $O/rocprof_$W.log 2>&1); echo rc=$?
  python scripts/rocpd_summary.py $(find $O/prof_$W -name "*results.db" | head -1) > $O/${W}_kernel_stats.md 2>$O/${W}_kernel_stats.err
  head -12 $O/${W}_kernel_stats.md | cut -c1-200
}
for STEP in "$@"; do
  case $STEP in
    bench) echo "== bench (driver command)"; S=$(date +%s); timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench.out 2> $O/bench.err; echo "rc=$? wall=$(( $(date +%s) - S )) s"; tail -n 1 $O/bench.out | wc -c; tail -n 1 $O/bench.out; cp bench_details.json $O/ 2>/dev/null ;;
    tests) echo "== pytest -m gpu"; timeout 2400 python -m pytest tests -q -m gpu -p no:cacheprovider > $O/pytest_gpu_full.log 2>&1; grep -E "passed|failed|error" $O/pytest_gpu_full.log | tail -5 | tee $O/pytest_gpu.log ;;
    tests:*) K=${STEP#tests:}; echo "== pytest -m gpu -k $K"; timeout 1500 python -m pytest tests -q -m gpu -p no:cacheprovider -x -k "$K" > $O/pytest_gpu_k.log 2>&1; tail -15 $O/pytest_gpu_k.log ;;
    k1m) echo "== k1m probe"; timeout 400 python scripts/k1m_probe.py --batches 1,8,9,16,32,64 2>&1 | tee $O/k1m_probe.log ;;
    serving-f32) serving_build; echo "== serving vec f32"; timeout 600 scripts/native/bench_serving vec 10000000 60 8,64,256 f32 2>&1 | tee $O/serving_vec_f32.log ;;
    serving) serving_build
      timeout 600 scripts/native/bench_serving bm25 10000000 300 1,8,32,128 2>&1 | tee $O/serving_bm25.log
      timeout 600 scripts/native/bench_serving hybrid 10000000 20 1,8,32,128 2>&1 | tee $O/serving_hybrid.log
      timeout 600 scripts/native/bench_serving vec 10000000 60 8,64,256 f32 2>&1 | tee $O/serving_vec_f32.log ;;
    profall) prof driver_command python $R/bench.py --steps 20 --warmup 5 --no-pmc ;;
    pmc)  # FETCH_SIZE / WRITE_SIZE passes (separate, kernel trace only) over the scan kernels: total traffic per step / algorithmic
      for W in nsb:vec_scan_f32_cvt nsb_mfma:vec_scan_f32_mfma ns:vec_scan_f32_kernel; do
        N=${W%%:*}; KERN=${W#*:}
        for C in FETCH_SIZE WRITE_SIZE; do
          echo "== pmc $N $C"; (cd /tmp && timeout 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $O/pmc_$N/$C -o p -- python $R/scripts/pmc_scan_probe.py $N 6 > $O/pmc_${N}_$C.log 2>&1); echo rc=$?
        done
        python scripts/pmc_total.py $O/pmc_$N $KERN 30720000000 6 > $O/pmc_${N}_vec_scan.json 2>$O/pmc_$N.err; grep -E "traffic_over|launches" $O/pmc_${N}_vec_scan.json
        find $O/pmc_$N -name "*.csv" -size +1M -delete
      done ;;
    py:*) A=${STEP#py:}; P=${A%%:*}; ARGS=""; [ "$A" != "$P" ] && ARGS=$(echo "${A#*:}" | tr ',' ' '); echo "== python $P $ARGS"; timeout 900 python $P $ARGS 2>&1 | tee $O/$(basename $P .py).log | tail -40 ;;
    *) echo "unknown step $STEP" ;;
  esac
done
