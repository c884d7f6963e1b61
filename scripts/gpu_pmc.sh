#!/bin/bash
# PMC passes for the K1 scan (separate runs, kernel-trace only — no sys/runtime tracing with --pmc).
set +e
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R=$PWD
mkdir -p gpurun_out/pmc
export TMPDIR=/tmp
cd /tmp
for C in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES" "GRBM_GUI_ACTIVE"; do
  tag=$(echo $C | tr ' ' '_')
  echo "== pmc $C"
  timeout 600 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $R/gpurun_out/pmc/$tag -o p -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $R/gpurun_out/pmc/$tag.log 2>&1
  echo "rc=$?"
done
cd $R
find gpurun_out/pmc -name "*.csv" | head -30
for f in $(find gpurun_out/pmc -name "*counter_collection.csv"); do echo "--- $f"; head -3 $f | cut -c1-400; grep vec_scan $f | head -3 | cut -c1-400; done
