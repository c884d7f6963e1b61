#!/bin/bash
set -x
mkdir -p gpurun_out/r02c6
cd $GRAFT_REPO_ROOT
O=gpurun_out/r02c6
timeout 300 python scripts/k2d_trace.py > $O/k2d_trace.log 2>&1; cat $O/k2d_trace.log
export TMPDIR=/tmp; R=$PWD; cd /tmp
i=0
for C in "FETCH_SIZE" "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum" "TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  MODES=1,3 timeout 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $R/$O/pmc$i -o p -- python $R/scripts/k2d_probe.py > $R/$O/pmc$i.log 2>&1
done
cd $R
python - <<'PY'
import csv, glob, collections, statistics, re
vals = collections.defaultdict(lambda: collections.defaultdict(list))
for path in glob.glob("gpurun_out/r02c6/pmc*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(path)):
        kn = row["Kernel_Name"]
        if "f16_wide" in kn: key = "K2c"
        elif "f16_pc" in kn: key = "K2d"
        else: continue
        d = int(row["End_Timestamp"]) - int(row["Start_Timestamp"])
        if d < 300_000: continue
        vals[key][row["Counter_Name"]].append(float(row["Counter_Value"]))
        vals[key]["dur_us"].append(d / 1e3)
for k, v in sorted(vals.items()):
    print(k, {c: (round(statistics.fmean(x)), len(x)) for c, x in sorted(v.items())})
PY
