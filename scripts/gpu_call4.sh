#!/bin/bash
set -x
mkdir -p gpurun_out/r02c4
cd $GRAFT_REPO_ROOT
O=gpurun_out/r02c4
timeout 200 python scripts/power_probe.py > $O/power_probe.log 2>&1; cat $O/power_probe.log
export TMPDIR=/tmp; R=$PWD; cd /tmp
i=0
for C in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_WAIT_INST_LDS" "SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_RD SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU"; do
  i=$((i+1))
  DBGS=0,2,1,4 timeout 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $R/$O/pmc$i -o p -- python $R/scripts/k2c_ablate.py > $R/$O/pmc$i.log 2>&1
done
cd $R
python - <<'PY'
import csv, glob, collections, statistics, re
vals = collections.defaultdict(lambda: collections.defaultdict(list))
for path in glob.glob("gpurun_out/r02c4/pmc*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(path)):
        if "f16_wide" not in row["Kernel_Name"]: continue
        d = int(row["End_Timestamp"]) - int(row["Start_Timestamp"])
        if d < 300_000: continue
        m = re.search(r"Li4ELi3ELi(\d+)E", row["Kernel_Name"]) or re.search(r"<4, 3, (\d+)>", row["Kernel_Name"])
        key = m.group(1) if m else row["Kernel_Name"][-40:]
        vals[key][row["Counter_Name"]].append(float(row["Counter_Value"]))
        vals[key]["dur_us"].append(d / 1e3)
for k, v in sorted(vals.items()):
    print("K2c DBG", k, {c: round(statistics.median(x)) for c, x in sorted(v.items())})
PY
