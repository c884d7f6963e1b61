#!/bin/bash
set +e
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R=$PWD; O=$R/gpurun_out/k1b_pmc; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp; cd /tmp
for C in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_VMEM_RD" "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum" "GRBM_GUI_ACTIVE SQ_INSTS_SALU SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU"; do
  tag=$(echo $C | cut -d' ' -f1)
  timeout 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $O/$tag -o p -- python $R/scripts/k1b_probe.py > $O/$tag.log 2>&1
done
cd $R
python - <<'PY'
import csv, glob, statistics, collections
for tag in sorted(glob.glob("gpurun_out/k1b_pmc/*/")):
    vals = collections.defaultdict(lambda: collections.defaultdict(list))
    for path in glob.glob(tag + "*counter_collection.csv"):
        for row in csv.DictReader(open(path)):
            if "vec_scan_f32" not in row["Kernel_Name"]: continue
            d = int(row["End_Timestamp"]) - int(row["Start_Timestamp"])
            if d < 3_000_000: continue
            name = "K1" if "multi" not in row["Kernel_Name"] else ("K1b_QB8" if "4, 0, 8>" in row["Kernel_Name"] or ", 8>" in row["Kernel_Name"] else "K1b_QB4")
            vals[name][row["Counter_Name"]].append(float(row["Counter_Value"])); vals[name]["dur_us"].append(d/1e3)
    for name, v in vals.items():
        print(tag.split("/")[-2], name, {k: round(statistics.median(x)) for k, x in v.items()})
PY
