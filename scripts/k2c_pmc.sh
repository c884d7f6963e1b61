#!/bin/bash
set +e
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R=$PWD; O=$R/gpurun_out/k2c_pmc; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp; cd /tmp
for C in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_WAIT_INST_LDS" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_IDX_ACTIVE" "GRBM_GUI_ACTIVE TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum" "SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_INSTS_VMEM_RD"; do
  tag=$(echo $C | cut -d' ' -f1)
  NQ=256 timeout 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $O/$tag -o p -- python $R/scripts/k2c_probe.py > $O/$tag.log 2>&1
done
cd $R
python - <<'PY'
import csv, glob, statistics, collections
for tag in sorted(glob.glob("gpurun_out/k2c_pmc/*/")):
    vals = collections.defaultdict(list)
    for path in glob.glob(tag + "*counter_collection.csv"):
        for row in csv.DictReader(open(path)):
            if "f16_wide" not in row["Kernel_Name"]: continue
            d = int(row["End_Timestamp"]) - int(row["Start_Timestamp"])
            if d < 1_000_000: continue
            vals[row["Counter_Name"]].append(float(row["Counter_Value"])); vals["dur_us"].append(d/1e3)
    print(tag.split("/")[-2], {k: round(statistics.median(x)) for k, x in vals.items()})
PY
tail -2 $O/SQ_WAVE_CYCLES.log
