#!/bin/bash
set +e
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R=$PWD
rm -rf gpurun_out/pmc_cmp; mkdir -p gpurun_out/pmc_cmp
cd /tmp
for K in f16 f32; do
for C in "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum" "GRBM_GUI_ACTIVE" "TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_RD SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES"; do
  tag=${K}_$(echo $C | cut -d' ' -f1)
  timeout 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $R/gpurun_out/pmc_cmp/$tag -o p -- python $R/scripts/${K}_scan_only.py > $R/gpurun_out/pmc_cmp/$tag.log 2>&1
done
done
cd $R
python - <<'PY'
import csv, glob, statistics, collections
for tag in sorted(glob.glob("gpurun_out/pmc_cmp/*/")):
    vals = collections.defaultdict(list); dur=[]
    for path in glob.glob(tag + "*counter_collection.csv"):
        for row in csv.DictReader(open(path)):
            if "vec_scan" not in row["Kernel_Name"]: continue
            d = int(row["End_Timestamp"]) - int(row["Start_Timestamp"])
            if d < 1_000_000: continue
            vals[row["Counter_Name"]].append(float(row["Counter_Value"])); dur.append(d)
    print(tag.split("/")[-2], "dur_us", round(statistics.median(dur)/1e3) if dur else None, {k: round(statistics.median(v)) for k, v in vals.items()})
PY
tail -2 gpurun_out/pmc_cmp/f16_TCP_TCC_READ_REQ_LATENCY_sum.log | cut -c1-200
