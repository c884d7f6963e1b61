#!/bin/bash
set +e
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R=$PWD
mkdir -p gpurun_out/pmc_f16
cd /tmp
for D in 0 1; do
for C in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_IFETCH SQ_IFETCH_LEVEL SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_RD" "SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS"; do
  tag=d${D}_$(echo $C | cut -d' ' -f1)
  ORAMA_F16_DEBUG=$D timeout 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $R/gpurun_out/pmc_f16/$tag -o p -- python $R/scripts/f16_scan_only.py > /dev/null 2>&1
done
done
cd $R
python - <<'PY'
import csv, glob, statistics, collections
for tag in sorted(glob.glob("gpurun_out/pmc_f16/*")):
    vals = collections.defaultdict(list); dur=[]
    for path in glob.glob(tag + "/*counter_collection.csv"):
        for row in csv.DictReader(open(path)):
            if "vec_scan_f16" not in row["Kernel_Name"]: continue
            if int(row["Grid_Size"]) < 100000: continue
            d = int(row["End_Timestamp"]) - int(row["Start_Timestamp"])
            if d < 1_000_000: continue   # main launch only
            vals[row["Counter_Name"]].append(float(row["Counter_Value"])); dur.append(d)
    print(tag.split("/")[-1], "dur_us", round(statistics.median(dur)/1e3) if dur else None, {k: round(statistics.median(v)) for k, v in vals.items()})
PY
