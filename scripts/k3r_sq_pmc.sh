#!/bin/bash
# SQ counters of K3r's scoring launch (one set of launches over 32 C4-shaped queries): where do the wave cycles go?
#   comparison flavour (ORAMA_COMPARISON_KERNELS=1), both bodies of the plain batch:
#   KERNEL=range_score_fast_kernel K3R_FAST=1 OUT=k3r_sq_fast scripts/k3r_sq_pmc.sh ; KERNEL=range_score_compact_kernel K3R_FAST=0 OUT=k3r_sq_r05 scripts/k3r_sq_pmc.sh
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R=$PWD
O=$R/gpurun_out/${OUT:-k3r_sq}
rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
cd /tmp
for SET in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_BRANCH" \
           "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_ANY SQ_INST_LEVEL_LDS" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INST_CYCLES_VMEM_RD SQ_INST_LEVEL_VMEM SQ_LEVEL_WAVES"; do
  T=$(echo $SET | tr ' ' '_' | cut -c1-40)
  timeout 200 rocprofv3 --pmc $SET --kernel-trace --output-format csv -d $O/$T -o p -- python $R/scripts/k3r_chunk_probe.py > $O/$T.log 2>&1
done
cd $R
python - <<'PY'
import csv, glob, statistics, collections, os
KERNEL = os.environ.get("KERNEL", "range_score_kernel")
vals = collections.defaultdict(list)
for path in glob.glob("gpurun_out/" + os.environ.get("OUT", "k3r_sq") + "/*/*counter_collection.csv"):
    for row in csv.DictReader(open(path, newline="")):
        if KERNEL in row["Kernel_Name"]:
            vals[row["Counter_Name"]].append(float(row["Counter_Value"]))
for k in sorted(vals):
    print(f"{k:28s} {statistics.median(vals[k]):16.0f}   (launches {len(vals[k])})")
PY
find $O -name "*.csv" -size +1M -delete
