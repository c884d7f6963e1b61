#!/bin/bash
# SQ / GRBM counters of K1m's filter launch (32 queries over 10M x 768 fp32): how busy is the matrix pipe, what do the waves wait for,
# what clock did the launch run at (GRBM_GUI_ACTIVE summed over the 8 XCDs / 8 / wall time).  Separate passes, kernel trace only.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R=$PWD
O=$R/gpurun_out/k1m_sq
rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep -o "SQ_[A-Z_]*MFMA[A-Z_]*\|SQ_VALU_MFMA[A-Z_]*\|SQ_INSTS_MFMA\|SQ_INSTS_VALU_MFMA[A-Z_0-9]*" | sort -u > $O/mfma_counters.txt
echo "MFMA counters on this box:"; cat $O/mfma_counters.txt
cd /tmp
for SET in "GRBM_GUI_ACTIVE" \
           "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_MFMA SQ_INSTS_VALU_MFMA_MOPS_F32" \
           "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_ANY" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INST_CYCLES_VMEM_RD SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS"; do
  T=$(echo $SET | tr ' ' '_' | cut -c1-40)
  timeout 200 rocprofv3 --pmc $SET --kernel-trace --output-format csv -d $O/$T -o p -- python $R/scripts/k1m_probe.py --batches 32 --reps 3 > $O/$T.log 2>&1 || echo "pass $T failed: $(tail -2 $O/$T.log)"
done
cd $R
python - <<'PY'
import csv, glob, statistics, collections
vals = collections.defaultdict(list); dur = []
for path in glob.glob("gpurun_out/k1m_sq/*/*counter_collection.csv"):
    for row in csv.DictReader(open(path, newline="")):
        if "vec_scan_f32_mfma_kernel<false>" in row["Kernel_Name"]:
            vals[row["Counter_Name"]].append(float(row["Counter_Value"]))
            if row["Counter_Name"] == "GRBM_GUI_ACTIVE":
                dur.append((float(row["Counter_Value"]), int(row["End_Timestamp"]) - int(row["Start_Timestamp"])))
for k in sorted(vals):
    print(f"{k:32s} {statistics.median(vals[k]):18.0f}   (launches {len(vals[k])})")
if dur:
    print("effective clock of the (profiled) filter launches: " + ", ".join(f"{c / 8 / ns * 1e3:.0f} MHz over {ns / 1e6:.3f} ms" for c, ns in dur))
PY
find $O -name "*.csv" -size +1M -delete
