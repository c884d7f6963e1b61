#!/bin/bash
set -x
mkdir -p gpurun_out/r02c3
cd $GRAFT_REPO_ROOT
O=gpurun_out/r02c3
# power hypothesis: the same kernels on a corpus of constant rows and all-(almost)-zero queries
MODES=1,3,6 CONST_ROWS=1 ZERO_Q=1 timeout 300 python scripts/k2d_probe.py > $O/k2d_const_rows_zero_q.log 2>&1; cat $O/k2d_const_rows_zero_q.log
MODES=1,3,6 timeout 300 python scripts/k2d_probe.py > $O/k2d_random.log 2>&1; cat $O/k2d_random.log
# ablations of geometry 2 (mode 3)
for d in 9 13 1 2 4; do MODES=3 DBG=$d timeout 200 python scripts/k2d_probe.py 2>&1 | sed "s/^/DBG=$d /" >> $O/k2d_ablate_geom2.log; done; cat $O/k2d_ablate_geom2.log
# shader clock under each variant of K2c: GRBM_GUI_ACTIVE / duration
export TMPDIR=/tmp; R=$PWD; cd /tmp
DBGS=0,9,2,4 timeout 400 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $R/$O/pmc_clk -o p -- python $R/scripts/k2c_ablate.py > $R/$O/pmc_clk.log 2>&1
cd $R
python - <<'PY'
import csv, glob, collections, statistics
vals = collections.defaultdict(lambda: collections.defaultdict(list))
for path in glob.glob("gpurun_out/r02c3/pmc_clk/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(path)):
        if "f16_wide" not in row["Kernel_Name"]: continue
        d = int(row["End_Timestamp"]) - int(row["Start_Timestamp"])
        if d < 300_000: continue
        import re
        m = re.search(r"Li4ELi3ELi(\d+)E", row["Kernel_Name"]) or re.search(r"<4, 3, (\d+)>", row["Kernel_Name"])
        key = m.group(1) if m else row["Kernel_Name"][-40:]
        vals[key][row["Counter_Name"]].append(float(row["Counter_Value"]) / d)  # per ns
for k, v in sorted(vals.items()):
    print("K2c DBG", k, {c: round(statistics.median(x), 3) for c, x in v.items()}, "(counter per ns; GRBM_GUI_ACTIVE per ns = GHz x XCDs?)")
PY
