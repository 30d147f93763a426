#!/bin/bash
# SQ counters of the fused conv1 -> conv2 kernel (one rocprofv3 --pmc pass over tools/conv12_bench.py)
cd "$GRAFT_REPO_ROOT" || exit 1
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/c12pmc; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS SQ_INSTS_VALU GRBM_GUI_ACTIVE --output-format csv -d $O/a -o p -- python3 $R/tools/conv12_bench.py --iters 3 > $O/a.log 2>&1 || exit 3
cd $R
python3 - <<'PY'
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for path in glob.glob("gpurun_out/c12pmc/a/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(path)):
        n = r["Kernel_Name"]
        key = "conv12" if "conv12_kernel" in n else "conv256" if "conv256" in n else "conv1" if "conv1_mfma" in n else None
        if key: acc[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in acc.items():
    m = {c: sum(v) / len(v) for c, v in d.items()}
    wc = m.get("SQ_WAVE_CYCLES", 1)
    print(k, {c: round(v / wc, 3) for c, v in m.items() if c.startswith("SQ_") and c not in ("SQ_WAVE_CYCLES", "SQ_VALU_MFMA_BUSY_CYCLES", "SQ_INSTS_VALU")},
          "mfma_busy", round(m["SQ_VALU_MFMA_BUSY_CYCLES"] / (m["GRBM_GUI_ACTIVE"] / 8 * 1024), 3), "wave_cycles", wc, "gui", m["GRBM_GUI_ACTIVE"], "insts_valu", m.get("SQ_INSTS_VALU"))
PY
