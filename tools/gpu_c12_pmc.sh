#!/bin/bash
# SQ counters of the two patch-in-LDS front-end kernels and of the kernels they replaced (one rocprofv3 --pmc pass over
# tools/conv12_bench.py and one over tools/conv3_bench.py), summarised into gpurun_out/c12pmc/sq_counters.json
cd "$GRAFT_REPO_ROOT" || exit 1
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/c12pmc; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
PMC="SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS SQ_INSTS_VALU GRBM_GUI_ACTIVE"
timeout -s KILL 120 rocprofv3 --pmc $PMC --output-format csv -d $O/a -o p -- python3 $R/tools/conv12_bench.py --iters 3 > $O/a.log 2>&1 || exit 3
timeout -s KILL 120 rocprofv3 --pmc $PMC --output-format csv -d $O/b -o p -- python3 $R/tools/conv3_bench.py --iters 3 > $O/b.log 2>&1 || exit 4
cd $R
python3 - <<'PY'
import csv, glob, collections, json
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for path in glob.glob("gpurun_out/c12pmc/*/**/*counter_collection.csv", recursive=True):
    which = "conv3_bench" if "/b/" in path else "conv12_bench"
    for r in csv.DictReader(open(path)):
        n = r["Kernel_Name"]
        key = ("conv12_kernel" if "conv12_kernel" in n else "conv3p_kernel" if "conv3p_kernel" in n else
               "conv256_kernel (%s)" % ("conv3" if which == "conv3_bench" else "conv2") if "conv256" in n else
               "conv1_mfma_kernel" if "conv1_mfma" in n else None)
        if key:
            acc[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {"source": "rocprofv3 --pmc (one pass each) over tools/conv12_bench.py and tools/conv3_bench.py, batch 64 x 20 s; "
                 "fractions are of SQ_WAVE_CYCLES; mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 x 1024 SIMDs)"}
for k, d in sorted(acc.items()):
    m = {c: sum(v) / len(v) for c, v in d.items()}
    wc = m.get("SQ_WAVE_CYCLES", 1)
    out[k] = {c.lower(): round(v / wc, 3) for c, v in m.items()
              if c.startswith("SQ_") and c not in ("SQ_WAVE_CYCLES", "SQ_VALU_MFMA_BUSY_CYCLES", "SQ_INSTS_VALU")}
    out[k]["mfma_busy"] = round(m["SQ_VALU_MFMA_BUSY_CYCLES"] / (m["GRBM_GUI_ACTIVE"] / 8 * 1024), 3)
    out[k]["valu_instructions_per_launch"] = round(m.get("SQ_INSTS_VALU", 0))
    out[k]["gui_active_cycles_per_launch"] = round(m["GRBM_GUI_ACTIVE"])
json.dump(out, open("gpurun_out/c12pmc/sq_counters.json", "w"), indent=1)
print(json.dumps(out, indent=1))
PY
