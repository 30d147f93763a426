#!/bin/bash
# kernel durations (rocprofv3) of conv1 through the diagnostic variants of tools/diag_build.sh c1_*
cd "$GRAFT_REPO_ROOT" || exit 1
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/c1; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
for v in normal ${C1_VARIANTS:-c1_nostore c1_nosilu c1_neither}; do
  if [ $v = normal ]; then unset FCENC_LIB_PATH; else export FCENC_LIB_PATH=$R/tools/diag/libfcenc_$v.so; fi
  [ $v = normal ] || [ -f "$FCENC_LIB_PATH" ] || continue
  timeout -s KILL 120 rocprofv3 --kernel-trace --stats --output-format csv -d $O/$v -o p -- python3 $R/tools/conv1_bench.py > $O/$v.log 2>&1 || exit 3
  python3 - <<PY
import csv,glob
for f in glob.glob("$O/$v/**/p_kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "conv1" in r["Name"]: print("$v", r["Name"][:40], r["Calls"], "avg_us", round(float(r["AverageNs"])/1e3,2), "min", round(float(r["MinNs"])/1e3,2))
PY
done
