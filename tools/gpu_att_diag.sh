#!/bin/bash
# kernel durations (rocprofv3) of the attention kernel through the diagnostic variants of tools/diag_build.sh att_*
cd "$GRAFT_REPO_ROOT" || exit 1
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/att; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
for v in normal att_nostage att_noband att_nopv att_nos att_none; do
  if [ $v = normal ]; then unset FCENC_LIB_PATH; else export FCENC_LIB_PATH=$R/tools/diag/libfcenc_$v.so; fi
  [ $v = normal ] || [ -f "$FCENC_LIB_PATH" ] || continue
  timeout -s KILL 120 rocprofv3 --kernel-trace --stats --output-format csv -d $O/$v -o p -- python3 $R/tools/attn_bench.py > $O/$v.log 2>&1 || exit 3
  python3 - <<PY
import csv,glob
for f in glob.glob("$O/$v/**/p_kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "attn" in r["Name"]: print("$v", r["Name"][:60], r["Calls"], "avg_us", round(float(r["AverageNs"])/1e3,2), "min", round(float(r["MinNs"])/1e3,2))
PY
done
