#!/bin/bash
# conv2 / conv3 durations (rocprofv3) of the timing-only builds of conv256.hip against the shipped library
cd "$GRAFT_REPO_ROOT" || exit 1
R=$GRAFT_REPO_ROOT
cd /tmp; export TMPDIR=/tmp
for v in "" ${C2_VARIANTS:-c2_noload c2_noldsread c2_neither}; do
  O=$R/gpurun_out/c2diag/${v:-shipped}; mkdir -p $O
  if [ -z "$v" ]; then unset FCENC_LIB_PATH; else export FCENC_LIB_PATH=$R/tools/diag/libfcenc_$v.so; fi
  timeout -s KILL 240 rocprofv3 --kernel-trace --stats --output-format csv -d $O -o p -- python3 $R/bench.py --no-cpu-baseline --no-extras --steps 10 > $O/log.txt 2>&1 || true   # the timing-only builds fail the bench's finite-output check after the kernels have run
  python3 - <<PY
import csv
rows=list(csv.DictReader(open("$O/p_kernel_stats.csv")))
for r in rows:
    if "conv256" in r["Name"]:
        print("${v:-shipped}", "conv256 calls", r["Calls"], "avg_us", round(float(r["AverageNs"])/1e3,1), "min", round(float(r["MinNs"])/1e3,1), "max", round(float(r["MaxNs"])/1e3,1))
PY
done
