#!/bin/bash
# A/B of one diagnostic library variant against the shipped one: kernel statistics of the bench both ways, twice each
cd "$GRAFT_REPO_ROOT" || exit 1
R=$GRAFT_REPO_ROOT; V=${1:-prio}
cd /tmp; export TMPDIR=/tmp
for run in a0 b0 a1 b1; do
  O=$R/gpurun_out/ab_$V/$run; mkdir -p $O
  case $run in a*) unset FCENC_LIB_PATH;; b*) export FCENC_LIB_PATH=$R/tools/diag/libfcenc_$V.so;; esac
  timeout -s KILL 240 rocprofv3 --kernel-trace --stats --output-format csv -d $O -o p -- python3 $R/bench.py --no-cpu-baseline --no-extras > $O/log.txt 2>&1 || exit 3
  python3 - <<PY
import csv
rows=list(csv.DictReader(open("$O/p_kernel_stats.csv")))
for r in rows[:1]:
    print("$run", f'{r["Name"][:60]:60s} calls {r["Calls"]:>5s} avg_us {float(r["AverageNs"])/1e3:8.2f}')
PY
  grep -o '"ms_per_step": [0-9.]*' $O/log.txt
done
