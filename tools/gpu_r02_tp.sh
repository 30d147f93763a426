#!/bin/bash
# kernel-trace stats of one training-step variant: gpu_r02_tp.sh <name> <train_bench args...>
cd "$GRAFT_REPO_ROOT" || exit 1
name=$1; shift
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$name; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
timeout -s KILL 240 rocprofv3 --kernel-trace --stats --output-format csv -d $O -o p -- python3 $R/tools/train_bench.py "$@" > $O/log.txt 2>&1 || exit 3
python3 - <<PY
import csv
rows=list(csv.DictReader(open("$O/p_kernel_stats.csv")))
tot=sum(float(r["TotalDurationNs"]) for r in rows)
print("total kernel ms", tot/1e6)
for r in rows[:28]:
    print(f'{r["Name"][:110]:110s} calls {r["Calls"]:>6s} avg_us {float(r["AverageNs"])/1e3:8.1f} pct {float(r["TotalDurationNs"])/tot*100:5.1f}')
PY
tail -2 $O/log.txt
