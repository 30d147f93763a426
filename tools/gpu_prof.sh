#!/bin/bash
# usage: gpu_prof.sh <name> <bench args...>   kernel-trace stats of one bench command into gpurun_out/<name>/
cd "$GRAFT_REPO_ROOT" || exit 1
name=$1; shift
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$name; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
timeout -s KILL 240 rocprofv3 --kernel-trace --stats --output-format csv -d $O -o p -- python3 $R/bench.py --no-cpu-baseline --no-extras "$@" > $O/log.txt 2>&1 || exit 3
python3 - <<PY
import csv
rows=list(csv.DictReader(open("$O/p_kernel_stats.csv")))
tot=sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[:16]:
    print(f'{r["Name"][:100]:100s} calls {r["Calls"]:>5s} avg_us {float(r["AverageNs"])/1e3:8.1f} pct {float(r["TotalDurationNs"])/tot*100:5.1f}')
PY
