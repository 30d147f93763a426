#!/bin/bash
# Profile set of the default bench (run on the GPU box through gpurun): kernel trace + statistics, then three
# counter passes of their own (FETCH_SIZE, WRITE_SIZE, MFMA busy), summarised into profiles-shaped files under
# gpurun_out/r3prof/.  Each rocprofv3 run has its own time limit and the script stops at the first failure.
cd "$GRAFT_REPO_ROOT" || exit 1
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3prof; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
ARGS="--no-cpu-baseline --no-extras --steps 20 $*"
timeout -s KILL 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o p -- python3 $R/bench.py $ARGS > $O/stats.log 2>&1 || exit 3
timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch -o p -- python3 $R/bench.py $ARGS > $O/fetch.log 2>&1 || exit 4
timeout -s KILL 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write -o p -- python3 $R/bench.py $ARGS > $O/write.log 2>&1 || exit 5
timeout -s KILL 200 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/mfma -o p -- python3 $R/bench.py $ARGS > $O/mfma.log 2>&1 || exit 6
cd $R
python3 tools/pmc_bench_summary.py $O/pmc.json $O/fetch $O/write $O/stats $O/mfma > $O/summary.log 2>&1 || exit 7
cp $(find $O/stats -name "*kernel_stats.csv" | head -1) $O/kernel_stats.csv
tail -n 40 $O/summary.log
