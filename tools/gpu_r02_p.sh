#!/bin/bash
# profile refresh: kernel-trace stats of the default bench, then the two HBM counter passes (separate runs),
# summarised into the JSON bench.py quotes; the streaming bench's kernel stats as well
cd "$GRAFT_REPO_ROOT" || exit 1
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2p; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
timeout -s KILL 240 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o p -- python3 $R/bench.py --no-cpu-baseline --no-extras > $O/stats.log 2>&1 || exit 3
timeout -s KILL 240 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/fetch -o f -- python3 $R/bench.py --no-cpu-baseline --no-extras --steps 5 --warmup 2 > $O/fetch.log 2>&1 || exit 4
timeout -s KILL 240 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/write -o w -- python3 $R/bench.py --no-cpu-baseline --no-extras --steps 5 --warmup 2 > $O/write.log 2>&1 || exit 5
timeout -s KILL 240 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stream -o p -- python3 $R/bench.py --no-cpu-baseline --no-extras --workload stream > $O/stream.log 2>&1 || exit 6
cd $R
python tools/pmc_bench_summary.py $O/pmc.json $O/fetch $O/write $O/stats && cat $O/pmc.json | head -60
