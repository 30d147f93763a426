#!/bin/bash
# Kernel statistics of the other bench lines (run on the GPU box through gpurun): streaming, medium, large,
# the 8-utterance shard.  One rocprofv3 --kernel-trace --stats run each, with its own time limit; stops at the first failure.
cd "$GRAFT_REPO_ROOT" || exit 1
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3more; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
run() {  # name, bench arguments
  local name=$1; shift
  timeout -s KILL 240 rocprofv3 --kernel-trace --stats --output-format csv -d $O/$name -o p -- python3 $R/bench.py --no-cpu-baseline --no-extras --steps 10 "$@" > $O/$name.log 2>&1 || exit 3
  cp $(find $O/$name -name "*kernel_stats.csv" | head -1) $O/${name}_kernel_stats.csv
  tail -n 1 $O/$name.log | cut -c1-300
}
run stream --workload stream
run medium_b32 --config $R/configs/fastconformer_medium.json --batch 32 --frames 3000
run large_b16 --config $R/configs/fastconformer_large.json --batch 16 --frames 6000
run small_b8 --batch 8
