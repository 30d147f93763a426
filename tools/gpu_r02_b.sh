#!/bin/bash
# GPU session B: no-allocation entry point test, kernel breakdown at the 8-GPU shard size, and the one
# counter pass over the streaming bench that once aborted under the profiler (first with a bounded launch queue, then as it was)
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r2b; mkdir -p $O
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout -k 10 300 python -u -m pytest tests/wip_no_alloc.py -x -q --timeout 120 --timeout-method thread > $O/noalloc.log 2>&1; rc=$?
tail -3 $O/noalloc.log
[ $rc -ne 0 ] && exit $rc
cd /tmp
timeout -s KILL 180 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_b8 -o p -- python3 $R/bench.py --no-cpu-baseline --batch 8 --steps 10 > $R/$O/prof_b8.log 2>&1 || exit 3
timeout -s KILL 180 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/$O/pmcs_sync -o s -- python3 $R/tools/stream_bench.py --lanes 1 --warm 90 --steps 10 --sync-every 1 > $R/$O/pmcs_sync.log 2>&1; echo "synced rc=$?"
tail -2 $R/$O/pmcs_sync.log | cut -c1-300
timeout -s KILL 180 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/$O/pmcs_orig -o s -- python3 $R/tools/stream_bench.py --lanes 1 --warm 90 --steps 10 > $R/$O/pmcs_orig.log 2>&1; echo "original rc=$?"
tail -2 $R/$O/pmcs_orig.log | cut -c1-300
exit 0
