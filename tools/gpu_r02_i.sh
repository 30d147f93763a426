#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r2i; mkdir -p $O
timeout -k 10 400 python -u -m pytest tests/test_gpu_streaming.py tests/test_gpu_pipeline.py -m gpu -x -q --timeout 200 --timeout-method thread > $O/stream_tests.log 2>&1; rc=$?
tail -12 $O/stream_tests.log
[ $rc -ne 0 ] && exit $rc
for n in 1024 128 16 1; do
  timeout -k 10 120 python tools/stream_bench.py --streams $n --warm 95 --steps 50 2>&1 | tail -1 || exit 4
done
timeout -k 10 120 python tools/stream_bench.py --streams 1024 --warm 95 --steps 50 --lanes 1 2>&1 | tail -1
timeout -k 10 120 python tools/stream_bench.py --streams 1 --warm 95 --steps 50 --graph 2>&1 | tail -1
