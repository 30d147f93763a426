#!/bin/bash
# validation session: whole GPU suite (one process), the default bench, the streaming bench line
mkdir -p gpurun_out/r2v
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > gpurun_out/r2v/full_suite.log 2>&1
rc=$?; tail -5 gpurun_out/r2v/full_suite.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 240 python bench.py > gpurun_out/r2v/bench_default.json 2> gpurun_out/r2v/bench_default.err || exit 1
cat gpurun_out/r2v/bench_default.json
timeout -k 10 200 python bench.py --workload stream --no-cpu-baseline > gpurun_out/r2v/bench_stream.json 2> gpurun_out/r2v/bench_stream.err || exit 1
cat gpurun_out/r2v/bench_stream.json
