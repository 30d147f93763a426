#!/bin/bash
# validation session: training tests, the rest of the GPU suite, the default bench, the block diagnostics
mkdir -p gpurun_out
timeout -k 10 300 python -u -m pytest tests/test_gpu_train.py -x -q --timeout 300 --timeout-method thread > gpurun_out/train_suite.log 2>&1
rc=$?; tail -8 gpurun_out/train_suite.log
[ $rc -eq 0 ] || [ $rc -eq 1 ] || exit $rc
timeout -k 10 700 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread --ignore tests/test_gpu_train.py > gpurun_out/full_suite.log 2>&1
rc=$?; tail -5 gpurun_out/full_suite.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 200 python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err || exit 1
cat gpurun_out/bench_default.json
tools/gpu_diag.sh
