#!/bin/bash
# training path: its test file, then the step-time comparison against ATen
mkdir -p gpurun_out/r2t
timeout -k 10 400 python -u -m pytest tests/test_gpu_train.py -x -q --timeout 300 --timeout-method thread > gpurun_out/r2t/train_suite.log 2>&1
rc=$?; tail -15 gpurun_out/r2t/train_suite.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python tools/train_bench.py > gpurun_out/r2t/train_bench.json 2> gpurun_out/r2t/train_bench.err || exit 1
cat gpurun_out/r2t/train_bench.json
timeout -k 10 300 python tools/train_bench.py --batch 32 --frames 2000 --steps 5 > gpurun_out/r2t/train_bench_32x20.json 2>> gpurun_out/r2t/train_bench.err || exit 1
cat gpurun_out/r2t/train_bench_32x20.json
