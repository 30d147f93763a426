#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r2f; mkdir -p $O
timeout -k 10 300 python -u -m pytest tests/test_gpu_kernels.py -m gpu -x -q -k "ffn or convmod or tail" --timeout 120 --timeout-method thread > $O/ffn.log 2>&1; rc=$?
tail -15 $O/ffn.log
[ $rc -ne 0 ] && exit $rc
timeout -k 10 400 python -u -m pytest tests/test_gpu_encoder.py -m gpu -x -q -k "medium or large or fixture or lanes" --timeout 300 --timeout-method thread > $O/enc.log 2>&1; rc=$?
tail -8 $O/enc.log
[ $rc -ne 0 ] && exit $rc
for c in medium large; do
  b=32; f=3000; [ $c = large ] && b=16 && f=6000
  timeout -k 10 200 python bench.py --config configs/fastconformer_$c.json --batch $b --frames $f --no-cpu-baseline --no-extras --steps 10 > $O/bench_$c.log 2>&1 || exit 5
  python - <<PY
import json
l=json.loads(open("$O/bench_$c.log").read().strip().splitlines()[-1]); print("$c", l["ms_per_step"], l["config"]["whole_forward_mfma_frac"])
PY
done
bash tools/gpu_prof.sh r2f/prof_med --config $GRAFT_REPO_ROOT/configs/fastconformer_medium.json --batch 32 --frames 3000 --steps 5 --warmup 2 | head -12
