#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r2m; mkdir -p $O
timeout -k 10 300 python -u -m pytest tests/test_gpu_kernels.py tests/test_gpu_encoder.py -m gpu -x -q --timeout 300 --timeout-method thread > $O/tests.log 2>&1; rc=$?
tail -6 $O/tests.log
[ $rc -ne 0 ] && exit $rc
for i in 1 2; do
  timeout -k 10 200 python bench.py --no-cpu-baseline --no-extras > $O/new$i.log 2>&1 || exit 4
  FCENC_ATTN_TILED=1 timeout -k 10 200 python bench.py --no-cpu-baseline --no-extras > $O/old$i.log 2>&1 || exit 5
done
python - <<PY
import json
for n in ("new1","old1","new2","old2"):
    l=json.loads(open("$O/"+n+".log").read().strip().splitlines()[-1]); print(n, l["ms_per_step"], l["config"]["forward_ms_min_median_max"])
PY
bash tools/gpu_prof.sh r2m/prof --steps 5 --warmup 2 | head -6
bash tools/gpu_prof.sh r2m/prof_b8 --batch 8 --steps 5 --warmup 2 | head -4
