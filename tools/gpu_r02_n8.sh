#!/bin/bash
# rehearsal of bench.py's 4- and 8-rank paths on the one GPU of the box (gloo staging; ranks share the device)
mkdir -p gpurun_out/n8
export FCENC_DIST_BACKEND=gloo OMP_NUM_THREADS=2
for spec in "4 weak 8" "8 weak 4" "8 strong 64"; do
  set -- $spec
  timeout -k 10 280 python -m torch.distributed.run --nnodes=1 --nproc-per-node $1 --master-addr 127.0.0.1 --master-port 2970$1 bench.py --gpus $1 --steps 3 --warmup 1 --batch $3 --frames 800 --scaling $2 --no-cpu-baseline --no-extras > gpurun_out/n8/$1_$2.json 2> gpurun_out/n8/$1_$2.err || { echo "FAILED $spec"; tail -20 gpurun_out/n8/$1_$2.err; exit 1; }
  python - <<PY
import json
l=[x for x in open("gpurun_out/n8/$1_$2.json").read().splitlines() if x.startswith("{")][-1]
d=json.loads(l); print("$spec", d["n_gpus"], d["scaling"], d["ms_per_step"], round(d["value"]), d["config"]["batch_per_gpu"], d["config"]["global_batch"])
PY
done
