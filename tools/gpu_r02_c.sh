#!/bin/bash
# the other BASELINE configurations on one GPU: 8-GPU shards and the whole single-GPU batches of medium / large
mkdir -p gpurun_out/r2c
for spec in "fastconformer_medium 32 3000" "fastconformer_medium 256 3000" "fastconformer_large 16 6000" "fastconformer_large 128 6000"; do
  set -- $spec
  timeout -k 10 200 python bench.py --config configs/$1.json --batch $2 --frames $3 --no-cpu-baseline --no-extras --steps 10 > gpurun_out/r2c/$1_$2.json 2> gpurun_out/r2c/$1_$2.err || exit 1
  python - <<PY
import json
d=json.loads(open("gpurun_out/r2c/$1_$2.json").read().strip().splitlines()[-1])
print("$1 B=$2 T=$3", d["ms_per_step"], "ms", round(d["value"]), "audio-s/s", "padded", d["config"].get("padded_audio_sec_per_sec"), "mfma", d["config"].get("whole_forward_mfma_frac"))
PY
done
