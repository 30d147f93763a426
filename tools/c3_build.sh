#!/bin/bash
# timing-only variants of the conv3 patch kernel: tools/c12/libfcenc_<variant>.so
set -e
cd "$(dirname "$0")/.."
python -m conformer_amd.build >/dev/null
OBJ=conformer_amd/csrc/build
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=fast -fno-slp-vectorize"
declare -A DEFS=([w2]="-DFCENC_C3_WDEPTH=2" [noload]="-DFCENC_C3_NOLOAD" [l2load]="-DFCENC_C3_L2LOAD" [nowrite]="-DFCENC_C3_NOWRITE" [ring5]="-DFCENC_C3_RING=5" [noread_noload]="-DFCENC_C3_NOREAD -DFCENC_C3_NOLOAD")
mkdir -p tools/c12
for name in ${@:-ring9 noload}; do
  ( hipcc $FLAGS ${DEFS[$name]} -c conformer_amd/csrc/conv3p.hip -o tools/c12/conv3p_$name.o
    objs=$(ls $OBJ/*.o | grep -v "/conv3p.o")
    hipcc --offload-arch=gfx950 -shared -fPIC $objs tools/c12/conv3p_$name.o -o tools/c12/libfcenc_$name.so
    echo built $name ) &
done
wait
