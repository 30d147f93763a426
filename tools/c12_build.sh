#!/bin/bash
# timing-only variants of the fused conv1 -> conv2 kernel (wrong results): tools/c12/libfcenc_<variant>.so
set -e
cd "$(dirname "$0")/.."
python -m conformer_amd.build >/dev/null
OBJ=conformer_amd/csrc/build
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=fast -fno-slp-vectorize"
declare -A DEFS=([noprod]="-DFCENC_C12_NOPROD" [nomain]="-DFCENC_C12_NOMAIN" [nosilu]="-DFCENC_C12_NOSILU" [neither]="-DFCENC_C12_NOPROD -DFCENC_C12_NOMAIN" [allearly]="-DFCENC_C12_ALLEARLY" [oddlate]="-DFCENC_C12_ODDLATE" [serialpro]="-DFCENC_C12_SERIALPRO" [pair]="-DFCENC_C12_PAIR" [prio0]="-DFCENC_C12_PRIO_LEVEL=0" [prio1]="-DFCENC_C12_PRIO_LEVEL=1" [prio2]="-DFCENC_C12_PRIO_LEVEL=2")
for name in ${@:-noprod nomain nosilu neither}; do
  ( hipcc $FLAGS ${DEFS[$name]} -c conformer_amd/csrc/conv12.hip -o tools/c12/conv12_$name.o
    objs=$(ls $OBJ/*.o | grep -v "/conv12.o")
    hipcc --offload-arch=gfx950 -shared -fPIC $objs tools/c12/conv12_$name.o -o tools/c12/libfcenc_$name.so
    echo built $name ) &
done
wait
