#!/bin/bash
# Diagnostic variants of the library for tools/diag_block.py: the block kernel (convmod.hip, which carries the
# feed-forward body of ffn_body.h) is rebuilt with one experiment switch and linked with the normal objects.
#   tools/diag_build.sh [variant ...]     builds tools/diag/libfcenc_<variant>.so (default: all of them)
# The shipped library is never built with any of these switches.
set -e
cd "$(dirname "$0")/.."
python -m conformer_amd.build >/dev/null
OBJ=conformer_amd/csrc/build
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=fast"
declare -A DEFS=([stamps]="-DFCENC_STAMPS" [noload]="-DFCENC_ABL_NOLOAD" [nomfma]="-DFCENC_ABL_NOMFMA" [nolds]="-DFCENC_ABL_NOLDS"
                 [noq]="-DFCENC_ABL_NOQ" [nox]="-DFCENC_ABL_NOX" [noq_stamps]="-DFCENC_ABL_NOQ -DFCENC_STAMPS" [noload_nolds]="-DFCENC_ABL_NOLOAD -DFCENC_ABL_NOLDS" [nomfma_nolds]="-DFCENC_ABL_NOMFMA -DFCENC_ABL_NOLDS")
for name in ${@:-stamps noload nomfma nolds noload_nolds nomfma_nolds}; do
  defs=${DEFS[$name]}
  ( hipcc $FLAGS $defs -c conformer_amd/csrc/convmod.hip -o tools/diag/convmod_$name.o
    objs=$(ls $OBJ/*.o | grep -v '/convmod.o')
    hipcc --offload-arch=gfx950 -shared -fPIC $objs tools/diag/convmod_$name.o -o tools/diag/libfcenc_$name.so
    echo built tools/diag/libfcenc_$name.so ) &
done
wait
