#!/bin/bash
# Diagnostic variants of the library for tools/diag_block.py: the block kernel (convmod.hip, which carries the
# feed-forward body of ffn_body.h) is rebuilt with one experiment switch and linked with the normal objects.
#   tools/diag_build.sh [variant ...]     builds tools/diag/libfcenc_<variant>.so (default: all of them)
# The shipped library is never built with any of these switches.
set -e
cd "$(dirname "$0")/.."
python -m conformer_amd.build >/dev/null
OBJ=conformer_amd/csrc/build
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=fast"   # elementwise.hip variants: add -fno-slp-vectorize as build.py does
declare -A DEFS=([stamps]="-DFCENC_STAMPS" [noload]="-DFCENC_ABL_NOLOAD" [nomfma]="-DFCENC_ABL_NOMFMA" [nolds]="-DFCENC_ABL_NOLDS"
                 [noq]="-DFCENC_ABL_NOQ" [prio]="-DFCENC_PRIO_YOUNG" [noslp]="-fno-slp-vectorize" [nox]="-DFCENC_ABL_NOX" [noq_stamps]="-DFCENC_ABL_NOQ -DFCENC_STAMPS" [noload_nolds]="-DFCENC_ABL_NOLOAD -DFCENC_ABL_NOLDS" [nomfma_nolds]="-DFCENC_ABL_NOMFMA -DFCENC_ABL_NOLDS")
DEFS+=([c1_neither]="-DFCENC_C1_NOSILU -DFCENC_C1_NOSTORE" [c1_nostore]="-DFCENC_C1_NOSTORE" [c1_nosilu]="-DFCENC_C1_NOSILU" [c1_noslp]="-fno-slp-vectorize" [c2_samea]="-DFCENC_C2_SAMEA" [c2_noload]="-DFCENC_C2_NOLOAD" [c2_noload_a]="-DFCENC_C2_NOLOAD_A" [c2_noload_w]="-DFCENC_C2_NOLOAD_W" [c2_noldsread]="-DFCENC_C2_NOLDSREAD" [c2_neither]="-DFCENC_C2_NOLOAD -DFCENC_C2_NOLDSREAD" [att_nostage]="-DFCENC_ATT_NOSTAGE" [att_noslp]="-fno-slp-vectorize" [att_noband]="-DFCENC_ATT_NOBAND" [att_nopv]="-DFCENC_ATT_NOPV" [att_nos]="-DFCENC_ATT_NOS"
       [att_none]="-DFCENC_ATT_NOSTAGE -DFCENC_ATT_NOBAND -DFCENC_ATT_NOPV -DFCENC_ATT_NOS")
DEFS+=([ilp]="-mllvm -amdgpu-sched-strategy=max-ilp" [memclause]="-mllvm -amdgpu-sched-strategy=max-memory-clause" [att_ilp]="-mllvm -amdgpu-sched-strategy=max-ilp" [att_memclause]="-mllvm -amdgpu-sched-strategy=max-memory-clause")
for name in ${@:-stamps noload nomfma nolds noload_nolds nomfma_nolds}; do
  defs=${DEFS[$name]}
  src=convmod; case $name in att_*) src=attention;; c1_*) src=elementwise;; c2_*) src=conv256;; esac     # the file the experiment lives in
  ( hipcc $FLAGS $defs -c conformer_amd/csrc/$src.hip -o tools/diag/${src}_$name.o
    objs=$(ls $OBJ/*.o | grep -v "/$src.o")
    hipcc --offload-arch=gfx950 -shared -fPIC $objs tools/diag/${src}_$name.o -o tools/diag/libfcenc_$name.so
    echo built tools/diag/libfcenc_$name.so ) &
done
wait
