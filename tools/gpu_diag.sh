#!/bin/bash
# one GPU session: block timing through every diagnostic variant (tools/diag_build.sh), then the phase clocks
set -e
mkdir -p gpurun_out
: > gpurun_out/diag.log
for v in "" ${DIAG_VARIANTS:-noload nomfma nolds noload_nolds nomfma_nolds stamps}; do
  if [ -z "$v" ]; then unset FCENC_LIB_PATH; else export FCENC_LIB_PATH=$PWD/tools/diag/libfcenc_$v.so; fi
  [ -z "$v" ] || [ -f "$FCENC_LIB_PATH" ] || continue
  timeout -k 10 120 python tools/diag_block.py >> gpurun_out/diag.log 2>&1
done
[ ! -f tools/diag/libfcenc_stamps.so ] || DIAG_BATCH=8 timeout -k 10 120 python tools/diag_block.py >> gpurun_out/diag.log 2>&1
cat gpurun_out/diag.log
