"""Builds conformer_amd/libfcenc_hip.so from csrc/*.hip with hipcc for gfx950 (in-tree).

    python -m conformer_amd.build [--force]

hipcc cross-compiles without a GPU, so this runs in the build container; the resulting .so
travels to the GPU box with the tree.  Nothing is compiled at import time.
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(CSRC, "build")
LIB = os.path.join(HERE, "libfcenc_hip.so")
SOURCES = ["api.hip", "gemm.hip", "conv256.hip", "conv12.hip", "conv3p.hip", "ffn.hip", "ffn_wide.hip", "convmod.hip", "convmod_wide.hip", "elementwise.hip", "attention.hip",
           "stream.hip", "decode.hip", "mel.hip", "train.hip"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=fast", "-Wall",
         "-Wno-unused-function", "-Wno-unused-variable"]
# per-file additions.  elementwise.hip: conv1 is paced by the SiLU of its 655 M outputs; -O3 packs pairs of its fp32
# multiplies into v_pk_mul_f32, which issue slower than the two scalar instructions they replace (measured on
# MI355X: 280.5 -> 275.6 us with the packing off, same bits); the block kernels measured no difference.
EXTRA_FLAGS = {"elementwise.hip": ["-fno-slp-vectorize"], "conv12.hip": ["-fno-slp-vectorize"], "conv3p.hip": ["-fno-slp-vectorize"]}


def _hipcc() -> str:
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found; cannot build libfcenc_hip.so")
    return exe


def _stale(target: str, deps) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build_native(force: bool = False, verbose: bool = False) -> str:
    os.makedirs(OBJ, exist_ok=True)
    hipcc = _hipcc()
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    headers.append(os.path.join(os.path.dirname(HERE), "include", "fcenc.h"))
    jobs = []
    for src in SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(OBJ, src.replace(".hip", ".o"))
        if force or _stale(o, [s] + headers):
            jobs.append([hipcc, *FLAGS, *EXTRA_FLAGS.get(src, []), "-c", s, "-o", o])

    def run(cmd):
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed:\n" + " ".join(cmd) + "\n" + r.stdout + r.stderr)
        if verbose and r.stderr.strip():
            print(r.stderr, file=sys.stderr)

    with ThreadPoolExecutor(max_workers=4) as ex:
        list(ex.map(run, jobs))
    objs = [os.path.join(OBJ, s.replace(".hip", ".o")) for s in SOURCES]
    if force or jobs or _stale(LIB, objs):
        run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", *objs, "-o", LIB])
    return LIB


if __name__ == "__main__":
    print(build_native(force="--force" in sys.argv, verbose=True))
