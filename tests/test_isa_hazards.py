"""The kernels that read LDS with inline-asm `ds_read_b128` and counted `s_waitcnt lgkmcnt(N)` are compiled
for gfx950 and their assembly is checked, along every control-flow path, for instructions that touch a
fragment register before the wait that covers its read (tools/isa_lint.py).  Such a use is invisible to
the compiler and to single-stream tests -- the data usually has arrived -- and shows up as corrupted tiles
once LDS latency grows, e.g. when kernels of several HIP streams share a compute unit."""
import os
import shutil
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import isa_lint  # noqa: E402

SOURCES = ["gemm.hip", "conv256.hip", "stream.hip", "conv12.hip", "conv3p.hip"]


@pytest.mark.skipif(shutil.which("hipcc") is None and not os.path.exists("/opt/rocm/bin/hipcc"), reason="needs hipcc")
@pytest.mark.parametrize("src", SOURCES)
def test_no_use_of_lds_reads_in_flight(src, tmp_path):
    asm = isa_lint.compile_to_asm(os.path.join(ROOT, "conformer_amd", "csrc", src), str(tmp_path))
    problems = isa_lint.lint_file(asm)
    assert not problems, "\n".join(f"{n[:60]} line {no}: {why}: {text}" for n, no, text, why in problems[:10])


def test_lint_catches_a_copy_ahead_of_its_wait(tmp_path):
    asm = tmp_path / "k.s"
    asm.write_text("""_Zkernel:
	;;#ASMSTART
	ds_read_b128 v[10:13], v2 offset:0
	;;#ASMEND
	s_cbranch_vccz .LBB0_2
	v_mov_b64_e32 v[20:21], v[10:11]
	;;#ASMSTART
	s_waitcnt lgkmcnt(0)
	;;#ASMEND
	s_branch .LBB0_3
.LBB0_2:
	;;#ASMSTART
	s_waitcnt lgkmcnt(0)
	;;#ASMEND
	v_mov_b64_e32 v[20:21], v[10:11]
.LBB0_3:
	v_mfma_f32_16x16x32_bf16 v[30:33], v[20:23], v[20:23], v[30:33]
	s_endpgm
""")
    problems = isa_lint.lint_file(str(asm))
    assert len(problems) == 1 and "v_mov_b64_e32 v[20:21], v[10:11]" in problems[0][2]
