"""Checks compiled gfx950 assembly for uses of LDS-read results that are still in flight.

The MFMA kernels read their LDS fragments with inline-asm `ds_read_b128` and wait for them with
counted `s_waitcnt lgkmcnt(N)` statements that carry the fragment registers as in/out operands.  The
compiler does not know that the registers are written asynchronously, so nothing stops it from
placing a register copy (or any other use) of a fragment between the read and its wait -- which
works until LDS latency grows, e.g. when another kernel shares the compute unit.  This tool follows
every path through a kernel's control-flow graph, keeps the FIFO of outstanding inline-asm LDS reads
(LDS returns in order; `lgkmcnt(N)` retires all but the N youngest), and reports any instruction
that touches a destination register of a read that has not been retired yet.

    python tools/isa_lint.py build/x-hip-amdgcn-amd-amdhsa-gfx950.s [kernel-name-substring ...]
    python tools/isa_lint.py --compile conformer_amd/csrc/gemm.hip [...]     (runs hipcc -save-temps)

Exit status 1 if anything is found.
"""
import os
import re
import subprocess
import sys
import tempfile

REG = re.compile(r"\bv\[(\d+):(\d+)\]|\bv(\d+)\b")
LGKM = re.compile(r"lgkmcnt\((\d+)\)")
MAX_STATES = 200000
SCALAR = "scalar"


def regs_of(text):
    out = set()
    for m in REG.finditer(text):
        if m.group(1) is not None:
            out.update(range(int(m.group(1)), int(m.group(2)) + 1))
        else:
            out.add(int(m.group(3)))
    return frozenset(out)


def parse(lines):
    """-> (instructions [(line_no, text, op, in_asm)], label -> instruction index)"""
    ins, labels, in_asm = [], {}, False
    for no, raw in lines:
        s = raw.strip()
        if s.startswith(";;#ASMSTART"):
            in_asm = True
            continue
        if s.startswith(";;#ASMEND"):
            in_asm = False
            continue
        s = s.split(";")[0].strip()
        if not s:
            continue
        m = re.match(r"^([.\w$]+):", s)
        if m:
            labels[m.group(1)] = len(ins)
            continue
        if s.startswith("."):
            continue
        ins.append((no, s, s.split()[0], in_asm))
    return ins, labels


def lint_kernel(name, lines):
    ins, labels = parse(lines)
    problems = {}
    seen = set()
    work = [(0, ())]                 # (instruction index, fifo of dest register sets; () entries = other LGKM ops)
    while work and len(seen) < MAX_STATES:
        pc, fifo = work.pop()
        while pc < len(ins):
            if not any(e for e in fifo if e is not SCALAR):
                # nothing of ours in flight: all that matters later is whether a scalar load still is
                fifo = (SCALAR,) if SCALAR in fifo else ()
            key = (pc, fifo)
            if key in seen:
                break
            seen.add(key)
            no, text, op, in_asm = ins[pc]
            rest = text[len(op):]
            if op == "s_endpgm":
                break
            if op == "s_waitcnt":
                m = LGKM.search(text)
                # other LGKM traffic (scalar loads, compiler-issued LDS ops) only adds to the counter, so a
                # counted wait can get stricter with it but never looser: LDS reads retire in order
                if m and len(fifo) > int(m.group(1)):
                    fifo = fifo[len(fifo) - int(m.group(1)):] if int(m.group(1)) else ()
                pc += 1
                continue
            if op == "s_branch":
                pc = labels.get(rest.strip(), len(ins))
                continue
            if op.startswith("s_cbranch"):
                tgt = labels.get(rest.strip().split(",")[-1].strip())
                if tgt is not None:
                    work.append((tgt, fifo))
                pc += 1
                continue
            pending = frozenset().union(*[e for e in fifo if e is not SCALAR]) if fifo else frozenset()
            if op.startswith("ds_read") or op.startswith("ds_load"):
                operands = rest.split(",")
                dest, addr = regs_of(operands[0]), regs_of(",".join(operands[1:]))
                if addr & pending:
                    problems[no] = (text, "address register is the target of a read in flight")
                if dest & pending:
                    problems[no] = (text, "overwrites the target of a read in flight")
                fifo = fifo + ((dest if in_asm else frozenset()),)
                pc += 1
                continue
            if op.startswith("ds_") or op.startswith("s_load") or op.startswith("s_buffer_load"):
                fifo = fifo + ((SCALAR if op.startswith("s_") else frozenset()),)
            hit = regs_of(rest) & pending
            if hit:
                problems[no] = (text, "touches v%s, target of a read in flight" % sorted(hit)[:4])
            pc += 1
    out = [(name, no, text, why) for no, (text, why) in sorted(problems.items())]
    if len(seen) >= MAX_STATES:
        out.append((name, 0, "", "state limit reached: analysis incomplete"))
    return out


def kernels_of(path):
    cur, body = None, []
    with open(path) as f:
        for no, raw in enumerate(f, 1):
            m = re.match(r"^(_Z\w+):", raw)
            if m and cur is None:
                cur, body = m.group(1), []
                continue
            if cur is not None:
                body.append((no, raw))
                if "s_endpgm" in raw:
                    yield cur, body
                    cur = None


def lint_file(path, filters=()):
    found = []
    for name, body in kernels_of(path):
        if filters and not any(f in name for f in filters):
            continue
        found += lint_kernel(name, body)
    return found


def compile_to_asm(src, outdir):
    flags = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=fast"]
    try:                         # the per-file flags of the shipped build, so that the linted code is the shipped code
        sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        from conformer_amd.build import EXTRA_FLAGS
        flags += EXTRA_FLAGS.get(os.path.basename(src), [])
    except ImportError:
        pass
    obj = os.path.join(outdir, os.path.basename(src).replace(".hip", ".o"))
    subprocess.run(["hipcc", *flags, "-save-temps=obj", "-c", src, "-o", obj], check=True,
                   stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    stem = os.path.basename(src).replace(".hip", "")
    return os.path.join(outdir, stem + "-hip-amdgcn-amd-amdhsa-gfx950.s")


def main(argv):
    if argv and argv[0] == "--compile":
        total = []
        with tempfile.TemporaryDirectory() as tmp:
            for src in argv[1:]:
                total += lint_file(compile_to_asm(src, tmp))
    else:
        total = lint_file(argv[0], argv[1:])
    for name, no, text, why in total:
        print(f"{name[:70]}: line {no}: {why}\n    {text}")
    print(f"{len(total)} problem(s)")
    return 1 if total else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
