#!/usr/bin/env python3
"""Kernels that issue loads from asm statements and wait for them with hand-counted ``s_waitcnt vmcnt(N)`` must never have
such a register touched before its load has landed: the compiler takes an asm's output as ready at once and may copy it,
spill it or read it in front of the wait.  This compiles a unit to ISA and replays every kernel's instruction stream in
program order with the hardware's rule (the vector-memory counter retires in issue order):

* every vector-memory instruction (global / scratch / buffer loads and stores, LDS-DMA) enters the queue of outstanding
  operations; ``s_waitcnt vmcnt(N)`` retires all but the N youngest;
* a *scalar-base* global load (``global_load_dwordx2/x4 v[..], vOFF, s[..]`` -- the form the asm helpers emit; the
  compiler's own loads of this form are tracked the same way, which only makes the check stricter) marks its destination
  registers as in flight until it retires;
* reported: a scratch store (spill) of a register in flight, and any other instruction that READS a register in flight.

A linear replay ignores control flow; the staged loops it is meant for wait at the top of their bodies and request at the
bottom, which a linear pass sees in the right order.  Registers requested at the very end of a loop body for the next
trip stay "in flight" into whatever follows the loop, so a finding names a candidate, to be read in the ISA.

  python tools/check_asm_spills.py hydra_pspec_amd/csrc/hpx_backsolve_lds.hip [extra hipcc flags]
exit status 1 if anything is reported."""
import pathlib
import re
import subprocess
import sys
import tempfile

VMEM = re.compile(r"\s*(global_load|global_store|global_atomic|scratch_load|scratch_store|buffer_load|buffer_store|flat_load|flat_store)")
SBASE_LOAD = re.compile(r"\s*global_load_dwordx[24] v\[(\d+):(\d+)\], v\d+, s\[\d+:\d+\]")
VREG = re.compile(r"\bv(\d+)\b|\bv\[(\d+):(\d+)\]")
WAIT = re.compile(r"\s*s_waitcnt\b(.*)")


def regs_of(text):
    out = set()
    for m in VREG.finditer(text):
        if m.group(1) is not None:
            out.add(int(m.group(1)))
        else:
            out.update(range(int(m.group(2)), int(m.group(3)) + 1))
    return out


def check_kernel(name, lines):
    queue = []            # outstanding vector-memory operations, oldest first: set of destination registers (or empty)
    findings, spills = [], 0
    for no, raw in lines:
        l = raw.split(";")[0]
        if not l.strip() or l.strip().startswith("."):
            continue
        w = WAIT.match(l)
        if w:
            m = re.search(r"vmcnt\((\d+)\)", w.group(1))
            if m:
                keep = int(m.group(1))
                queue = queue[len(queue) - keep:] if keep else []
            continue
        inflight = set().union(*queue) if queue else set()
        body = l.strip()
        op, _, rest = body.partition(" ")
        operands = rest.split(",")
        if op.startswith("scratch_store"):
            spills += 1
            src = regs_of(rest)
            if src & inflight:
                findings.append((no, "spill of a register in flight", body))
        elif inflight and not op.startswith("s_") and not op.startswith("global_load_lds"):
            # sources: every operand but the first for ordinary instructions; stores read all of theirs
            reads = regs_of(rest) if "store" in op else regs_of(",".join(operands[1:]))
            if op.startswith("v_mfma") or op.startswith("v_fma") or op.startswith("v_fmac") or "mac" in op:
                reads |= regs_of(operands[0]) if op.startswith("v_fmac") or "mac" in op else set()
            if reads & inflight:
                findings.append((no, "read of a register in flight", body))
        if VMEM.match(l):
            m = SBASE_LOAD.match(l)
            queue.append(set(range(int(m.group(1)), int(m.group(2)) + 1)) if m else set())
    print(f"{name}: {spills} scratch stores, {len(findings)} uses of a scalar-base load's register before its wait")
    for no, what, body in findings[:20]:
        print(f"   line {no}: {what}: {body[:100]}")
    return len(findings)


def main():
    src = pathlib.Path(sys.argv[1])
    with tempfile.TemporaryDirectory() as td:
        out = pathlib.Path(td) / "unit.s"
        subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-Wno-unused-function",
                        f"-I{src.parent}", "-S", "--cuda-device-only", "-o", str(out), str(src)] + sys.argv[2:],
                       check=True, stderr=subprocess.DEVNULL)
        text = out.read_text().split("\n")
    bad, kernel, rows = 0, None, []
    for i, l in enumerate(text):
        m = re.match(r"^(_Z\w+):", l)
        if m:
            if kernel and rows:
                bad += check_kernel(kernel, rows)
            kernel, rows = m.group(1), []
        elif kernel:
            if l.strip().startswith("s_endpgm"):
                bad += check_kernel(kernel, rows)
                kernel, rows = None, []
            else:
                rows.append((i + 1, l))
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
