#!/usr/bin/env python3
"""Kernels that issue loads from asm statements (hand-counted vmcnt) must never have such a register spilled: the
compiler takes the asm's output as ready and may park it in scratch before the load has landed.  Compiles a unit to
ISA and lists every scratch store whose source overlaps a register written by a scalar-base global load.

  python tools/check_asm_spills.py hydra_pspec_amd/csrc/hpx_backsolve_lds.hip [extra hipcc flags]
exit status 1 if there is such a spill."""
import pathlib
import re
import subprocess
import sys
import tempfile


def main():
    src = pathlib.Path(sys.argv[1])
    with tempfile.TemporaryDirectory() as td:
        out = pathlib.Path(td) / "unit.s"
        subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-Wno-unused-function",
                        f"-I{src.parent}", "-S", "--cuda-device-only", "-o", str(out), str(src)] + sys.argv[2:],
                       check=True, stderr=subprocess.DEVNULL)
        text = out.read_text().split("\n")
    bad, kernel = 0, "?"
    dst = set()
    rows = []
    for i, l in enumerate(text):
        m = re.match(r"^(_Z\w+):", l)
        if m:
            if rows:
                bad += report(kernel, dst, rows)
            kernel, dst, rows = m.group(1), set(), []
        m = re.match(r"\s*global_load_dwordx[24] v\[(\d+):(\d+)\], v\d+, s\[\d+:\d+\]", l)
        if m:
            dst.update(range(int(m.group(1)), int(m.group(2)) + 1))
        m = re.match(r"\s*scratch_store_dword(?:x\d)? off, v\[?(\d+)(?::(\d+))?\]?", l)
        if m:
            a = int(m.group(1))
            rows.append((i + 1, a, int(m.group(2)) if m.group(2) else a, l.strip()))
    if rows:
        bad += report(kernel, dst, rows)
    sys.exit(1 if bad else 0)


def report(kernel, dst, rows):
    hits = [r for r in rows if any(x in dst for x in range(r[1], r[2] + 1))]
    print(f"{kernel}: {len(rows)} scratch stores, {len(hits)} of a register that a scalar-base load writes")
    for r in hits:
        print(f"   line {r[0]}: {r[3][:90]}")
    return len(hits)


if __name__ == "__main__":
    main()
