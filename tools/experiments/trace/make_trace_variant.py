#!/usr/bin/env python3
"""Build a TRACED copy of the library: hpx_factor.hip with wall-clock stamps (s_memrealtime, 10 ns ticks) at the
phase boundaries of k_factor's block-column loop, for workgroup 0, one stamp per wave -- inserted into a temporary
copy of the product source at anchor lines (the product source itself carries no tracing code).  Output:
tools/experiments/trace/libhpx_trace.so (load it with HPX_LIB_PATH) exporting hpx_trace_read().

  python tools/experiments/trace/make_trace_variant.py && \
  HPX_LIB_PATH=tools/experiments/trace/libhpx_trace.so python tools/experiments/trace/trace_factor.py C2
"""
import pathlib
import shutil
import subprocess
import sys
import tempfile

ROOT = pathlib.Path(__file__).resolve().parents[3]
CSRC = ROOT / "hydra_pspec_amd" / "csrc"
OUT = pathlib.Path(__file__).resolve().parent / "libhpx_trace.so"

HEAD = r'''
__device__ unsigned long long hpx_trace_buf[4 * 64 * 16];
#define HPX_TRACE(j_, id_) do { if (blockIdx.x == 0 && (threadIdx.x & 63) == 0)                                   \
    hpx_trace_buf[((((int)threadIdx.x >> 6) * 64 + (j_)) << 4) + (id_)] = wall_clock64(); } while (0)
'''
TAIL = r'''
extern "C" int hpx_trace_read(unsigned long long* host) {
  return hipMemcpyFromSymbol(host, HIP_SYMBOL(hpx_trace_buf), sizeof(unsigned long long) * 4 * 64 * 16) == hipSuccess ? 0 : -2;
}
'''
# (anchor line prefix, text inserted BEFORE (False) or AFTER (True) it, occurrence index or None = all)
INSERTS = [
    ("  if (c0 > 0 && wave < (CT == 2 ? 3 : 1)) {", "  HPX_TRACE(c0 >> 5, 0);\n", False),
    ("  __builtin_amdgcn_s_setprio(2);", "  HPX_TRACE(c0 >> 5, 1);\n", False),
    ("  __syncthreads();                         // the combined block is complete", "  HPX_TRACE(c0 >> 5, 2);\n", True),
    ("      if (nsteps == 0 && dia) dg[ib] = dr;", "      HPX_TRACE(c0 >> 5, 3 + 2 * hb);\n", False),
    ("      if (hb == 1) __syncthreads();", "      if (hb == 1) HPX_TRACE(c0 >> 5, 4);\n", True),
    ("  __builtin_amdgcn_s_setprio(0);", "  HPX_TRACE(c0 >> 5, 6);\n", False),
    ("      // early part of the next diagonal block", "      HPX_TRACE(jb, 7);\n", False),
    ("      while (itp < nitems) {", "      HPX_TRACE(jb, 7);\n", False),
    ("    c0 += wj;", "    HPX_TRACE(jb, 9);\n", False),
]


def main():
    srcfile = pathlib.Path(sys.argv[1]) if len(sys.argv) > 1 else CSRC / "hpx_factor.hip"
    out_so = pathlib.Path(sys.argv[2]).resolve() if len(sys.argv) > 2 else OUT
    src = srcfile.read_text().split("\n")
    out = []
    done = [0] * len(INSERTS)
    for ln in src:
        hit = None
        for i, (anchor, text, after) in enumerate(INSERTS):
            if ln.startswith(anchor):
                hit = i
        if hit is not None and not INSERTS[hit][2]:
            out.append(INSERTS[hit][1].rstrip("\n"))
        out.append(ln)
        if hit is not None:
            if INSERTS[hit][2]:
                out.append(INSERTS[hit][1].rstrip("\n"))
            done[hit] += 1
        if ln.startswith('#include "hpx_internal.h"'):
            out.append(HEAD)
    assert all(d >= 1 for i, d in enumerate(done) if i not in (6, 7)) and done[6] + done[7] == 1, done
    text = "\n".join(out)
    # the pass-end barrier: stamp 8 just before it
    key = "    __syncthreads();\n    HPX_TRACE(jb, 9);"
    assert text.count(key) == 1
    text = text.replace(key, "    HPX_TRACE(jb, 8);\n" + key) + TAIL
    with tempfile.TemporaryDirectory() as td:
        td = pathlib.Path(td)
        for f in CSRC.iterdir():
            if f.suffix in (".hip", ".h"):
                shutil.copy(f, td / f.name)
        (td / "hpx_factor.hip").write_text(text)
        inc = ROOT / "include"
        srcs = sorted(p.name for p in td.glob("*.hip"))
        cmd = ["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-Wno-unused-function",
               f"-I{inc}", "-shared", "-o", str(out_so)] + srcs
        # the sources include "../../include/hpx.h": recreate that relative layout
        (td / ".." ).resolve()
        work = td / "a" / "b"
        work.mkdir(parents=True)
        for f in list(td.iterdir()):
            if f.is_file():
                shutil.move(str(f), work / f.name)
        (td / "include").mkdir()
        shutil.copy(inc / "hpx.h", td / "include" / "hpx.h")
        subprocess.check_call(cmd, cwd=work)
    print("built", out_so)


if __name__ == "__main__":
    main()
