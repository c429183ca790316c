#!/usr/bin/env python3
"""Diagnostic build of the library with s_memtime stamps around the phases of dyn_phase_kernel (the product source has no
diagnostic hooks): patches a COPY of kernels.hip, compiles it and links towr_amd/libtowr_amd_stamps.so.  Run
scripts/diag/pdyn_stamps.py with TWR_AMD_LIB pointing at that library to get the per-phase split.
Every stamp waits for the wave's LDS / scalar queue (s_memtime returns through lgkmcnt): the split is what matters."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
SRC = os.path.join(ROOT, "towr_amd", "csrc")
s = open(os.path.join(SRC, "kernels.hip")).read()


def rep(old, new):
    global s
    assert old in s, old[:60]
    s = s.replace(old, new, 1)


rep("// LDS: the image of one pass (dynamic size).  State at the top",
    "__device__ unsigned long long g_pdyn_stamps[1024 * 8];\n"
    "#define STAMP(k) { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); acc[k] += t_ - t_last; t_last = t_; }\n"
    "// LDS: the image of one pass (dynamic size).  State at the top")
rep("  for (; i < n_work; i += stride) {\n    const bool has1 = i + stride < n_work;   // (the last pass of a workgroup prefetches itself once more: harmless)\n    if (has1) w1 = work[i + stride];\n    APut ap;",
    "  unsigned long long acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, t_last = __builtin_amdgcn_s_memtime();\n"
    "  for (; i < n_work; i += stride) {\n    const bool has1 = i + stride < n_work;\n    if (has1) w1 = work[i + stride];\n    APut ap;")
rep("    const int nv = w0.cnt * w0.node_vals;\n    if (WANT_J) lds_clear(pdyn_lds, nv, lane);\n    PDynVals V;",
    "    const int nv = w0.cnt * w0.node_vals;\n    STAMP(0)\n    if (WANT_J) lds_clear(pdyn_lds, nv, lane);\n    STAMP(1)\n    PDynVals V;")
rep("    pdyn_math(w0, r0, in, g, lane, WANT_G, V);                                      // (G inside)\n    if (WANT_J) {\n      PDynPut pu;\n      pdyn_wait_put<6 + (WANT_G ? 1 : 0)>(ap, pu);",
    "    pdyn_math(w0, r0, in, g, lane, WANT_G, V);\n    STAMP(2)\n    if (WANT_J) {\n      PDynPut pu;\n      pdyn_wait_put<6 + (WANT_G ? 1 : 0)>(ap, pu);\n      STAMP(3)")
rep("    PDynRec r1;\n    pdyn_wait_rec<(WANT_G ? 1 : 0)>(ar, r1);\n    pdyn_issue_in(w1, r1, x, lane, ai);                                             // X\n",
    "    STAMP(4)\n    PDynRec r1;\n    pdyn_wait_rec<(WANT_G ? 1 : 0)>(ar, r1);\n    pdyn_issue_in(w1, r1, x, lane, ai);\n    STAMP(5)\n")
rep("    pdyn_wait_in<(WANT_J ? (NIT < 63 ? NIT : 63) : 0)>(ai, in);   // (NIT = 0: drains the copy-out)\n    w0 = w1;\n    r0 = r1;\n  }\n}",
    "    STAMP(6)\n    pdyn_wait_in<(WANT_J ? (NIT < 63 ? NIT : 63) : 0)>(ai, in);\n    w0 = w1;\n    r0 = r1;\n    acc[7] += 1;\n  }\n"
    "  if (lane == 0 && blockIdx.x < 1024 && WANT_G && WANT_J)\n    for (int q = 0; q < 8; ++q) g_pdyn_stamps[blockIdx.x * 8 + q] = acc[q];\n}\n"
    "extern \"C\" int twr_debug_pdyn_stamps(unsigned long long* out, int n) {\n"
    "  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_pdyn_stamps), sizeof(unsigned long long) * (size_t)n);\n}")
tmp = os.path.join(SRC, "_kernels_stamps.hip")
open(tmp, "w").write(s)
hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
try:
    subprocess.check_call([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-c", "-o", os.path.join(SRC, "_kernels_stamps.o"), tmp], cwd=SRC)
    subprocess.check_call([hipcc, "--offload-arch=gfx950", "--hip-link", "-fPIC", "-shared", "-o", os.path.join(ROOT, "towr_amd", "libtowr_amd_stamps.so"),
                           "_kernels_stamps.o", "rom_tu.o", "structure.o", "capi.o"], cwd=SRC)
finally:
    for f in (tmp, os.path.join(SRC, "_kernels_stamps.o")):
        if os.path.exists(f):
            os.remove(f)
print("towr_amd/libtowr_amd_stamps.so built; run: TWR_AMD_LIB=$PWD/towr_amd/libtowr_amd_stamps.so python3 scripts/diag/pdyn_stamps.py")
