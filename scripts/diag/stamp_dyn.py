#!/usr/bin/env python3
"""Diagnostic build of the library with s_memtime stamps around the phases of dyn_kernel's loop (the product source has no
diagnostic hooks): patches a COPY of kernels.hip, compiles it and links towr_amd/libtowr_amd_dynstamps.so.  Run
scripts/diag/dyn_stamps.py with TWR_AMD_LIB pointing at that library for the per-phase split.
Every stamp waits for the wave's LDS / scalar queue (s_memtime returns through lgkmcnt): the split is what matters, and a
phase that follows LDS traffic is charged for draining it."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
SRC = os.path.join(ROOT, "towr_amd", "csrc")
s = open(os.path.join(SRC, "kernels.hip")).read()


def rep(old, new):
    global s
    assert old in s, old[:70]
    s = s.replace(old, new, 1)


rep("// XC = 64-entry chunks of the staging map the slices of the batch use",
    "__device__ unsigned long long g_dyn_stamps[1024 * 8];\n"
    "#define STAMP(k) { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); acc[k] += t_ - t_last; t_last = t_; }\n"
    "// XC = 64-entry chunks of the staging map the slices of the batch use")
rep("  for (; i <= last; i += stride) {\n    const DynWork w3 = work[min(i + 3 * stride, last)];",
    "  unsigned long long acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, t_last = __builtin_amdgcn_s_memtime();\n"
    "  const int i_first = i;\n"
    "  for (; i <= last; i += stride) {\n    const DynWork w3 = work[min(i + 3 * stride, last)];")
rep("    dyn2_front(fr0, xs, lane, S);                                                            // F\n",
    "    dyn2_front(fr0, xs, lane, S);\n    STAMP(0)\n")
rep("    dyn2_load_front(w1, sel1, lane, fr1);                                                    //    front records of slice i+1\n",
    "    dyn2_load_front(w1, sel1, lane, fr1);\n    STAMP(1)\n")
rep("    __builtin_amdgcn_s_setprio(0);\n    dyn2_back(", "    __builtin_amdgcn_s_setprio(0);\n    STAMP(2)\n    dyn2_back(")
rep("    stage_x(xr);                                                                             // S\n", "    STAMP(3)\n    stage_x(xr);\n")
rep("    mapr = load_map(w3);\n    wp = w0;", "    mapr = load_map(w3);\n    STAMP(4)\n    wp = w0;")
rep("    w2 = w3;\n  }\n  copy_out(pdst, pg, wp.nvals, wp.cnt);                 // last slice of this workgroup",
    "    w2 = w3;\n    acc[7] += 1;\n  }\n"
    "  if (lane == 0 && i_first < 1024 && WANT_G && WANT_J)\n    for (int q = 0; q < 8; ++q) g_dyn_stamps[i_first * 8 + q] = acc[q];\n"
    "  copy_out(pdst, pg, wp.nvals, wp.cnt);                 // last slice of this workgroup")
rep("int dyn_dump_doubles() { return kDynImage + 2 + 96; }",
    "int dyn_dump_doubles() { return kDynImage + 2 + 96; }\n"
    "extern \"C\" int twr_debug_dyn_stamps(unsigned long long* out, int n) {\n"
    "  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_dyn_stamps), sizeof(unsigned long long) * (size_t)n);\n}")
tmp = os.path.join(SRC, "_kernels_dynstamps.hip")
open(tmp, "w").write(s)
hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
try:
    subprocess.check_call([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-c", "-o", os.path.join(SRC, "_kernels_dynstamps.o"), tmp], cwd=SRC)
    subprocess.check_call([hipcc, "--offload-arch=gfx950", "--hip-link", "-fPIC", "-shared", "-o", os.path.join(ROOT, "towr_amd", "libtowr_amd_dynstamps.so"),
                           "_kernels_dynstamps.o", "rom_tu.o", "structure.o", "capi.o"], cwd=SRC)
finally:
    for f in (tmp, os.path.join(SRC, "_kernels_dynstamps.o")):
        if os.path.exists(f):
            os.remove(f)
print("towr_amd/libtowr_amd_dynstamps.so built; run: TWR_AMD_LIB=$PWD/towr_amd/libtowr_amd_dynstamps.so python3 scripts/diag/dyn_stamps.py")
