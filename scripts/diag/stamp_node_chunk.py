#!/usr/bin/env python3
"""Diagnostic build with s_memtime stamps around the phases of node_chunk_kernel's loop (patched COPY of kernels.hip ->
towr_amd/libtowr_amd_nodestamps.so; the product source has no diagnostic hooks).  Run scripts/diag/node_stamps.py with
TWR_AMD_LIB pointing at it.  Every stamp waits for the wave's LDS / scalar queue (s_memtime returns through lgkmcnt)."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
SRC = os.path.join(ROOT, "towr_amd", "csrc")
s = open(os.path.join(SRC, "kernels.hip")).read()


def rep(old, new):
    global s
    assert old in s, old[:70]
    s = s.replace(old, new, 1)


rep("template <int FAM, bool WANT_G, bool WANT_J>\nTWR_DEV void fam_body(",
    "__device__ unsigned long long g_node_stamps[4 * 256 * 8];\n"
    "#define NSTAMP(k) { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); acc[k] += t_ - t_last; t_last = t_; }\n"
    "template <int FAM, bool WANT_G, bool WANT_J>\nTWR_DEV void fam_body(")
rep("  for (; i <= last; i += stride) {\n    const FamWork w3 = work[min(i + 3 * stride, last)];",
    "  unsigned long long acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, t_last = __builtin_amdgcn_s_memtime();\n  const int i_first = i;\n"
    "  for (; i <= last; i += stride) {\n    const FamWork w3 = work[min(i + 3 * stride, last)];")
rep("    fam_load_x<FAM>(w1, lane, x, in1);         // x one chunk ahead (its records arrived an iteration ago)\n",
    "    fam_load_x<FAM>(w1, lane, x, in1);\n    NSTAMP(0)\n")
rep("    fam_compute<FAM, WANT_G, WANT_J>(w0, in0, stage, gst, par, lane);\n", "    fam_compute<FAM, WANT_G, WANT_J>(w0, in0, stage, gst, par, lane);\n    NSTAMP(1)\n")
rep("    fam_store<FAM, WANT_G, WANT_J>(w0, g, jac, stage, gst, par, lane);\n", "    fam_store<FAM, WANT_G, WANT_J>(w0, g, jac, stage, gst, par, lane);\n    NSTAMP(2)\n")
rep("    w1 = w2; in1 = in2;\n    w2 = w3;\n  }",
    "    w1 = w2; in1 = in2;\n    w2 = w3;\n    NSTAMP(3)\n    acc[7] += 1;\n  }\n"
    "  if (lane == 0 && i_first < 256 && WANT_G && WANT_J)\n    for (int q = 0; q < 8; ++q) g_node_stamps[(FAM * 256 + i_first) * 8 + q] = acc[q];")
rep("int dyn_dump_doubles() { return kDynImage + 2 + 96; }",
    "int dyn_dump_doubles() { return kDynImage + 2 + 96; }\n"
    "extern \"C\" int twr_debug_node_stamps(unsigned long long* out, int n) {\n"
    "  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_node_stamps), sizeof(unsigned long long) * (size_t)n);\n}")
tmp = os.path.join(SRC, "_kernels_nodestamps.hip")
open(tmp, "w").write(s)
hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
try:
    subprocess.check_call([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-c", "-o", os.path.join(SRC, "_kernels_nodestamps.o"), tmp], cwd=SRC)
    subprocess.check_call([hipcc, "--offload-arch=gfx950", "--hip-link", "-fPIC", "-shared", "-o", os.path.join(ROOT, "towr_amd", "libtowr_amd_nodestamps.so"),
                           "_kernels_nodestamps.o", "rom_tu.o", "structure.o", "capi.o"], cwd=SRC)
finally:
    for f in (tmp, os.path.join(SRC, "_kernels_nodestamps.o")):
        if os.path.exists(f):
            os.remove(f)
print("towr_amd/libtowr_amd_nodestamps.so built; run: TWR_AMD_LIB=$PWD/towr_amd/libtowr_amd_nodestamps.so python3 scripts/diag/node_stamps.py")
