"""Diagnostic build helper: writes towr_amd/csrc/k_stamp.hip = kernels.hip + s_memtime stamps around
the phases of the dyn kernel loop (never part of the product build)."""
import sys
s = open('towr_amd/csrc/kernels.hip').read()
s = s.replace("namespace twr {\n\n#define TWR_DEV __device__ __forceinline__", "namespace twr {\n\n#define TWR_DEV __device__ __forceinline__\n__device__ unsigned long long twr_stamps[2048 * 8];\n#define STAMP(var) { unsigned long long t_; asm volatile(\"s_memtime %0\\n\\ts_waitcnt lgkmcnt(0)\" : \"=s\"(t_) :: \"memory\"); var = t_; }")
a = '''    DynFront S;
    {
      DynX X;
      dyn_load_x(w0, sh0, ln0, x, X);
      dyn_front<NEE>(w0, sh0, ln0, X, par, __builtin_amdgcn_readfirstlane(sh0.voff), lane, S);
    }
    if (pending) {'''
assert a in s
s = s.replace(a, '''    DynFront S;
    unsigned long long t0, t1, t2, t3, t4;
    STAMP(t0)
    {
      DynX X;
      dyn_load_x(w0, sh0, ln0, x, X);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      STAMP(t1)
      dyn_front<NEE>(w0, sh0, ln0, X, par, __builtin_amdgcn_readfirstlane(sh0.voff), lane, S);
    }
    STAMP(t2)
    if (pending) {''')
b = '''    dyn_back<NEE>(w0, ln0, S, gst, stage, trash, lane, want_g, want_j);
    DynShared sh2 = sh1;'''
assert b in s
s = s.replace(b, '''    STAMP(t3)
    dyn_back<NEE>(w0, ln0, S, gst, stage, trash, lane, want_g, want_j);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    STAMP(t4)
    if (lane == 0 && blockIdx.x < 2048) {
      unsigned long long* o = twr_stamps + blockIdx.x * 8;
      unsigned long long rt; asm volatile("s_memrealtime %0\\n\\ts_waitcnt lgkmcnt(0)" : "=s"(rt) :: "memory");
      if (o[4] == 0) { o[5] = t0; o[6] = rt; }
      o[0] += t1 - t0; o[1] += t2 - t1; o[2] += t3 - t2; o[3] += t4 - t3; o[4] += 1;
      o[7] = ((t4 - o[5]) << 20) / (rt - o[6] + 1);   // shader cycles per 100 MHz tick, x 2^20
    }
    DynShared sh2 = sh1;''')
s = s.replace("int dyn_stage_capacity() { return kDynStage; }", '''extern "C" void twr_debug_stamps(unsigned long long* out, int clear) {
  if (clear) { static unsigned long long z[2048 * 8]; (void)hipMemcpyToSymbol(HIP_SYMBOL(twr_stamps), z, sizeof(z)); return; }
  (void)hipDeviceSynchronize();
  (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(twr_stamps), sizeof(unsigned long long) * 2048 * 8);
}
int dyn_stage_capacity() { return kDynStage; }''')
open('towr_amd/csrc/k_stamp.hip', 'w').write(s)
