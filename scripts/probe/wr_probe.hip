// micro-benchmark: persistent waves write contiguous slices (16 B per lane per store), optionally
// draining their stores (s_waitcnt vmcnt(0)) after every slice -- the copy-out pattern of the kernels.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
template <int NIT, bool DRAIN>
__global__ __launch_bounds__(64) void wr(double* __restrict__ out, long n_slices) {
  const int lane = threadIdx.x;
  for (long s = blockIdx.x; s < n_slices; s += gridDim.x) {
    double* dst = out + s * (long)(NIT * 128);
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      double2 v = make_double2((double)s, (double)it);
      *reinterpret_cast<double2*>(dst + it * 128 + lane * 2) = v;
    }
    if (DRAIN) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
}
template <int NIT, bool DRAIN>
void run(double* d, long total_doubles, int waves_per_cu) {
  long n_slices = total_doubles / (NIT * 128);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  int grid = waves_per_cu * 256;
  for (int i = 0; i < 2; ++i) hipLaunchKernelGGL((wr<NIT, DRAIN>), dim3(grid), dim3(64), 0, 0, d, n_slices);
  hipEventRecord(a);
  for (int i = 0; i < 5; ++i) hipLaunchKernelGGL((wr<NIT, DRAIN>), dim3(grid), dim3(64), 0, 0, d, n_slices);
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b); ms /= 5;
  printf("NIT=%2d (%5.1f KB/slice) drain=%d waves/CU=%2d : %.3f ms  %.2f TB/s\n", NIT, NIT * 1.0, (int)DRAIN, waves_per_cu, ms,
         n_slices * NIT * 1024.0 / ms / 1e9);
}
int main() {
  long total = 3400000000L / 8;
  double* d; hipMalloc(&d, total * 8);
  for (int w : {4, 7, 8, 16, 32}) {
    run<21, true>(d, total, w); run<21, false>(d, total, w);
    run<39, true>(d, total, w); run<39, false>(d, total, w);
  }
  run<8, true>(d, total, 8); run<8, true>(d, total, 32); run<63, true>(d, total, 4); run<63,false>(d,total,4);
  return 0;
}
