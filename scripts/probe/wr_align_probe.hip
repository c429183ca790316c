// micro-benchmark: does the 128-byte alignment of a wave's 1-KB store instructions matter?  Persistent waves write
// contiguous slices of NIT KB (16 B per lane per store) whose start is (A) 1-KB aligned, (B) only 16-B aligned with the
// store instructions laid from the slice start (what copy_out_fixed does), (C) 16-B aligned slices but the store
// instructions laid on 128-B line boundaries (first and last instruction partly predicated off).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
template <int NIT, int MODE>
__global__ __launch_bounds__(64) void wr(double* __restrict__ out, long n_slices, int shrink) {
  const int lane = threadIdx.x;
  const long stride = NIT * 128 - shrink;   // doubles per slice (even)
  for (long s = blockIdx.x; s < n_slices; s += gridDim.x) {
    double* dst = out + s * stride;
    double2 v = make_double2((double)s, 1.0);
    if (MODE != 2) {
      const long npairs = stride / 2;
#pragma unroll
      for (int it = 0; it < NIT; ++it) {
        const long p = it * 64 + lane;
        if (p < npairs) *reinterpret_cast<double2*>(dst + 2 * p) = v;
      }
    } else {
      const uintptr_t a = reinterpret_cast<uintptr_t>(dst);
      double* al = reinterpret_cast<double*>(a & ~(uintptr_t)1023);   // instruction chunks on 1-KB boundaries
      const long first = (long)((a & 1023) >> 4), last = first + stride / 2;   // pair range inside the aligned frame
#pragma unroll
      for (int it = 0; it < NIT + 1; ++it) {
        const long p = it * 64 + lane;
        if (p >= first && p < last) *reinterpret_cast<double2*>(al + 2 * p) = v;
      }
    }
  }
}
template <int NIT, int MODE>
void run(double* d, long total_doubles, int waves_per_cu, int shrink) {
  long n_slices = total_doubles / (NIT * 128) - 1;
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  int grid = waves_per_cu * 256;
  for (int i = 0; i < 2; ++i) hipLaunchKernelGGL((wr<NIT, MODE>), dim3(grid), dim3(64), 0, 0, d, n_slices, shrink);
  hipEventRecord(a);
  for (int i = 0; i < 5; ++i) hipLaunchKernelGGL((wr<NIT, MODE>), dim3(grid), dim3(64), 0, 0, d, n_slices, shrink);
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b); ms /= 5;
  printf("NIT=%2d mode=%d shrink=%2d waves/CU=%2d : %.3f ms  %.2f TB/s\n", NIT, MODE, shrink, waves_per_cu, ms,
         n_slices * (NIT * 128 - shrink) * 8.0 / ms / 1e9);
}
int main() {
  long total = 4600000000L / 8;
  double* d; hipMalloc(&d, total * 8);
  for (int rep = 0; rep < 2; ++rep)
    for (int w : {4}) {
      run<39, 0>(d, total, w, 0);
      run<39, 1>(d, total, w, 6);
      run<39, 1>(d, total, w, 16);
      run<39, 1>(d, total, w, 32);
      run<39, 1>(d, total, w, 48);
      run<39, 1>(d, total, w, 64);
      run<39, 1>(d, total, w, 96);
      run<18, 0>(d, total, w, 0);
      run<18, 1>(d, total, w, 16);
      run<18, 1>(d, total, w, 32);
      run<18, 1>(d, total, w, 64);
    }
  return 0;
}
