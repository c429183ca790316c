// micro-benchmark: persistent waves write contiguous slices with 8 B per lane per store (512 B per
// instruction) at an odd 8-byte alignment -- the row-expansion pattern of rom_phase_kernel -- vs 16 B per lane.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
template <int BYTES, bool DRAIN>
__global__ __launch_bounds__(64) void wr(double* __restrict__ out, long n_slices, int slice_doubles, int shift) {
  const int lane = threadIdx.x;
  for (long s = blockIdx.x; s < n_slices; s += gridDim.x) {
    double* dst = out + s * (long)slice_doubles + shift;
    if (BYTES == 8) {
      for (int it = 0; it * 64 < slice_doubles - 2; ++it) dst[min(it * 64 + lane, slice_doubles - 3)] = (double)it;
    } else {
      for (int it = 0; it * 128 < slice_doubles - 2; ++it) {
        double2 v = make_double2((double)s, (double)it);
        *reinterpret_cast<double2*>(dst + min(it * 128 + lane * 2, slice_doubles - 4)) = v;
      }
    }
    if (DRAIN) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
}
template <int BYTES, bool DRAIN>
void run(double* d, long total_doubles, int waves_per_cu, int slice_doubles, int shift) {
  long n_slices = total_doubles / slice_doubles - 1;
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  int grid = waves_per_cu * 256;
  for (int i = 0; i < 2; ++i) hipLaunchKernelGGL((wr<BYTES, DRAIN>), dim3(grid), dim3(64), 0, 0, d, n_slices, slice_doubles, shift);
  hipEventRecord(a);
  for (int i = 0; i < 5; ++i) hipLaunchKernelGGL((wr<BYTES, DRAIN>), dim3(grid), dim3(64), 0, 0, d, n_slices, slice_doubles, shift);
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b); ms /= 5;
  printf("%2d B/lane slice %5d doubles shift %d drain=%d waves/CU=%2d : %.3f ms  %.2f TB/s\n", BYTES, slice_doubles, shift, (int)DRAIN,
         waves_per_cu, ms, n_slices * (double)slice_doubles * 8.0 / ms / 1e9);
}
int main() {
  long total = 3400000000L / 8;
  double* d; hipMalloc(&d, total * 8);
  for (int w : {5, 8, 16}) {
    run<8, true>(d, total, w, 5824, 1);    // 32 nodes x 182 values, odd start
    run<8, false>(d, total, w, 5824, 1);
    run<16, true>(d, total, w, 5824, 0);
    run<16, false>(d, total, w, 5824, 0);
    run<8, false>(d, total, w, 2912, 1);
    run<8, false>(d, total, w, 20496, 1);  // 16 dynamic nodes x 1281 values
    run<16, false>(d, total, w, 20496, 0);
  }
  return 0;
}
