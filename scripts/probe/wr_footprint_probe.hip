// micro-benchmark: does the footprint of a store stream matter?  Persistent waves write contiguous slices (the kernels'
// copy-out pattern, see wr_probe.hip) into buffers of 1.2 ... 7 GB; and the C3 layout itself: per problem a block of 13
// dyn slices and a block of 16 rom slices inside an 823-KB row of the Jacobian buffer (kernel A writes only the dyn
// blocks, kernel B only the rom blocks, as dyn_kernel / rom_kernel do).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
template <int NIT>
__global__ __launch_bounds__(64) void wr(double* __restrict__ out, long n_slices, long per_row, long row_stride, long row_off) {
  const int lane = threadIdx.x;
  for (long s = blockIdx.x; s < n_slices; s += gridDim.x) {
    const long row = s / per_row, k = s - row * per_row;
    double* dst = out + row * row_stride + row_off + k * (long)(NIT * 128);
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      double2 v = make_double2((double)s, (double)it);
      *reinterpret_cast<double2*>(dst + it * 128 + lane * 2) = v;
    }
  }
}
template <int NIT>
void run(const char* what, double* d, long n_slices, long per_row, long row_stride, long row_off, int waves_per_cu) {
  hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
  int grid = waves_per_cu * 256;
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((wr<NIT>), dim3(grid), dim3(64), 0, 0, d, n_slices, per_row, row_stride, row_off);
  (void)hipEventRecord(a);
  for (int i = 0; i < 10; ++i) hipLaunchKernelGGL((wr<NIT>), dim3(grid), dim3(64), 0, 0, d, n_slices, per_row, row_stride, row_off);
  (void)hipEventRecord(b); (void)hipEventSynchronize(b);
  float ms; (void)hipEventElapsedTime(&ms, a, b); ms /= 10;
  printf("%-34s NIT=%2d waves/CU=%d  %6.2f GB written over a %5.2f-GB range: %.3f ms  %.2f TB/s\n", what, NIT, waves_per_cu,
         n_slices * NIT * 1024.0 / 1e9, (n_slices / per_row) * row_stride * 8.0 / 1e9, ms, n_slices * NIT * 1024.0 / ms / 1e9);
  fflush(stdout);
}
int main() {
  const long cap = 7200000000L / 8;
  double* d; if (hipMalloc(&d, cap * 8) != hipSuccess) { printf("alloc failed\n"); return 1; }
  (void)hipMemset(d, 0, cap * 8);
  // (1) compact streams of growing size
  for (double gb : {1.2, 2.35, 3.5, 4.6, 7.0}) {
    long n18 = (long)(gb * 1e9 / (18 * 1024)), n34 = (long)(gb * 1e9 / (34 * 1024));
    run<18>("compact", d, n18, 1, 18 * 128, 0, 8);
    run<34>("compact", d, n34, 1, 34 * 128, 0, 4);
  }
  // (2) the C3 layout: rows of 102896 doubles; dyn block = 13 slices x 18 KB at the row start, rom block = 16 x 34 KB behind it
  const long row = 102896;
  for (long rows : {2048L, 4096L, 8192L}) {
    run<18>("C3 rows, dyn blocks only", d, rows * 13, 13, row, 0, 8);
    run<34>("C3 rows, rom blocks only", d, rows * 16, 16, row, 13 * 18 * 128 + 37, 4);
    run<34>("C3 rows, rom blocks, 128-B aligned", d, rows * 16, 16, row + 16 - row % 16, 13 * 18 * 128, 4);
  }
  return 0;
}
