// micro-benchmark: does the shape of the store stream matter?  (a) every wave streams its own contiguous
// 39 KB slice (the kernels' copy-out), (b) the four waves of a workgroup stream one 156 KB region together
// (4 KB per step), (c) a grid-stride fill (16 B per thread, consecutive threads consecutive addresses).
#include <hip/hip_runtime.h>
#include <cstdio>
constexpr int NIT = 39;
__global__ __launch_bounds__(64) void wave_slices(double* __restrict__ out, long n_slices) {
  const int lane = threadIdx.x;
  for (long s = blockIdx.x; s < n_slices; s += gridDim.x) {
    double* dst = out + s * (long)(NIT * 128);
#pragma unroll
    for (int it = 0; it < NIT; ++it) *reinterpret_cast<double2*>(dst + it * 128 + lane * 2) = make_double2((double)s, (double)it);
  }
}
__global__ __launch_bounds__(256) void block_regions(double* __restrict__ out, long n_regions) {
  const int t = threadIdx.x;
  for (long s = blockIdx.x; s < n_regions; s += gridDim.x) {
    double* dst = out + s * (long)(NIT * 512);
#pragma unroll
    for (int it = 0; it < NIT; ++it) *reinterpret_cast<double2*>(dst + it * 512 + t * 2) = make_double2((double)s, (double)it);
  }
}
__global__ __launch_bounds__(256) void grid_fill(double* __restrict__ out, long n_pairs) {
  for (long i = blockIdx.x * 256L + threadIdx.x; i < n_pairs; i += gridDim.x * 256L)
    reinterpret_cast<double2*>(out)[i] = make_double2(1.0, 2.0);
}
template <typename F>
void timeit(const char* name, double bytes, F launch) {
  hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
  for (int i = 0; i < 2; ++i) launch();
  (void)hipEventRecord(a);
  for (int i = 0; i < 5; ++i) launch();
  (void)hipEventRecord(b); (void)hipEventSynchronize(b);
  float ms; (void)hipEventElapsedTime(&ms, a, b); ms /= 5;
  printf("%-40s %.3f ms  %.2f TB/s\n", name, ms, bytes / ms / 1e9);
}
int main() {
  long total = 4600000000L / 8;
  double* d; (void)hipMalloc(&d, total * 8);
  long n_slices = total / (NIT * 128), n_regions = total / (NIT * 512);
  for (int w : {4, 8}) {
    char nm[64];
    snprintf(nm, 64, "wave slices, %d waves/CU", w);
    timeit(nm, n_slices * NIT * 1024.0, [&] { hipLaunchKernelGGL(wave_slices, dim3(w * 256), dim3(64), 0, 0, d, n_slices); });
    snprintf(nm, 64, "block regions, %d blocks/CU", w / 4 ? w / 4 : 1);
    timeit(nm, n_regions * NIT * 4096.0, [&] { hipLaunchKernelGGL(block_regions, dim3((w / 4 ? w / 4 : 1) * 256), dim3(256), 0, 0, d, n_regions); });
  }
  for (int g : {2048, 8192, 65536})  {
    char nm[64]; snprintf(nm, 64, "grid-stride fill, %d blocks", g);
    timeit(nm, total * 8.0, [&] { hipLaunchKernelGGL(grid_fill, dim3(g), dim3(256), 0, 0, d, total / 2); });
  }
  return 0;
}
