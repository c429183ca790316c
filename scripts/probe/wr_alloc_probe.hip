// Does a store-only micro-benchmark see the slow / fast state of an allocation?  The buffer is allocated several
// times in one process behind ballasts of different sizes; on each allocation: persistent waves writing contiguous 39-KB slices
// (the copy-out pattern of rom_kernel: four waves per CU, stores drained after every slice), the same with 4-KB slices and many
// waves, and one dense grid-stride fill.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
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
__global__ __launch_bounds__(256) void fill(double2* __restrict__ out, long n) {
  for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += gridDim.x * 256L) out[i] = make_double2(1.0, 2.0);
}
template <int NIT, bool DRAIN>
double run(double* d, long total_doubles, int waves_per_cu) {
  long n_slices = total_doubles / (NIT * 128);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  int grid = waves_per_cu * 256;
  for (int i = 0; i < 2; ++i) hipLaunchKernelGGL((wr<NIT, DRAIN>), dim3(grid), dim3(64), 0, 0, d, n_slices);
  hipEventRecord(a);
  for (int i = 0; i < 5; ++i) hipLaunchKernelGGL((wr<NIT, DRAIN>), dim3(grid), dim3(64), 0, 0, d, n_slices);
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b); ms /= 5;
  hipEventDestroy(a); hipEventDestroy(b);
  return n_slices * NIT * 1024.0 / ms / 1e9;
}
double run_fill(double* d, long total_doubles) {
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  for (int i = 0; i < 2; ++i) hipLaunchKernelGGL(fill, dim3(256 * 8), dim3(256), 0, 0, reinterpret_cast<double2*>(d), total_doubles / 2);
  hipEventRecord(a);
  for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(fill, dim3(256 * 8), dim3(256), 0, 0, reinterpret_cast<double2*>(d), total_doubles / 2);
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b); ms /= 5;
  hipEventDestroy(a); hipEventDestroy(b);
  return total_doubles * 8.0 / ms / 1e9;
}
// mode 2 (argv[1] = "held"): N buffers allocated and HELD together -- N different pieces of the physical memory --, each measured
int held(int n, long bytes) {
  std::vector<double*> bufs;
  for (int i = 0; i < n; ++i) {
    double* d;
    if (hipMalloc(&d, bytes) != hipSuccess) break;
    bufs.push_back(d);
  }
  for (int rep = 0; rep < 2; ++rep)
    for (size_t i = 0; i < bufs.size(); ++i) {
      const double e = run<21, true>(bufs[i], bytes / 8, 8), a = run<39, true>(bufs[i], bytes / 8, 4);
      printf("pass %d buffer %2zu @ %p (%.1f GB): 21-KB slices x8/CU %.2f, 39-KB slices x4/CU %.2f TB/s\n", rep, i, (void*)bufs[i], bytes / 1e9, e, a);
      fflush(stdout);
    }
  for (double* d : bufs) hipFree(d);
  return 0;
}
int main(int argc, char** argv) {
  if (argc > 1 && argv[1][0] == 'h') return held(argc > 2 ? atoi(argv[2]) : 24, (argc > 3 ? atol(argv[3]) : 6700L) * 1000000L);
  const long total = 6700000000L / 8;   // the C3 Jacobian buffer
  std::vector<double> ballast_gb = {0, 2, 5, 10, 1, 3, 7, 14, 0, 2, 4, 6};
  for (double gb : ballast_gb) {
    void* ballast = nullptr;
    if (gb > 0 && hipMalloc(&ballast, (size_t)(gb * (1L << 30))) != hipSuccess) return 1;
    double* d;
    if (hipMalloc(&d, total * 8) != hipSuccess) return 1;
    if (ballast) hipFree(ballast);
    const double a = run<39, true>(d, total, 4), b = run<39, false>(d, total, 4), c = run<4, true>(d, total, 16), e = run<21, true>(d, total, 8), f = run_fill(d, total);
    printf("ballast %4.1f GB  buffer @ %p: 39-KB slices x4/CU drained %.2f, undrained %.2f; 4-KB slices x16/CU %.2f; 21-KB slices x8/CU %.2f; dense fill %.2f TB/s\n",
           gb, (void*)d, a, b, c, e, f);
    fflush(stdout);
    hipFree(d);
  }
  return 0;
}
