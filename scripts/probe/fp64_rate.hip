// Issue rate of FP64 vector instructions on gfx950: cycles per wave64 instruction for v_fma_f64 / v_add_f64 / v_mul_f64 /
// v_fmac (e32) / 32-bit VALU, one to four waves per SIMD, from s_memtime around an unrolled block of independent instructions.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
template <int KIND>
__global__ __launch_bounds__(256) void rate(double* out, long long* cyc, int iters) {
  double a[8];
  for (int i = 0; i < 8; ++i) a[i] = out[threadIdx.x + 256 * i];
  double b = out[1000], c = out[1001];
  unsigned u[8];
  for (int i = 0; i < 8; ++i) u[i] = (unsigned)a[i];
  __syncthreads();
  long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 8; ++r) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        if (KIND == 0) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
        if (KIND == 1) asm volatile("v_add_f64 %0, %0, %1" : "+v"(a[i]) : "v"(b));
        if (KIND == 2) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(a[i]) : "v"(b));
        if (KIND == 3) asm volatile("v_fmac_f64_e32 %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
        if (KIND == 4) asm volatile("v_add_u32 %0, %0, %1" : "+v"(u[i]) : "v"(u[(i + 1) & 7]));
        if (KIND == 5) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a[i]) : "s"(b), "v"(c));
      }
    }
  }
  long long t1 = __builtin_readcyclecounter();
  double s = 0;
  for (int i = 0; i < 8; ++i) s += a[i] + u[i];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
int main() {
  double* out; long long* cyc;
  CHECK(hipMalloc(&out, 8 << 20)); CHECK(hipMemset(out, 0, 8 << 20));
  CHECK(hipMalloc(&cyc, 8192 * 8));
  const char* names[] = {"v_fma_f64", "v_add_f64", "v_mul_f64", "v_fmac_f64_e32", "v_add_u32", "v_fma_f64 (sgpr operand)"};
  const int iters = 2000;
  for (int kind = 0; kind < 6; ++kind)
    for (int wpc = 1; wpc <= 4; ++wpc) {   // blocks of 256 threads = one wave per SIMD each; wpc blocks per CU
      const int blocks = 256 * wpc;
      hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
      auto launch = [&]() {
        switch (kind) {
          case 0: rate<0><<<blocks, 256>>>(out, cyc, iters); break;
          case 1: rate<1><<<blocks, 256>>>(out, cyc, iters); break;
          case 2: rate<2><<<blocks, 256>>>(out, cyc, iters); break;
          case 3: rate<3><<<blocks, 256>>>(out, cyc, iters); break;
          case 4: rate<4><<<blocks, 256>>>(out, cyc, iters); break;
          default: rate<5><<<blocks, 256>>>(out, cyc, iters); break;
        }
      };
      launch(); CHECK(hipDeviceSynchronize());
      CHECK(hipEventRecord(e0)); launch(); CHECK(hipEventRecord(e1)); CHECK(hipDeviceSynchronize());
      float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
      std::vector<long long> h(blocks);
      CHECK(hipMemcpy(h.data(), cyc, blocks * 8, hipMemcpyDeviceToHost));
      double avg = 0; for (auto v : h) avg += (double)v; avg /= blocks;
      const double n_inst = (double)iters * 64;   // per wave
      // s_memtime ticks at a constant 100 MHz on gfx9 (REFCLK); the event time gives the wall clock
      printf("%-26s %d wave(s)/SIMD: %.3f ms, %.2f ns per wave-instruction per SIMD (= %.2f cycles at 2.4 GHz), memtime ticks/instr %.3f\n", names[kind], wpc, ms,
             ms * 1e6 / (n_inst * wpc), ms * 1e6 / (n_inst * wpc) * 2.4, avg / n_inst);
    }
  return 0;
}
