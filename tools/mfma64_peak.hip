// Microbenchmark: issue rate of v_mfma_f64_16x16x4_f64 (and 4x4x4_4b) on gfx950.
// Sweeps waves per SIMD and independent accumulators.  Build:
//   hipcc -O3 --offload-arch=gfx950 tools/mfma64_peak.hip -o tools/mfma64_peak
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef double d4 __attribute__((ext_vector_type(4)));

template <int NACC>
__global__ void k16(double* sink, long long* clk, int iters) {
  d4 c[NACC];
  for (int i = 0; i < NACC; ++i) c[i] = (d4){0., 0., 0., 0.};
  const double a = 1.0 + 1e-9 * threadIdx.x, b = 1.0 - 1e-9 * threadIdx.x;
  long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) c[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c[i], 0, 0, 0);
  }
  long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  d4 s = c[0];
  for (int i = 1; i < NACC; ++i) s += c[i];
  if (s[0] == -1.0) sink[threadIdx.x] = s[0] + s[1] + s[2] + s[3];
  if (blockIdx.x == 0 && threadIdx.x == 0) { clk[0] = t1 - t0; clk[1] = r1 - r0; }
}

template <int NACC>
__global__ void k4(double* sink, long long* clk, int iters) {
  double c[NACC];
  for (int i = 0; i < NACC; ++i) c[i] = 0.;
  const double a = 1.0 + 1e-9 * threadIdx.x, b = 1.0 - 1e-9 * threadIdx.x;
  long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) c[i] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c[i], 0, 0, 0);
  }
  long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  double s = 0;
  for (int i = 0; i < NACC; ++i) s += c[i];
  if (s == -1.0) sink[threadIdx.x] = s;
  if (blockIdx.x == 0 && threadIdx.x == 0) { clk[0] = t1 - t0; clk[1] = r1 - r0; }
}

template <typename K>
void run(const char* name, K kern, int nacc, int wpersimd, int iters, double flops_per_mfma) {
  int ncu = 0;
  hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, 0);
  double* sink; long long* clk;
  hipMalloc(&sink, 1024 * 8); hipMalloc(&clk, 16);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int threads = 256 * wpersimd > 1024 ? 1024 : 256 * wpersimd;
  const int blocks = ncu * (256 * wpersimd / threads);
  hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), 0, 0, sink, clk, iters);
  hipEventRecord(e0, 0);
  hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), 0, 0, sink, clk, iters);
  hipEventRecord(e1, 0); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  long long h[2]; hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
  const double nm = (double)blocks * (threads / 64) * (double)iters * nacc;
  const double clock_ghz = (double)h[0] / ((double)h[1] / 100e6) / 1e9;
  printf("%-8s acc=%d waves/SIMD=%d : %.2f ms  %.1f TFLOP/s  in-kernel clock %.2f GHz  cycles/MFMA/SIMD %.1f\n", name,
         nacc, wpersimd, ms, nm * flops_per_mfma / (ms * 1e-3) / 1e12, clock_ghz,
         (double)h[0] / ((double)iters * nacc * wpersimd));
  hipFree(sink); hipFree(clk);
}

int main() {
  const int it = 20000;
  for (int w : {1, 2, 4}) {
    run("16x16x4", k16<1>, 1, w, it * 4, 2048.);
    run("16x16x4", k16<2>, 2, w, it * 2, 2048.);
    run("16x16x4", k16<4>, 4, w, it, 2048.);
    run("16x16x4", k16<8>, 8, w, it / 2, 2048.);
  }
  for (int w : {1, 2, 4}) {
    run("4x4x4_4b", k4<4>, 4, w, it, 512.);
    run("4x4x4_4b", k4<8>, 8, w, it / 2, 512.);
  }
  return 0;
}
