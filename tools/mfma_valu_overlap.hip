// How many fp64 vector FMAs fit in the shadow of v_mfma_f64_16x16x4_f64 (64 cycles each)?
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef double d4 __attribute__((ext_vector_type(4)));
template <int NV>
__global__ __launch_bounds__(256, 2) void k(double* out, long long* clk, int iters, double x0) {
  d4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
  double v[8];
  for (int i = 0; i < 8; ++i) v[i] = x0 + threadIdx.x + i;
  const double a = 1.0 + 1e-6 * threadIdx.x, b = 1.0 - 1e-6 * threadIdx.x, m = 1.0000001, s = 1e-9;
  long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0);
#pragma unroll
      for (int j = 0; j < NV; ++j) v[(j + 0) & 7] = fma(v[(j + 0) & 7], m, s);
      c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(b, a, c1, 0, 0, 0);
#pragma unroll
      for (int j = 0; j < NV; ++j) v[(j + 2) & 7] = fma(v[(j + 2) & 7], m, s);
      c2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, a, c2, 0, 0, 0);
#pragma unroll
      for (int j = 0; j < NV; ++j) v[(j + 4) & 7] = fma(v[(j + 4) & 7], m, s);
      c3 = __builtin_amdgcn_mfma_f64_16x16x4f64(b, b, c3, 0, 0, 0);
#pragma unroll
      for (int j = 0; j < NV; ++j) v[(j + 6) & 7] = fma(v[(j + 6) & 7], m, s);
    }
  }
  long long t1 = __builtin_amdgcn_s_memtime();
  if (threadIdx.x == 0 && blockIdx.x == 0) clk[0] = t1 - t0;
  d4 cs = c0 + c1 + c2 + c3;
  double acc = cs[0] + cs[1] + cs[2] + cs[3];
  for (int i = 0; i < 8; ++i) acc += v[i];
  out[blockIdx.x * 256 + threadIdx.x] = acc;
}
template <typename K> void run(int nv, K kern, int blocks) {
  double* o; long long* c; (void)hipMalloc(&o, 8 * 256 * 1024); (void)hipMalloc(&c, 8);
  hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, o, c, 2000, 1.5);
  (void)hipDeviceSynchronize();
  long long h; (void)hipMemcpy(&h, c, 8, hipMemcpyDeviceToHost);
  printf("%d fp64 FMAs per MFMA, %d waves/SIMD: %.1f cycles per MFMA (wave time)\n", nv, blocks / 256, h / 2000.0 / 8);
}
int main() {
  for (int b : {256, 512}) {
    run(0, k<0>, b); run(1, k<1>, b); run(2, k<2>, b); run(4, k<4>, b); run(6, k<6>, b); run(8, k<8>, b);
  }
}
