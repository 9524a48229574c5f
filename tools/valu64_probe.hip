#include <hip/hip_runtime.h>
#include <stdio.h>
template <int MODE>
__global__ __launch_bounds__(256) void k(double* out, long long* clk, int iters, double x0) {
  double a = x0 + threadIdx.x, b = a + 1, c = a + 2, d = a + 3, e = a + 4, f = a + 5, g = a + 6, h = a + 7;
  const double m = 1.0000001, s = 1e-9;
  long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
    if (MODE == 0) { a = fma(a, m, s); a = fma(a, m, s); a = fma(a, m, s); a = fma(a, m, s); a = fma(a, m, s); a = fma(a, m, s); a = fma(a, m, s); a = fma(a, m, s); }
    if (MODE == 1) { a = fma(a, m, s); b = fma(b, m, s); c = fma(c, m, s); d = fma(d, m, s); e = fma(e, m, s); f = fma(f, m, s); g = fma(g, m, s); h = fma(h, m, s); }
    if (MODE == 2) { a = 1.0 / a + 3.0; }
    if (MODE == 3) { a = sqrt(a) + 3.0; }
  }
  long long t1 = __builtin_amdgcn_s_memtime();
  if (threadIdx.x == 0 && blockIdx.x == 0) clk[0] = t1 - t0;
  out[blockIdx.x * 256 + threadIdx.x] = a + b + c + d + e + f + g + h;
}
template <typename K> void run(const char* n, K kern, int blocks, double per) {
  double* o; long long* c; (void)hipMalloc(&o, 8 * 256 * 1024); (void)hipMalloc(&c, 8);
  hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, o, c, 4000, 1.5);
  (void)hipDeviceSynchronize();
  long long h; (void)hipMemcpy(&h, c, 8, hipMemcpyDeviceToHost);
  printf("%-44s blocks=%d : %.1f cycles per op\n", n, blocks, h / 4000.0 / per);
}
int main() {
  for (int b : {256, 1024}) {
    run("v_fma_f64 dependent chain (latency)", k<0>, b, 8);
    run("v_fma_f64 8 independent (throughput/wave)", k<1>, b, 8);
    run("1.0/x IEEE division (dependent)", k<2>, b, 1);
    run("sqrt(x) (dependent)", k<3>, b, 1);
  }
}
