// Does v_mfma_f64_16x16x4_f64 issue slower when consecutive MFMAs read different A/B registers?
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef double d4 __attribute__((ext_vector_type(4)));
#define MF(a, b, c) __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0)

template <int MODE>
__global__ void k(double* sink, long long* clk, int iters, const double* src) {
  d4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
  double a0 = src[threadIdx.x], a1 = src[64 + threadIdx.x], a2 = src[128 + threadIdx.x], a3 = src[192 + threadIdx.x];
  double b0 = src[256 + threadIdx.x], b1 = src[320 + threadIdx.x], b2 = src[384 + threadIdx.x], b3 = src[448 + threadIdx.x];
  long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < iters; ++it) {
    if (MODE == 0) { c0 = MF(a0, b0, c0); c1 = MF(a0, b0, c1); c2 = MF(a0, b0, c2); c3 = MF(a0, b0, c3); }
    if (MODE == 1) { c0 = MF(a0, b0, c0); c1 = MF(a0, b1, c1); c2 = MF(a0, b2, c2); c3 = MF(a0, b3, c3); }
    if (MODE == 2) { c0 = MF(a0, b0, c0); c1 = MF(a1, b0, c1); c2 = MF(a2, b0, c2); c3 = MF(a3, b0, c3); }
    if (MODE == 3) { c0 = MF(a0, b0, c0); c1 = MF(a1, b1, c1); c2 = MF(a2, b2, c2); c3 = MF(a3, b3, c3); }
    if (MODE == 4) { c0 = MF(a0, b0, c0); c1 = MF(b0, a0, c1); c2 = MF(a0, a0, c2); c3 = MF(b0, b0, c3); }
    if (MODE == 5) { c0 = MF(a0, b0, c0); c0 = MF(a1, b1, c0); c1 = MF(a0, b1, c1); c1 = MF(a1, b0, c1); }  // gemm-like chains
  }
  long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  d4 s = c0 + c1 + c2 + c3;
  if (s[0] == -1.0) sink[threadIdx.x] = s[0] + s[1] + s[2] + s[3];
  if (blockIdx.x == 0 && threadIdx.x == 0) { clk[0] = t1 - t0; clk[1] = r1 - r0; }
}

template <typename K>
void run(const char* name, K kern, int wpersimd, int iters, const double* src) {
  int ncu = 0;
  (void)hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, 0);
  double* sink; long long* clk;
  (void)hipMalloc(&sink, 1024 * 8); (void)hipMalloc(&clk, 16);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  const int threads = 256, blocks = ncu * wpersimd;
  hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), 0, 0, sink, clk, iters, src);
  (void)hipEventRecord(e0, 0);
  hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), 0, 0, sink, clk, iters, src);
  (void)hipEventRecord(e1, 0); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  long long h[2]; (void)hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
  const double nm = (double)blocks * 4 * (double)iters * 4;
  printf("%-28s waves/SIMD=%d : %.2f ms  %.1f TFLOP/s  clock %.2f GHz  wave-cycles per MFMA %.1f\n", name, wpersimd, ms,
         nm * 2048. / (ms * 1e-3) / 1e12, (double)h[0] / ((double)h[1] / 100e6) / 1e9, (double)h[0] / ((double)iters * 4));
}

int main() {
  double h[512]; for (int i = 0; i < 512; ++i) h[i] = 1.0 + 1e-6 * i;
  double* src; (void)hipMalloc(&src, sizeof(h)); (void)hipMemcpy(src, h, sizeof(h), hipMemcpyHostToDevice);
  const int it = 20000;
  for (int w : {1, 2}) {
    run("same A same B", k<0>, w, it, src);
    run("same A, 4 different B", k<1>, w, it, src);
    run("4 different A, same B", k<2>, w, it, src);
    run("4 different A and B", k<3>, w, it, src);
    run("(a,b)(b,a)(a,a)(b,b)", k<4>, w, it, src);
    run("2 chains x 2 k-steps", k<5>, w, it, src);
  }
  return 0;
}
