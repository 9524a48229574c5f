// What does one step of an in-LDS elimination cost?  Cycles per iteration of:
//  0: __syncthreads only   1: + LDS broadcast read + fp64 division   2: + 4 complex RMW per thread
#include <hip/hip_runtime.h>
#include <stdio.h>
template <int MODE>
__global__ __launch_bounds__(256) void k(double* out, long long* clk, int iters) {
  __shared__ double Dre[32 * 34], Dim[32 * 34];
  const int tid = threadIdx.x;
  for (int e = tid; e < 32 * 34; e += 256) { Dre[e] = 1.0 + 1e-3 * e; Dim[e] = 1e-3 * e; }
  __syncthreads();
  const int q = tid & 31, ib = tid >> 5;
  double acc = 0;
  long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
    const int k = it & 15;
    __syncthreads();
    if (MODE >= 1) {
      const double dkk = Dre[k * 34 + k];
      const double rinv2 = 1.0 / dkk;
      acc += rinv2;
      if (MODE >= 2) {
        const double sr = Dre[q * 34 + k], si = -Dim[q * 34 + k];
        double lr[4], lm[4], tr[4], tm[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int i = (k + 1 + ib + 8 * j) & 31;
          lr[j] = Dre[i * 34 + k]; lm[j] = Dim[i * 34 + k];
          tr[j] = Dre[i * 34 + q]; tm[j] = Dim[i * 34 + q];
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const double a = lr[j] * rinv2, b = lm[j] * rinv2;
          tr[j] -= 1e-9 * (a * sr - b * si); tm[j] -= 1e-9 * (a * si + b * sr);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int i = (k + 1 + ib + 8 * j) & 31;
          if (q > k) { Dre[i * 34 + q] = tr[j]; Dim[i * 34 + q] = tm[j]; }
        }
      }
    }
  }
  long long t1 = __builtin_amdgcn_s_memtime();
  if (tid == 0 && blockIdx.x == 0) clk[0] = t1 - t0;
  if (acc == -1.0) out[tid] = acc + Dre[tid];
}
// MODE 3 equivalent: in-register elimination with published column / row (as in diag_panel)
template <int VAR>
__global__ __launch_bounds__(256) void k3(double* out, long long* clk, int iters) {
  __shared__ double strip[512];
  const int tid = threadIdx.x, q = tid & 31, ib = tid >> 5, wj = 32;
  double dr[4], di[4], yr[4], yi[4];
  for (int j = 0; j < 4; ++j) { dr[j] = 1.0 + 1e-3 * (tid + j) + (ib + 8 * j == q ? 40.0 : 0.0); di[j] = 1e-3 * tid; yr[j] = (ib + 8 * j == q); yi[j] = 0; }
  double* colr = strip; double* coli = strip + 64; double* rowr = strip + 128; double* rowi = strip + 192; double* piv = strip + 256;
  long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
    const int k = it & 31;
    const int bo = (k & 1) << 5;
    if (q == k) {
#pragma unroll
      for (int j = 0; j < 4; ++j) { const int i = ib + 8 * j; if (i >= k) { colr[bo + i] = dr[j]; coli[bo + i] = di[j]; } }
    }
    if (q <= k) {
#pragma unroll
      for (int j = 0; j < 4; ++j) if (ib + 8 * j == k) { rowr[bo + q] = yr[j]; rowi[bo + q] = yi[j]; }
    }
    __syncthreads();
    if (VAR < 2) {
    const double dkk = colr[bo + k];
    if (tid == 0) piv[k] = dkk;
    double rinv2;
    if (VAR == 0) rinv2 = 1.0 / dkk;
    else { double r0 = __builtin_amdgcn_rcp(dkk); r0 = r0 * (2.0 - dkk * r0); rinv2 = r0 * (2.0 - dkk * r0); }
    const bool isY = q <= k;
    const double sr = isY ? rowr[bo + q] : colr[bo + q];
    const double si = isY ? rowi[bo + q] : -coli[bo + q];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int i = ib + 8 * j;
      if (i > k && i < wj) {
        const double lr = colr[bo + i] * rinv2, lm = coli[bo + i] * rinv2;
        const double ur = 1e-9 * (lr * sr - lm * si), ui = 1e-9 * (lr * si + lm * sr);
        if (isY) { yr[j] -= ur; yi[j] -= ui; }
        else if (q <= i) { dr[j] -= ur; di[j] -= ui; }
      }
    }
    } else {
      // branch-free: all LDS reads first, then register math with selects
      const double dkk = colr[bo + k];
      double cr[4], cm[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) { cr[j] = colr[bo + ib + 8 * j]; cm[j] = coli[bo + ib + 8 * j]; }
      const double a0 = rowr[bo + q], a1 = rowi[bo + q], b0 = colr[bo + q], b1 = coli[bo + q];
      if (tid == 0) piv[k] = dkk;
      const double rinv2 = 1.0 / dkk;
      const bool isY = q <= k;
      const double sr = isY ? a0 : b0, si = isY ? a1 : -b1;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int i = ib + 8 * j;
        const double lr = cr[j] * rinv2, lm = cm[j] * rinv2;
        const double ur = 1e-9 * (lr * sr - lm * si), ui = 1e-9 * (lr * si + lm * sr);
        const bool act = (i > k) && (i < wj);
        const bool toY = act && isY, toD = act && !isY && (q <= i);
        yr[j] -= toY ? ur : 0.0; yi[j] -= toY ? ui : 0.0;
        dr[j] -= toD ? ur : 0.0; di[j] -= toD ? ui : 0.0;
      }
    }
  }
  long long t1 = __builtin_amdgcn_s_memtime();
  if (tid == 0 && blockIdx.x == 0) clk[0] = t1 - t0;
  double acc = 0; for (int j = 0; j < 4; ++j) acc += dr[j] + di[j] + yr[j] + yi[j];
  if (acc == -1.0) out[tid] = acc;
}
template <typename K> void run(const char* n, K kern, int blocks) {
  double* o; long long* c; (void)hipMalloc(&o, 8192); (void)hipMalloc(&c, 8);
  hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, o, c, 2000);
  (void)hipDeviceSynchronize();
  long long h; (void)hipMemcpy(&h, c, 8, hipMemcpyDeviceToHost);
  printf("%-40s blocks=%d : %.0f cycles / iteration\n", n, blocks, h / 2000.0);
}
int main() {
  for (int b : {256, 512}) {
    run("barrier only", k<0>, b);
    run("barrier + LDS bcast read + f64 division", k<1>, b);
    run("barrier + read + div + 4 complex RMW", k<2>, b);
    run("in-register, published col/row, IEEE div", k3<0>, b);
    run("in-register, published col/row, rcp+2 Newton", k3<1>, b);
    run("in-register, branch-free loads-first", k3<2>, b);
  }
}
