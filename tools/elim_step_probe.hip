// What does one step of the 16 x 16 fused Cholesky+inverse elimination (diag_panel, step A/D)
// cost on its own?  256 threads, one D and one Y entry per thread, column k of D and row k of Y
// published through an LDS strip, one barrier per step.  Variants strip parts of the step.
//   0 full   1 no fp64 math   2 no barrier   3 no LDS traffic (registers only)   4 barrier only
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <math.h>
template <int VAR>
__global__ __launch_bounds__(256, 2) void k(double* out, long long* clk, int reps) {
  __shared__ double strip[2][4][16];
  const int tid = threadIdx.x, q = tid & 15, ib = tid >> 4;
  double dr0 = 1.0 + 1e-3 * tid + (ib == q ? 40.0 : 0.0), di0 = (ib == q) ? 0.0 : 1e-3 * tid;
  double acc = 0;
  long long t0 = __builtin_amdgcn_s_memtime();
  for (int r = 0; r < reps; ++r) {
    double dr = dr0, di = di0, yr = (ib == q) ? 1.0 : 0.0, yi = 0.0;
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      double* st = &strip[k & 1][0][0];
      if (VAR != 3) {
        if (q == k && ib >= k) { st[ib] = dr; st[16 + ib] = di; }
        if (ib == k && q <= k) { st[32 + q] = yr; st[48 + q] = yi; }
      }
      if (VAR != 2) __syncthreads();
      double dkk, cr, cm, a0, a1, b0, b1;
      if (VAR != 3 && VAR != 4) {
        dkk = st[k]; cr = st[ib]; cm = st[16 + ib];
        a0 = st[32 + q]; a1 = st[48 + q]; b0 = st[q]; b1 = st[16 + q];
      } else {
        dkk = dr + 40.0; cr = dr; cm = di; a0 = yr; a1 = yi; b0 = dr; b1 = di;
      }
      if (VAR == 1 || VAR == 4) { acc += dkk + cr + cm + a0 + a1 + b0 + b1; continue; }
      const bool isY = q <= k;
      const double sr = isY ? a0 : b0, si = isY ? a1 : -b1;
      const double pr = cr * sr - cm * si, pi = cr * si + cm * sr;
      const double r0 = __builtin_amdgcn_rcp(dkk);
      const double rinv = fma(r0, fma(-dkk, r0, 1.0), r0);
      const bool act = ib > k;
      const double my = (act && isY) ? -rinv : 0.0;
      const double md = (act && !isY && (q <= ib)) ? -rinv : 0.0;
      yr = fma(pr, my, yr); yi = fma(pi, my, yi);
      dr = fma(pr, md, dr); di = fma(pi, md, di);
    }
    acc += dr + di + yr + yi;
  }
  long long t1 = __builtin_amdgcn_s_memtime();
  if (tid == 0 && blockIdx.x == 0) clk[0] = t1 - t0;
  out[blockIdx.x * 256 + tid] = acc;
}
// Same step with a co-resident workgroup streaming f64 MFMAs on the same SIMDs (blocks
// 256..511 land on the CUs of blocks 0..255): PRIO = s_setprio level of the eliminating waves.
typedef double d4 __attribute__((ext_vector_type(4)));
template <int PRIO, int NR>
__global__ __launch_bounds__(256, 2) void kc(double* out, long long* clk, int reps) {
  __shared__ double strip[2][4][16];
  const int tid = threadIdx.x, q = tid & 15, ib = tid >> 4;
  if (blockIdx.x >= 256) {          // the MFMA streamer: runs longer than the timed blocks
    d4 a0 = {0, 0, 0, 0}, a1 = a0, a2 = a0, a3 = a0;
    const double x = 1.0 + tid * 1e-6;
    for (int i = 0; i < reps * 16 * 12; ++i) {
      a0 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, x, a0, 0, 0, 0);
      a1 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, x, a1, 0, 0, 0);
      a2 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, x, a2, 0, 0, 0);
      a3 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, x, a3, 0, 0, 0);
    }
    out[blockIdx.x * 256 + tid] = a0[0] + a1[1] + a2[2] + a3[3];
    return;
  }
  if (PRIO) __builtin_amdgcn_s_setprio(PRIO);
  double dr0 = 1.0 + 1e-3 * tid + (ib == q ? 40.0 : 0.0), di0 = (ib == q) ? 0.0 : 1e-3 * tid;
  double acc = 0;
  long long t0 = __builtin_amdgcn_s_memtime();
  for (int r = 0; r < reps; ++r) {
    double dr = dr0, di = di0, yr = (ib == q) ? 1.0 : 0.0, yi = 0.0;
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      double* st = &strip[k & 1][0][0];
      if (q == k && ib >= k) { st[ib] = dr; st[16 + ib] = di; }
      if (ib == k && q <= k) { st[32 + q] = yr; st[48 + q] = yi; }
      __syncthreads();
      const double dkk = st[k], cr = st[ib], cm = st[16 + ib];
      const double a0 = st[32 + q], a1 = st[48 + q], b0 = st[q], b1 = st[16 + q];
      const bool isY = q <= k;
      const double sr = isY ? a0 : b0, si = isY ? a1 : -b1;
      const double pr = cr * sr - cm * si, pi = cr * si + cm * sr;
      double rinv;
      if (NR == 9) rinv = 1.0 / dkk;
      else {
        rinv = __builtin_amdgcn_rcp(dkk);
        for (int n = 0; n < NR; ++n) rinv = fma(rinv, fma(-dkk, rinv, 1.0), rinv);
      }
      const bool act = ib > k;
      const double my = (act && isY) ? -rinv : 0.0;
      const double md = (act && !isY && (q <= ib)) ? -rinv : 0.0;
      yr = fma(pr, my, yr); yi = fma(pi, my, yi);
      dr = fma(pr, md, dr); di = fma(pi, md, di);
    }
    acc += dr + di + yr + yi;
  }
  long long t1 = __builtin_amdgcn_s_memtime();
  if (tid == 0 && blockIdx.x == 0) clk[0] = t1 - t0;
  out[blockIdx.x * 256 + tid] = acc;
}
// VALU-lean form: every thread stores its D and Y entry into LDS matrices each step (no publish
// conditions), D is kept strictly lower there (diagonal in a side array), so retired rows and
// columns read back as zeros and no per-step masks are needed; only loop-invariant masks remain.
struct __attribute__((aligned(16))) c2 { double re, im; };
template <int PRIO, int MFMA>
__global__ __launch_bounds__(256, 2) void kl(double* out, long long* clk, int reps) {
  __shared__ c2 Dm[2][16][17];
  __shared__ c2 Ym[2][16][17];
  __shared__ double dg[2][16];
  const int tid = threadIdx.x, q = tid & 15, ib = tid >> 4;
  if (MFMA && blockIdx.x >= 256) {
    d4 a0 = {0, 0, 0, 0}, a1 = a0, a2 = a0, a3 = a0;
    const double x = 1.0 + tid * 1e-6;
    for (int i = 0; i < reps * 16 * 12; ++i) {
      a0 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, x, a0, 0, 0, 0);
      a1 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, x, a1, 0, 0, 0);
      a2 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, x, a2, 0, 0, 0);
      a3 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, x, a3, 0, 0, 0);
    }
    out[blockIdx.x * 256 + tid] = a0[0] + a1[1] + a2[2] + a3[3];
    return;
  }
  if (PRIO) __builtin_amdgcn_s_setprio(PRIO);
  for (int e = tid; e < 2 * 16 * 17; e += 256) { (&Dm[0][0][0])[e] = c2{0, 0}; (&Ym[0][0][0])[e] = c2{0, 0}; }
  __syncthreads();
  const bool low = q < ib, dia = q == ib;
  double dr0 = 1.0 + 1e-3 * tid + (ib == q ? 40.0 : 0.0), di0 = (ib == q) ? 0.0 : 1e-3 * tid;
  if (!(low || dia)) { dr0 = 0; di0 = 0; }
  double acc = 0;
  long long t0 = __builtin_amdgcn_s_memtime();
  for (int r = 0; r < reps; ++r) {
    double dr = dr0, di = di0, yr = (ib == q) ? 1.0 : 0.0, yi = 0.0;
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      const int bf = k & 1;
      if (!dia) Dm[bf][ib][q] = c2{dr, di}; else dg[bf][ib] = dr;
      Ym[bf][ib][q] = c2{yr, yi};
      __syncthreads();
      const double dkk = dg[bf][k];
      const c2 c = Dm[bf][ib][k], cq = Dm[bf][q][k], sy = Ym[bf][k][q];
      double rinv = __builtin_amdgcn_rcp(dkk);
      rinv = fma(rinv, fma(-dkk, rinv, 1.0), rinv);
      const double lr = c.re * rinv, lm = c.im * rinv;
      yr = fma(-lr, sy.re, yr); yr = fma(lm, sy.im, yr);
      yi = fma(-lr, sy.im, yi); yi = fma(-lm, sy.re, yi);
      if (low || dia) {
        dr = fma(-lr, cq.re, dr); dr = fma(-lm, cq.im, dr);
        di = fma(-lm, cq.re, di); di = fma(lr, cq.im, di);
      }
    }
    acc += dr + di + yr + yi;
  }
  long long t1 = __builtin_amdgcn_s_memtime();
  if (tid == 0 && blockIdx.x == 0) clk[0] = t1 - t0;
  out[blockIdx.x * 256 + tid] = acc;
}
// Cost of individual ingredients next to a co-resident MFMA streamer:
//  0: 16 dependent v_fma_f64   1: 16 independent v_fma_f64   2: s_barrier   3: ds_write+ds_read (b64)
//  4: 16 v_mov/cndmask-like 32-bit VALU   5: 16 SALU ops
template <int WHAT>
__global__ __launch_bounds__(256, 2) void ki(double* out, long long* clk, int reps) {
  __shared__ double buf[512];
  const int tid = threadIdx.x;
  if (blockIdx.x >= 256) {
    d4 a0 = {0, 0, 0, 0}, a1 = a0, a2 = a0, a3 = a0;
    const double x = 1.0 + tid * 1e-6;
    for (int i = 0; i < reps * 40; ++i) {
      a0 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, x, a0, 0, 0, 0);
      a1 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, x, a1, 0, 0, 0);
      a2 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, x, a2, 0, 0, 0);
      a3 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, x, a3, 0, 0, 0);
    }
    out[blockIdx.x * 256 + tid] = a0[0] + a1[1] + a2[2] + a3[3];
    return;
  }
  double a = 1.0 + tid, b = a + 1, c = a + 2, d = a + 3;
  int iv = tid, sacc = reps;
  buf[tid] = a;
  __syncthreads();
  long long t0 = __builtin_amdgcn_s_memtime();
  for (int r = 0; r < reps; ++r) {
    if (WHAT == 0) {
#pragma unroll
      for (int i = 0; i < 16; ++i) a = fma(a, 1.0000001, 1e-9);
    } else if (WHAT == 1) {
#pragma unroll
      for (int i = 0; i < 4; ++i) { a = fma(a, 1.0000001, 1e-9); b = fma(b, 1.0000001, 1e-9); c = fma(c, 1.0000001, 1e-9); d = fma(d, 1.0000001, 1e-9); }
    } else if (WHAT == 2) {
      __syncthreads();
    } else if (WHAT == 3) {
      buf[tid ^ 1] = a;
      __builtin_amdgcn_s_waitcnt(0xc07f);
      a = buf[tid] + 1e-9;
    } else if (WHAT == 4) {
#pragma unroll
      for (int i = 0; i < 16; ++i) iv = (iv ^ (iv >> 3)) + 12345;
    } else {
#pragma unroll
      for (int i = 0; i < 16; ++i) sacc = __builtin_amdgcn_readfirstlane(sacc) * 3 + 1;
    }
  }
  long long t1 = __builtin_amdgcn_s_memtime();
  if (tid == 0 && blockIdx.x == 0) clk[0] = t1 - t0;
  out[blockIdx.x * 256 + tid] = a + b + c + d + iv + sacc;
}
template <typename K> void run(const char* n, K kern, int blocks) {
  double* o; long long* c; (void)hipMalloc(&o, 8 * 256 * 2048); (void)hipMalloc(&c, 8);
  const int reps = 200;
  hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, o, c, reps);
  (void)hipDeviceSynchronize();
  long long h; (void)hipMemcpy(&h, c, 8, hipMemcpyDeviceToHost);
  printf("%-34s blocks=%4d : %7.1f cycles per step\n", n, blocks, h / (double)(reps * 16));
  (void)hipFree(o); (void)hipFree(c);
}
int main() {
  for (int b : {256, 512}) {
    run("full step", k<0>, b);
    run("no fp64 math", k<1>, b);
    run("no barrier", k<2>, b);
    run("no LDS traffic", k<3>, b);
    run("barrier only", k<4>, b);
  }
  run("contended, prio 0, IEEE div", kc<0, 9>, 512);
  run("contended, prio 0, rcp+1NR", kc<0, 1>, 512);
  run("contended, prio 1, rcp+1NR", kc<1, 1>, 512);
  run("contended, prio 3, rcp+1NR", kc<3, 1>, 512);
  run("contended, prio 3, rcp only", kc<3, 0>, 512);
  run("alone (256 blocks), rcp+1NR", kc<0, 1>, 256);
  run("lean: alone", kl<0, 0>, 256);
  run("lean: two per CU", kl<0, 0>, 512);
  run("lean: contended, prio 0", kl<0, 1>, 512);
  run("lean: contended, prio 3", kl<3, 1>, 512);
  printf("ingredients, alone then next to an MFMA streamer (cycles per group of 16 / per op):\n");
  run("16 dependent v_fma_f64, alone", ki<0>, 256);   run("16 dependent v_fma_f64, contended", ki<0>, 512);
  run("16 independent v_fma_f64, alone", ki<1>, 256); run("16 independent v_fma_f64, contended", ki<1>, 512);
  run("s_barrier, alone", ki<2>, 256);                run("s_barrier, contended", ki<2>, 512);
  run("ds_write+wait+ds_read, alone", ki<3>, 256);    run("ds_write+wait+ds_read, contended", ki<3>, 512);
  run("32 int VALU (xor/shift/add), alone", ki<4>, 256); run("32 int VALU, contended", ki<4>, 512);
  run("SALU chain, alone", ki<5>, 256);               run("SALU chain, contended", ki<5>, 512);
  {   // same results as the masked form?
    double *o1, *o2; long long* c; (void)hipMalloc(&o1, 8 * 256 * 2048); (void)hipMalloc(&o2, 8 * 256 * 2048); (void)hipMalloc(&c, 8);
    hipLaunchKernelGGL((kc<0, 1>), dim3(1), dim3(256), 0, 0, o1, c, 1);
    hipLaunchKernelGGL((kl<0, 0>), dim3(1), dim3(256), 0, 0, o2, c, 1);
    static double h1[256], h2[256];
    (void)hipMemcpy(h1, o1, 2048, hipMemcpyDeviceToHost); (void)hipMemcpy(h2, o2, 2048, hipMemcpyDeviceToHost);
    double md = 0; for (int i = 0; i < 256; ++i) { const int q = i & 15, ib = i >> 4; if (q <= ib) md = fmax(md, fabs(h1[i] - h2[i]) / fmax(1.0, fabs(h1[i]))); }
    printf("max deviation lean vs masked (lower part): %.3e\n", md);
  }
  return 0;
}
