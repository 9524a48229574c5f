// Dense noise covariance with flagged channels: the Woodbury correction of the unflagged-noise solve
// (docs/HISTORY.md section 10.3).
#include "hpx_chain.h"

namespace {


// Everything after the solve of one iteration: back transform, residual / chi^2 / first
// ---- dense noise with flags: Woodbury correction of the unflagged-noise solution -----------------
// (hpx.h, hpx_plan_set_static_dense_flagged).  W[b] is the f x (f + T) system [I - Q^H Y_P | Q^H Y_r],
// row-major interleaved complex with leading dimension fmax + T, where (Q^H Y)[jf][col] is the model
// (U y + F f)[channel flist[jf]] of solution column col; Y_P are the columns T .. T+f-1 of X.
__global__ __launch_bounds__(256) void k_wb_system(const double* __restrict__ Sre, const double* __restrict__ Sim,
                                                   const double* __restrict__ Xre, const double* __restrict__ Xim,
                                                   const double* __restrict__ Fre, const double* __restrict__ Fim,
                                                   const int fg_shared, const int32_t* __restrict__ flist,
                                                   const int32_t* __restrict__ fcount, double* __restrict__ W_all,
                                                   const int fmax, const int N, const int M, const int T,
                                                   const int NP, const int TP, const int npad) {
  const int b = blockIdx.x, f = fcount[b], ldw = fmax + T;
  double* W = W_all + (long)b * fmax * ldw * 2;
  const double* sre = Sre + (long)b * NP * TP;
  const double* sim = Sim + (long)b * NP * TP;
  const double* xre = Xre + (long)b * npad * TP;
  const double* xim = Xim + (long)b * npad * TP;
  const double* fre = Fre + (fg_shared ? 0 : (long)b * N * M);
  const double* fim = Fim + (fg_shared ? 0 : (long)b * N * M);
  for (int e = threadIdx.x; e < f * (f + T); e += 256) {
    const int jf = e / (f + T), c = e % (f + T);
    const int col = (c < f) ? T + c : c - f;            // solution column: Y_P first, then Y_r
    const int j = flist[(long)b * fmax + jf];
    double mr = sre[(long)j * TP + col], mi = sim[(long)j * TP + col];
    for (int m = 0; m < M; ++m) {
      const double fr = fre[(long)j * M + m], fi = fim[(long)j * M + m];
      const double gr = xre[(long)(N + m) * TP + col], gi = xim[(long)(N + m) * TP + col];
      mr += gr * fr - gi * fi;
      mi += gr * fi + gi * fr;
    }
    double* w = W + ((long)jf * ldw + (c < f ? c : fmax + (c - f))) * 2;
    if (c < f) {
      w[0] = (jf == c ? 1.0 : 0.0) - mr;
      w[1] = -mi;
    } else {
      w[0] = mr;
      w[1] = mi;
    }
  }
}
// Gaussian elimination with partial pivoting on the f x (f + T) system of one baseline (global memory,
// one workgroup), then the back substitution: the coefficients c[kf][t] end up in the right-hand-side
// columns fmax .. fmax+T-1.  A vanishing pivot marks the baseline in info.
__global__ __launch_bounds__(256) void k_wb_solve(double* __restrict__ W_all, const int32_t* __restrict__ fcount,
                                                  const int fmax, const int T, int32_t* __restrict__ info,
                                                  const int iter_tag) {
  __shared__ double redv[4];
  __shared__ int redi[4], piv_s;
  __shared__ double lre[512], lim[512];
  const int b = blockIdx.x, f = fcount[b], ldw = fmax + T, tid = threadIdx.x;
  if (f == 0) return;
  double* W = W_all + (long)b * fmax * ldw * 2;
  const int ncol = fmax + T;                               // columns f .. fmax-1 are unused (never touched)
  for (int k = 0; k < f; ++k) {
    double best = -1.0;
    int at = k;
    for (int r = k + tid; r < f; r += 256) {
      const double a2 = W[((long)r * ldw + k) * 2] * W[((long)r * ldw + k) * 2] +
                        W[((long)r * ldw + k) * 2 + 1] * W[((long)r * ldw + k) * 2 + 1];
      if (a2 > best) { best = a2; at = r; }
    }
    for (int o = 32; o > 0; o >>= 1) {
      const double ob = __shfl_xor(best, o, 64);
      const int oa = __shfl_xor(at, o, 64);
      if (ob > best || (ob == best && oa < at)) { best = ob; at = oa; }
    }
    if ((tid & 63) == 0) { redv[tid >> 6] = best; redi[tid >> 6] = at; }
    __syncthreads();
    if (tid == 0) {
      int w = 0;
      for (int q = 1; q < 4; ++q)
        if (redv[q] > redv[w] || (redv[q] == redv[w] && redi[q] < redi[w])) w = q;
      piv_s = redi[w];
      // (the system is I - Q^H Y_P, entries O(1): a pivot below 1e-12 means the flags leave it singular, e.g. a unit
      // with every channel flagged, whose foreground amplitudes nothing constrains)
      if (!(redv[w] > 1e-24)) atomicCAS(&info[b], 0, iter_tag);
    }
    __syncthreads();
    const int pv = piv_s;
    if (pv != k)
      for (int c = k + tid; c < ncol; c += 256) {
        if (c >= f && c < fmax) continue;
        double* x = W + ((long)k * ldw + c) * 2;
        double* y = W + ((long)pv * ldw + c) * 2;
        const double t0 = x[0], t1 = x[1];
        x[0] = y[0]; x[1] = y[1];
        y[0] = t0; y[1] = t1;
      }
    __syncthreads();
    const double pr = W[((long)k * ldw + k) * 2], pi = W[((long)k * ldw + k) * 2 + 1];
    const double den = 1.0 / (pr * pr + pi * pi);
    for (int r = k + 1 + tid; r < f; r += 256) {           // multipliers l_r = W[r][k] / W[k][k]
      const double ar = W[((long)r * ldw + k) * 2], ai = W[((long)r * ldw + k) * 2 + 1];
      lre[r] = (ar * pr + ai * pi) * den;
      lim[r] = (ai * pr - ar * pi) * den;
    }
    __syncthreads();
    const int nc = (f - k - 1) + T, nr = f - k - 1;
    for (int e = tid; e < nr * nc; e += 256) {
      const int r = k + 1 + e / nc, ci = e % nc;
      const int c = (ci < f - k - 1) ? k + 1 + ci : fmax + (ci - (f - k - 1));
      const double ur = W[((long)k * ldw + c) * 2], ui = W[((long)k * ldw + c) * 2 + 1];
      double* x = W + ((long)r * ldw + c) * 2;
      x[0] -= lre[r] * ur - lim[r] * ui;
      x[1] -= lre[r] * ui + lim[r] * ur;
    }
    __syncthreads();
  }
  // back substitution, one thread per right-hand side
  for (int t = tid; t < T; t += 256) {
    for (int k = f - 1; k >= 0; --k) {
      double sr = W[((long)k * ldw + fmax + t) * 2], si = W[((long)k * ldw + fmax + t) * 2 + 1];
      for (int q = k + 1; q < f; ++q) {
        const double ur = W[((long)k * ldw + q) * 2], ui = W[((long)k * ldw + q) * 2 + 1];
        const double cr = W[((long)q * ldw + fmax + t) * 2], ci = W[((long)q * ldw + fmax + t) * 2 + 1];
        sr -= ur * cr - ui * ci;
        si -= ur * ci + ui * cr;
      }
      const double pr = W[((long)k * ldw + k) * 2], pi = W[((long)k * ldw + k) * 2 + 1];
      const double den = 1.0 / (pr * pr + pi * pi);
      W[((long)k * ldw + fmax + t) * 2] = (sr * pr + si * pi) * den;
      W[((long)k * ldw + fmax + t) * 2 + 1] = (si * pr - sr * pi) * den;
    }
  }
}
// X[:, t] += sum_kf X[:, T + kf] c[kf][t]  for the solution rows (npad) and the signal realisation S (N rows)
// The same solve with the whole system in LDS (f (f + T) complex entries: 134 KB at 77 flagged channels and 32
// times; the launch takes this form when it fits, k_wb_solve otherwise): LU with partial pivoting, right-looking, the
// right-hand sides swept along; the back substitution row-parallel (one barrier per unknown) instead of one thread
// per right-hand side -- with per-time units (T = 1) that was a single lane.  Same pivoting rule and operations as
// k_wb_solve.
__global__ __launch_bounds__(256) void k_wb_solve_lds(double* __restrict__ W_all, const int32_t* __restrict__ fcount,
                                                      const int fmax, const int T, int32_t* __restrict__ info,
                                                      const int iter_tag) {
  extern __shared__ double wl[];             // re [f][ldl] | im [f][ldl], ldl = f + T (+1 when even: bank spread)
  __shared__ double redv[4];
  __shared__ int redi[4], piv_s;
  const int b = blockIdx.x, f = fcount[b], ldw = fmax + T, tid = threadIdx.x;
  if (f == 0) return;
  double* W = W_all + (long)b * fmax * ldw * 2;
  const int nc = f + T, ldl = nc | 1;
  double* wr = wl;
  double* wi = wl + (size_t)f * ldl;
  // compact copy: columns 0 .. f-1 the matrix, f .. f+T-1 the right-hand sides (global columns fmax ..)
  for (int e = tid; e < f * nc; e += 256) {
    const int r = e / nc, c = e % nc;
    const int cg = (c < f) ? c : fmax + (c - f);
    wr[r * ldl + c] = W[((long)r * ldw + cg) * 2];
    wi[r * ldl + c] = W[((long)r * ldw + cg) * 2 + 1];
  }
  __syncthreads();
  for (int k = 0; k < f; ++k) {
    double best = -1.0;
    int at = k;
    for (int r = k + tid; r < f; r += 256) {
      const double a2 = wr[r * ldl + k] * wr[r * ldl + k] + wi[r * ldl + k] * wi[r * ldl + k];
      if (a2 > best) { best = a2; at = r; }
    }
    for (int o = 32; o > 0; o >>= 1) {
      const double ob = __shfl_xor(best, o, 64);
      const int oa = __shfl_xor(at, o, 64);
      if (ob > best || (ob == best && oa < at)) { best = ob; at = oa; }
    }
    if ((tid & 63) == 0) { redv[tid >> 6] = best; redi[tid >> 6] = at; }
    __syncthreads();
    if (tid == 0) {
      int w = 0;
      for (int q = 1; q < 4; ++q)
        if (redv[q] > redv[w] || (redv[q] == redv[w] && redi[q] < redi[w])) w = q;
      piv_s = redi[w];
      if (!(redv[w] > 1e-24)) atomicCAS(&info[b], 0, iter_tag);       // (see k_wb_solve)
    }
    __syncthreads();
    const int pv = piv_s;
    if (pv != k)
      for (int c = k + tid; c < nc; c += 256) {
        const double t0 = wr[k * ldl + c], t1 = wi[k * ldl + c];
        wr[k * ldl + c] = wr[pv * ldl + c]; wi[k * ldl + c] = wi[pv * ldl + c];
        wr[pv * ldl + c] = t0; wi[pv * ldl + c] = t1;
      }
    __syncthreads();
    const double pr = wr[k * ldl + k], pi = wi[k * ldl + k];
    const double den = 1.0 / (pr * pr + pi * pi);
    // rows below: the multiplier l_r = W[r][k] / W[k][k] is formed by every thread of the row for itself (the
    // column k entry is left alone until the step's barrier)
    const int ncu = nc - k - 1, nr = f - k - 1;
    for (int e = tid; e < nr * ncu; e += 256) {
      const int r = k + 1 + e / ncu, c = k + 1 + e % ncu;
      const double ar = wr[r * ldl + k], ai = wi[r * ldl + k];
      const double lre = (ar * pr + ai * pi) * den, lim = (ai * pr - ar * pi) * den;
      const double ur = wr[k * ldl + c], ui = wi[k * ldl + c];
      wr[r * ldl + c] -= lre * ur - lim * ui;
      wi[r * ldl + c] -= lre * ui + lim * ur;
    }
    __syncthreads();
  }
  // back substitution: unknown k of every right-hand side, then its column out of the rows above
  for (int k = f - 1; k >= 0; --k) {
    const double pr = wr[k * ldl + k], pi = wi[k * ldl + k];
    const double den = 1.0 / (pr * pr + pi * pi);
    for (int t = tid; t < T; t += 256) {
      const double sr = wr[k * ldl + f + t], si = wi[k * ldl + f + t];
      wr[k * ldl + f + t] = (sr * pr + si * pi) * den;
      wi[k * ldl + f + t] = (si * pr - sr * pi) * den;
    }
    __syncthreads();
    for (int e = tid; e < k * T; e += 256) {
      const int r = e / T, t = e % T;
      const double ur = wr[r * ldl + k], ui = wi[r * ldl + k];
      const double cr = wr[k * ldl + f + t], ci = wi[k * ldl + f + t];
      wr[r * ldl + f + t] -= ur * cr - ui * ci;
      wi[r * ldl + f + t] -= ur * ci + ui * cr;
    }
    __syncthreads();
  }
  for (int e = tid; e < f * T; e += 256) {
    const int r = e / T, t = e % T;
    W[((long)r * ldw + fmax + t) * 2] = wr[r * ldl + f + t];
    W[((long)r * ldw + fmax + t) * 2 + 1] = wi[r * ldl + f + t];
  }
}

__global__ __launch_bounds__(256) void k_wb_correct(double* __restrict__ Sre, double* __restrict__ Sim,
                                                    double* __restrict__ Xre, double* __restrict__ Xim,
                                                    const double* __restrict__ W_all,
                                                    const int32_t* __restrict__ fcount, const int fmax, const int T,
                                                    const int NP, const int TP, const int npad) {
  // out[r][t] += sum_k Y_P[r][k] c[k][t] over the rows of X and of S: a (rows x f) by (f x T) product per baseline, on
  // the matrix pipe (it had been a scalar loop per entry: 4.3 ms per iteration at the C3 shape with 77 flagged
  // channels).  One wave per 16-row tile; k beyond the baseline's own f contributes zeros on both sides (those
  // columns of X / S are never written).
  const int b = blockIdx.y, f = fcount[b], ldw = fmax + T;
  if (f == 0) return;
  const double* W = W_all + (long)b * fmax * ldw * 2;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, li = lane & 15, g = lane >> 4;
  const int ntile = (npad + NP) >> 4, nks = (f + 3) >> 2, ntt = (T + 15) >> 4;
  for (int rt = blockIdx.x * 4 + wave; rt < ntile; rt += gridDim.x * 4) {
    const int r0 = rt << 4;
    double* pr = (r0 < npad) ? Xre + ((long)b * npad + r0) * TP : Sre + ((long)b * NP + (r0 - npad)) * TP;
    double* pi = (r0 < npad) ? Xim + ((long)b * npad + r0) * TP : Sim + ((long)b * NP + (r0 - npad)) * TP;
    for (int tt = 0; tt < ntt; ++tt) {
      const int t = (tt << 4) + li;
      const bool tok = t < T;
      d4 ar = {0., 0., 0., 0.}, ai = ar;
      for (int ks = 0; ks < nks; ++ks) {
        const int k = 4 * ks + g;
        const bool kok = k < f;
        // A[m = li][k] = Y_P[r0 + li][k];  B[k][n = li] = c[k][t]
        const double yr = kok ? pr[(long)li * TP + T + k] : 0.0, yi = kok ? pi[(long)li * TP + T + k] : 0.0;
        const long wo = ((long)min(k, f - 1) * ldw + fmax + min(t, T - 1)) * 2;
        const double cr = (kok && tok) ? W[wo] : 0.0, ci = (kok && tok) ? W[wo + 1] : 0.0;
        ar = mfma64(yr, cr, ar);
        ar = mfma64(-yi, ci, ar);
        ai = mfma64(yr, ci, ai);
        ai = mfma64(yi, cr, ai);
      }
      if (tok) {
#pragma unroll
        for (int v = 0; v < 4; ++v) {               // accumulator: row g + 4 v, column li
          const long o = (long)HPX_ACC_ROW(g, v) * TP + t;
          pr[o] += ar[v];
          pi[o] += ai[v];
        }
      }
    }
  }
}

}  // namespace

// k_wb_solve in the form that fits: the system in LDS up to 150 KB
static int launch_wb_solve(int nbl, double* W, const int32_t* fcount, int fmax, int T, int32_t* info, int iter_tag,
                           hipStream_t st) {
  const size_t lds = (size_t)2 * fmax * ((fmax + T) | 1) * sizeof(double);
  if (lds <= (size_t)150 * 1024) {
    static hpx_lds_limit limit;
    HPX_TRY(limit.ensure(reinterpret_cast<const void*>(&k_wb_solve_lds), lds));
    hipLaunchKernelGGL(k_wb_solve_lds, dim3(nbl), dim3(256), lds, st, W, fcount, fmax, T, info, iter_tag);
  } else {
    hipLaunchKernelGGL(k_wb_solve, dim3(nbl), dim3(256), 0, st, W, fcount, fmax, T, info, iter_tag);
  }
  HPX_HIP(hipGetLastError());
  return HPX_OK;
}

// The correction on plan q's buffers: the f x (f + T) system from the model at the flagged channels (q->S holds U z of
// every right-hand-side column), its solve, and X <- Y_r + Y_P (I - Q^H Y_P)^-1 Q^H Y_r
int hpx_woodbury_correct(hpx_plan* q, int T, int iter_tag, hipStream_t st) {
  const int fm = q->wb_fmax, nbl = q->nbl;
  hipLaunchKernelGGL(k_wb_system, dim3(nbl), dim3(256), 0, st, q->Sre, q->Sim, q->Xre, q->Xim, q->Fre, q->Fim,
                     q->fg_shared, q->wb_flist, q->wb_fcount, q->wb_W, fm, q->N, q->M, T, q->NP, q->TP, q->npad);
  HPX_TRY(launch_wb_solve(nbl, q->wb_W, q->wb_fcount, fm, T, q->info, iter_tag, st));
  hipLaunchKernelGGL(k_wb_correct, dim3(8, nbl), dim3(256), 0, st, q->Sre, q->Sim, q->Xre, q->Xim, q->wb_W,
                     q->wb_fcount, fm, T, q->NP, q->TP, q->npad);
  HPX_HIP(hipGetLastError());
  return HPX_OK;
}

// Per-time units with a full noise matrix AND flagged channels: the transform of the unit solutions X = [Y_r | Y_P] of
// the unflagged-noise systems, then the correction per unit (one data column)
int hpx_child_woodbury(hpx_plan* c, int iter_tag, hipStream_t st) {
  const int N = c->N;
  HPX_TRY(hpx_launch_dft(c->nbl, c->NP, c->TP, c->Fopre, c->Fopim, 1, c->Xre, c->Xim, (long)c->npad * c->TP,
                         c->TP, nullptr, 0, c->Sre, c->Sim, (long)c->NP * c->TP, c->TP,
                         1.0 / sqrt((double)N), st, N == c->NP));
  return hpx_woodbury_correct(c, 1, iter_tag, st);
}
