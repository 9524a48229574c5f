// Structured solve for flagged baselines whose unflagged channels share one noise variance.
//
// With Ni = c diag(w), w the 0/1 channel mask, the circulant C = U^H Ni U is the flat-noise
// c I minus a rank-f term, f = number of flagged channels x_j:
//
//     C = c I - c Vf Vf^H,     Vf[k][j] = U^H[k][x_j] = Fop[k][x_j] / sqrt(N)      (N x f)
//
// so the scaled system of hpx_internal.h is diagonal plus a border of width M + f:
//
//     M_ss = D - c Vf Vf^H,  D = diag(c + N/ps)      <=>      [[ D      , Bd ],   Bd = [G | sqrt(c) Vf]
//                                                              [ Bd^H   , E  ]]   E  = blockdiag(H, I_f)
//
// (eliminating the f auxiliary unknowns y of the "I_f" block gives back D - c Vf Vf^H).  It is
// solved through the (M + f) x (M + f) Schur complement, Hermitian positive definite because
// D > c and Vf has orthonormal columns:
//
//     S  = E - Bd^H Dinv Bd,   Rf = [P4; 0] - Bd^H Dinv r1,   [f; y] = S^-1 Rf,   z = Dinv (r1 - Bd [f; y])
//
// O(N (M+f) (M+f+T)) instead of O(N^3): at 15 % flags a tenth of the dense flops.  The reference
// driver's default noise model (Ninv = I / 100 when no noise covariance is given,
// run-hydra-pspec.py:436-438) together with data flags is exactly this case.
//
// k_lr_schur forms S and Rf on the f64 MFMA (one workgroup per baseline, tiles over waves) and
// writes them in the factor layout; the small dense system goes through the batched Cholesky /
// back substitution of hpx_factor.hip; k_lr_back forms z.  X = [z; f] comes out in the layout
// k_backsolve produces, everything downstream is shared with the other solvers.
#include "hpx_internal.h"

namespace {

struct LrArgs {
  const double *ia, *cre, *rre, *rim, *p2re, *p2im, *hre, *him, *p4re, *p4im, *fopre, *fopim;
  const int32_t *flist, *fcount;     // [nbl][fmax] flagged channels, [nbl] their number
  const double* cval;                // [nbl] inverse noise variance of the unflagged channels
  const double *bre, *bim;           // [nbl][NP][npadS] the border [G | sqrt(c) Vf | 0], planar
  const double *tre, *tim;           // [nbl][npadS][NP] its transpose
  double* Ls;                        // [nbl] small system in the factor layout (npadS, ldS)
  const double *Yre, *Yim;           // [nbl][npadS][TP] its solution
  double *Xre, *Xim;
  int N, M, NP, TP, ncol, npad, has_omega, fmax, npadS, ldS;
  double isn;
};

// element (k, col) of the border Bd = [G | sqrt(c) Vf | 0]; k is a valid channel index
__device__ __forceinline__ void border(const LrArgs& A, const double* __restrict__ rre,
                                       const double* __restrict__ rim, const int* fl, const int fcnt,
                                       const double sc, const int k, const int col, double& vr, double& vi) {
  vr = 0.0;
  vi = 0.0;
  if (col < A.M) {
    vr = rre[(long)k * A.ncol + A.TP + col];
    vi = rim[(long)k * A.ncol + A.TP + col];
  } else if (col - A.M < fcnt) {
    const int x = fl[col - A.M];
    vr = sc * A.fopre[(long)k * A.NP + x];
    vi = sc * A.fopim[(long)k * A.NP + x];
  }
}

// The border does not change along the chain: it is laid out once, planar, [NP][npadS] per baseline
// (unit stride along the border index for both MFMA operands of k_lr_schur).
__global__ void k_lr_border(const LrArgs A, double* __restrict__ bre, double* __restrict__ bim,
                            double* __restrict__ tre, double* __restrict__ tim) {
  const int b = blockIdx.y, N = A.N, NP = A.NP, npadS = A.npadS;
  const double* rre = A.rre + (long)b * NP * A.ncol;
  const double* rim = A.rim + (long)b * NP * A.ncol;
  const int* fl = A.flist + (long)b * A.fmax;
  const int fcnt = A.fcount[b];
  const double sc = sqrt(A.cval[b]) * A.isn;
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < (long)NP * npadS; e += (long)gridDim.x * blockDim.x) {
    const int k = (int)(e / npadS), col = (int)(e % npadS);
    double vr = 0.0, vi = 0.0;
    if (k < N) border(A, rre, rim, fl, fcnt, sc, k, col, vr, vi);
    bre[(long)b * NP * npadS + e] = vr;
    bim[(long)b * NP * npadS + e] = vi;
    tre[(long)b * NP * npadS + (long)col * NP + k] = vr;      // transposed copy [npadS][NP] for k_lr_back
    tim[(long)b * NP * npadS + (long)col * NP + k] = vi;
  }
}

constexpr int LR_RT = 2, LR_CT = 3;    // row / column tiles per work item of k_lr_schur

// S = E - Bd^H Dinv Bd (lower tiles) and Rf = [P4; 0] - Bd^H Dinv r1, written in the factor
// layout.  A work item is a 2 x 3 block of 16 x 16 output tiles accumulated over all channels
// by one wave (5 operand tiles per k-step for 24 MFMAs), operands of the next k-step in flight
// while the current one is multiplied.
__global__ __launch_bounds__(256, 2) void k_lr_schur(const LrArgs A) {
  extern __shared__ double lds[];
  const int b = blockIdx.x, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int li = lane & 15, g = lane >> 4;
  const int N = A.N, M = A.M, NP = A.NP, TP = A.TP, ncol = A.ncol, TT = TP >> 4;
  const int npadS = A.npadS, mt = npadS >> 4;
  double* dinv = lds;
  double* iav = dinv + NP;
  const double* ia = A.ia + (long)b * N;
  const double* rre = A.rre + (long)b * NP * ncol;
  const double* rim = A.rim + (long)b * NP * ncol;
  const double* bre = A.bre + (long)b * NP * npadS;
  const double* bim = A.bim + (long)b * NP * npadS;
  const double c0 = A.cval[b];
  for (int k = tid; k < NP; k += 256) {
    const double v = (k < N) ? ia[k] : 0.0;
    iav[k] = v;
    dinv[k] = (k < N) ? 1.0 / fma(v, v, c0) : 0.0;
  }
  __syncthreads();
  double* L = A.Ls + (long)b * npadS * A.ldS * 2;
  const int nks = NP >> 2;                             // NP is a multiple of 16: nks is a multiple of 4
  int item = 0;
  for (int r0 = 0; r0 < mt; r0 += LR_RT) {
    const int nrt = min(LR_RT, mt - r0);
    const int rlast = r0 + nrt - 1;
    const int ncc = rlast + 1 + TT;                   // S tiles 0..rlast, then the Rf tiles
    for (int cb = 0; cb < ncc; cb += LR_CT, ++item) {
      if ((item & 3) != wave) continue;
      const int nct = min(LR_CT, ncc - cb);
      d4 ar[LR_RT][LR_CT], ai[LR_RT][LR_CT];
#pragma unroll
      for (int t = 0; t < LR_RT; ++t)
#pragma unroll
        for (int q = 0; q < LR_CT; ++q) {
          ar[t][q] = (d4){0., 0., 0., 0.};
          ai[t][q] = (d4){0., 0., 0., 0.};
        }
      double a0r[LR_RT], a0i[LR_RT], a1r[LR_RT], a1i[LR_RT];
      double b0r[LR_CT], b0i[LR_CT], b1r[LR_CT], b1i[LR_CT], d0, d1;
#define HPX_LR_LOAD(ar_, ai_, br_, bi_, dk_, ks_)                                              \
  {                                                                                            \
    const int k_ = 4 * (ks_) + g;                        /* rows >= N of the border are zero */ \
    const long ro_ = (long)k_ * npadS;                                                         \
    dk_ = dinv[k_];                                                                            \
    _Pragma("unroll") for (int t = 0; t < LR_RT; ++t) {                                        \
      const int rt_ = min(r0 + t, mt - 1);                                                     \
      ar_[t] = bre[ro_ + 16 * rt_ + li];                                                       \
      ai_[t] = bim[ro_ + 16 * rt_ + li];                                                       \
    }                                                                                          \
    _Pragma("unroll") for (int q = 0; q < LR_CT; ++q) {                                        \
      const int cc_ = min(cb + q, ncc - 1);                                                    \
      if (cc_ <= rlast) {                                                                      \
        br_[q] = bre[ro_ + 16 * cc_ + li];                                                     \
        bi_[q] = bim[ro_ + 16 * cc_ + li];                                                     \
      } else {                                                                                 \
        const int kc_ = min(k_, N - 1);                                                        \
        const int t_ = ((cc_ - rlast - 1) << 4) + li;                                          \
        double x_ = rre[(long)kc_ * ncol + t_], y_ = rim[(long)kc_ * ncol + t_];               \
        if (A.has_omega) {                                                                     \
          x_ = fma(iav[k_], A.p2re[(long)kc_ * TP + t_], x_);                                  \
          y_ = fma(iav[k_], A.p2im[(long)kc_ * TP + t_], y_);                                  \
        }                                                                                      \
        br_[q] = x_;                                                                           \
        bi_[q] = y_;                                                                           \
      }                                                                                        \
    }                                                                                          \
  }
#define HPX_LR_MMA(ar_, ai_, br_, bi_, dk_)                                                    \
  _Pragma("unroll") for (int q = 0; q < LR_CT; ++q) {                                          \
    const double x_ = dk_ * br_[q], y_ = dk_ * bi_[q];       /* B = Dinv_k (border | r1) */     \
    _Pragma("unroll") for (int t = 0; t < LR_RT; ++t) {      /* A = conj(border) */             \
      ar[t][q] = mfma64(ar_[t], x_, ar[t][q]);                                                 \
      ar[t][q] = mfma64(ai_[t], y_, ar[t][q]);                                                 \
      ai[t][q] = mfma64(ar_[t], y_, ai[t][q]);                                                 \
      ai[t][q] = mfma64(-ai_[t], x_, ai[t][q]);                                                \
    }                                                                                          \
  }
      HPX_LR_LOAD(a0r, a0i, b0r, b0i, d0, 0)
      for (int ks = 0; ks < nks; ks += 2) {
        HPX_LR_LOAD(a1r, a1i, b1r, b1i, d1, ks + 1)
        __builtin_amdgcn_sched_barrier(0);
        HPX_LR_MMA(a0r, a0i, b0r, b0i, d0)
        __builtin_amdgcn_sched_barrier(0);
        HPX_LR_LOAD(a0r, a0i, b0r, b0i, d0, min(ks + 2, nks - 1))
        __builtin_amdgcn_sched_barrier(0);
        HPX_LR_MMA(a1r, a1i, b1r, b1i, d1)
        __builtin_amdgcn_sched_barrier(0);
      }
#undef HPX_LR_LOAD
#undef HPX_LR_MMA
      // lane (li, g), register v holds row m = 16 (r0 + t) + g + 4v, column 16 cc + li
#pragma unroll
      for (int t = 0; t < LR_RT; ++t) {
        if (t >= nrt) break;
        const int ri = r0 + t;
#pragma unroll
        for (int q = 0; q < LR_CT; ++q) {
          if (q >= nct) break;
          const int cc = cb + q;
          if (cc <= rlast && cc > ri) continue;        // above the diagonal: not needed
#pragma unroll
          for (int v = 0; v < 4; ++v) {
            const int m = 16 * ri + HPX_ACC_ROW(g, v);
            if (cc <= rlast) {                         // S = E - acc
              const int mc = 16 * cc + li;
              double e_r = 0.0, e_i = 0.0;
              if (m < M && mc < M) {
                e_r = A.hre[(long)b * M * M + m * M + mc];
                e_i = A.him[(long)b * M * M + m * M + mc];
              } else if (m == mc) {
                e_r = 1.0;
              }
              const long o = HPX_LIDX(m, mc, npadS);
              L[o] = e_r - ar[t][q][v];
              L[o + 16] = e_i - ai[t][q][v];
            } else {                                   // row npadS + t' of the factor buffer = conj(Rf[m][t'])
              const int tc = ((cc - rlast - 1) << 4) + li;
              double e_r = 0.0, e_i = 0.0;
              if (m < M) {
                e_r = A.p4re[(long)b * M * TP + m * TP + tc];
                e_i = A.p4im[(long)b * M * TP + m * TP + tc];
              }
              const long o = HPX_LIDX(npadS + tc, m, npadS);
              L[o] = e_r - ar[t][q][v];
              L[o + 16] = -(e_i - ai[t][q][v]);
            }
          }
        }
      }
    }
  }
}

__global__ __launch_bounds__(256) void k_lr_back(const LrArgs A) {
  extern __shared__ double lds[];
  const int b = blockIdx.x, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int li = lane & 15, g = lane >> 4;
  const int N = A.N, M = A.M, NP = A.NP, TP = A.TP, ncol = A.ncol, TT = TP >> 4;
  double* dinv = lds;
  double* iav = dinv + NP;
  const double* ia = A.ia + (long)b * N;
  const double* rre = A.rre + (long)b * NP * ncol;
  const double* rim = A.rim + (long)b * NP * ncol;
  const double c0 = A.cval[b];
  const int fcnt = A.fcount[b];
  const double* tre = A.tre + (long)b * NP * A.npadS;
  const double* tim = A.tim + (long)b * NP * A.npadS;
  for (int k = tid; k < NP; k += 256) {
    const double v = (k < N) ? ia[k] : 0.0;
    iav[k] = v;
    dinv[k] = (k < N) ? 1.0 / fma(v, v, c0) : 0.0;
  }
  __syncthreads();
  const double* yre = A.Yre + (long)b * A.npadS * TP;
  const double* yim = A.Yim + (long)b * A.npadS * TP;
  double* Xre = A.Xre + (long)b * A.npad * TP;
  double* Xim = A.Xim + (long)b * A.npad * TP;
  const int nms = (M + fcnt + 3) >> 2;                // k-steps over the live border columns
  for (int kt = wave; kt < (NP >> 4); kt += 4) {
    const int k0 = kt << 4;
    for (int tt = 0; tt < TT; ++tt) {
      const int t = (tt << 4) + li;
      d4 zr, zi;
#pragma unroll
      for (int v = 0; v < 4; ++v) {                   // r1[k][t], k = k0 + g + 4v
        const int k = min(k0 + HPX_ACC_ROW(g, v), N - 1);
        double r_r = rre[(long)k * ncol + t], r_i = rim[(long)k * ncol + t];
        if (A.has_omega) {
          r_r = fma(iav[k], A.p2re[(long)k * TP + t], r_r);
          r_i = fma(iav[k], A.p2im[(long)k * TP + t], r_i);
        }
        zr[v] = r_r;
        zi[v] = r_i;
      }
      for (int ms = 0; ms < nms; ++ms) {
        const int m = 4 * ms + g;
        const double a_r = -tre[(long)m * NP + k0 + li];      // A[k = k0 + li][m] = -Bd[k][m] (unit stride in k)
        const double a_i = -tim[(long)m * NP + k0 + li];
        const double f_r = yre[(long)m * TP + t], f_i = yim[(long)m * TP + t];   // B[m][t] = Y[m][t]
        zr = mfma64(a_r, f_r, zr);
        zr = mfma64(-a_i, f_i, zr);
        zi = mfma64(a_r, f_i, zi);
        zi = mfma64(a_i, f_r, zi);
      }
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        const int k = k0 + HPX_ACC_ROW(g, v);
        if (k < N) {
          Xre[(long)k * TP + t] = zr[v] * dinv[k];
          Xim[(long)k * TP + t] = zi[v] * dinv[k];
        }
      }
    }
  }
  for (int e = tid; e < (A.npad - N) * TP; e += 256) {
    const int m = e / TP, t = e - m * TP;
    Xre[(long)(N + m) * TP + t] = (m < M) ? yre[(long)m * TP + t] : 0.0;
    Xim[(long)(N + m) * TP + t] = (m < M) ? yim[(long)m * TP + t] : 0.0;
  }
}

}  // namespace

static void lr_args(hpx_plan* p, LrArgs& A) {
  A.ia = p->ia; A.cre = p->Cre; A.rre = p->Rre; A.rim = p->Rim; A.p2re = p->P2re; A.p2im = p->P2im;
  A.hre = p->Hre; A.him = p->Him; A.p4re = p->P4re; A.p4im = p->P4im;
  A.fopre = p->Fopre; A.fopim = p->Fopim;
  A.flist = p->lr_flist; A.fcount = p->lr_fcount; A.cval = p->lr_c; A.Ls = p->lr_L;
  A.bre = p->lr_Bre; A.bim = p->lr_Bim; A.tre = p->lr_Tre; A.tim = p->lr_Tim; A.Yre = p->lr_Yre; A.Yim = p->lr_Yim;
  A.Xre = p->Xre; A.Xim = p->Xim;
  A.N = p->N; A.M = p->M; A.NP = p->NP; A.TP = p->TP; A.ncol = p->ncolR; A.npad = p->npad;
  A.has_omega = p->has_omega; A.fmax = p->lr_fmax; A.npadS = p->lr_npad; A.ldS = p->lr_npad + p->TP;
  A.isn = 1.0 / sqrt((double)p->N);
}

// the iteration-invariant border of every baseline (after hpx_plan_set_solver filled the lists)
int hpx_lowrank_prepare(hpx_plan* p, hipStream_t st) {
  LrArgs A;
  lr_args(p, A);
  hipLaunchKernelGGL(k_lr_border, dim3(64, p->nbl), dim3(256), 0, st, A, p->lr_Bre, p->lr_Bim, p->lr_Tre, p->lr_Tim);
  HPX_HIP(hipGetLastError());
  HPX_HIP(hipStreamSynchronize(st));
  return HPX_OK;
}

int hpx_launch_solve_lowrank(hpx_plan* p, int iter_tag, hipStream_t st) {
  LrArgs A;
  lr_args(p, A);
  const size_t lds = (size_t)2 * p->NP * sizeof(double);
  hipLaunchKernelGGL(k_lr_schur, dim3(p->nbl), dim3(256), lds, st, A);
  HPX_HIP(hipGetLastError());
  HPX_TRY(hpx_launch_factor(p->nbl, p->lr_npad, p->lr_npad + p->TP, p->lr_L, p->lr_Wre, p->lr_Wim, p->info,
                            iter_tag, nullptr, st));
  HPX_TRY(hpx_launch_backsolve(p->nbl, p->lr_npad, p->TP, p->lr_npad + p->TP, p->lr_L, p->lr_Wre, p->lr_Wim,
                               p->lr_Yre, p->lr_Yim, st));
  hipLaunchKernelGGL(k_lr_back, dim3(p->nbl), dim3(256), lds, st, A);
  HPX_HIP(hipGetLastError());
  return HPX_OK;
}
