// Structured solve for baselines with flat noise and no flags.
//
// With Ni = c I (no flagged channel, the same noise variance in every channel) the circulant
// C = U^H Ni U is c I, and the scaled system of hpx_internal.h becomes diagonal + a rank-M border:
//
//     M = [[ diag(c + N/ps) , G ],      [z; f] = M^-1 [r1; r2],   r1 = Q + P2/a,  r2 = P4
//          [ G^H            , H ]]
//
// solved exactly through the M x M Schur complement, O(N M (M + T)) instead of O(N^3):
//
//     Dinv = 1 / (c + N/ps)
//     S  = H  - G^H Dinv G          (M x M)
//     Rf = r2 - G^H Dinv r1         (M x T)
//     f  = S^-1 Rf
//     z  = Dinv (r1 - G f)
//
// This is the reference's own test configuration (test_data: noise-cov.npy = sigma^2 I, no
// flags) and SURVEY 8(d)'s unflagged synthetic recipe.  One workgroup per baseline; the two
// contractions over the N channels run on the f64 MFMA (K split over the 4 waves, fixed-order
// reduction through LDS), the M x M solve is a Gauss-Jordan elimination in LDS.  The kernel
// writes X = [z; f] in the layout k_backsolve produces, so everything downstream is shared
// with the dense path.  Selected by hpx_plan_set_solver(); results agree with the dense
// path to rounding (tests/test_gpu_chain.py).
#include "hpx_internal.h"

namespace {

#ifndef HPX_FLAT_DIAG
#define HPX_FLAT_DIAG 0            // timing-only ablations (wrong results): 1 no Gauss-Jordan, 2 no back product, 4 no contraction
#endif
#ifndef HPX_FT_MAX
#define HPX_FT_MAX 2
#endif
#ifndef HPX_FLAT_WGS
#define HPX_FLAT_WGS 1              // workgroups per CU the register budget is set for
#endif
constexpr int FT_MAX = HPX_FT_MAX; // t-tiles per pass of the first contraction

struct FlatArgs {
  const double *ia, *cre, *rre, *rim, *p2re, *p2im, *hre, *him, *p4re, *p4im;
  double *Xre, *Xim;
  int32_t* info;
  int N, M, NP, TP, ncol, npad, has_omega, iter_tag;
  // low-rank solver, FFT form: the operands of the contraction are also the input of its
  // transforms, [nbl][NP][CP + TP] planar = [dinv | dinv conj(G) | 0 | dinv r1]  (NULL: not written)
  double *xin_re, *xin_im;
  int CP;
};

// S = H - G^H Dinv G (16 x 16, identity padding beyond M) and Rf = P4 - G^H Dinv r1 (16 x TP) of
// baseline b into the LDS matrix sre/sim (row length SW = 16 + TP: [S | Rf]).  Also fills dinv
// and iav.  Shared by the flat-noise solver and by the FFT form of the low-rank solver, whose
// Schur complement has these as its foreground block.
template <bool XIN>
__device__ __forceinline__ void flat_blocks(const FlatArgs& A, const int b, double* dinv, double* iav,
                                            double* slab, double* sre, double* sim, const double c0) {
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int li = lane & 15, g = lane >> 4;
  const int N = A.N, M = A.M, NP = A.NP, TP = A.TP, ncol = A.ncol, TT = TP >> 4;
  const int SW = 16 + TP, XW = A.CP + TP;
  const double* ia = A.ia + (long)b * N;
  const double* rre = A.rre + (long)b * NP * ncol;
  const double* rim = A.rim + (long)b * NP * ncol;
  double* xin_r = XIN ? A.xin_re + (long)b * NP * XW : nullptr;      // XIN: also write the transform input
  double* xin_i = XIN ? A.xin_im + (long)b * NP * XW : nullptr;
  for (int k = tid; k < NP; k += 256) {
    const double v = (k < N) ? ia[k] : 0.0;
    iav[k] = v;
    dinv[k] = (k < N) ? 1.0 / fma(v, v, c0) : 0.0;
  }
  __syncthreads();

  // ---- S = H - G^H Dinv G  and  Rf = P4 - G^H Dinv r1, t-tiles in passes of FT_MAX
  const int nks = NP >> 2;
  for (int tb = 0; tb < TT; tb += FT_MAX) {
    const int nt = min(FT_MAX, TT - tb);
    const bool with_s = (tb == 0);
    d4 ar[1 + FT_MAX], ai[1 + FT_MAX];
#pragma unroll
    for (int q = 0; q < 1 + FT_MAX; ++q) {
      ar[q] = (d4){0., 0., 0., 0.};
      ai[q] = (d4){0., 0., 0., 0.};
    }
    // This wave's k-steps (ks = wave + 4 i) in chunks of KCH: the operands of the next chunk are in
    // flight while the current one is multiplied (one k-step is 12 MFMAs, a third of a memory round
    // trip: a chunk of one left every step waiting).  Loads are unconditional (clamped indices; P2
    // is zero without omega): a load under a runtime condition makes hipcc branch around it and
    // drain the queue.  Steps past the end carry Dinv = 0.
    constexpr int KCH = 2;
    const int nmy = (nks > wave) ? (nks - wave + 3) / 4 : 0;
    const int nchk = (nmy + KCH - 1) / KCH;
    double gr[2][KCH], gi[2][KCH], dk[2][KCH], br[2][KCH][FT_MAX], bi[2][KCH][FT_MAX];
    int kq[2][KCH];                                   // channel 4 ks + g of the step, or -1
#define HPX_FL_LOAD(S, creq_)                                                              \
  {                                                                                        \
    const int cl_ = min((creq_), nchk - 1);                                                \
    _Pragma("unroll") for (int u = 0; u < KCH; ++u) {                                      \
      const int i_ = KCH * cl_ + u;                                                        \
      const bool ok_ = ((creq_) < nchk) && (i_ < nmy);                                     \
      const int k_ = 4 * (wave + 4 * min(i_, nmy - 1)) + g;                                \
      const int kc_ = min(k_, N - 1);        /* padded channels: Dinv = 0, finite operand */ \
      const long ro_ = (long)kc_ * ncol;                                                   \
      gr[S][u] = rre[ro_ + TP + li];                                                       \
      gi[S][u] = rim[ro_ + TP + li];                                                       \
      dk[S][u] = ok_ ? dinv[k_] : 0.0;                                                     \
      kq[S][u] = ok_ ? k_ : -1;                                                            \
      const double ik_ = iav[k_];                                                          \
      _Pragma("unroll") for (int q = 0; q < FT_MAX; ++q) {                                 \
        const int t_ = ((tb + min(q, nt - 1)) << 4) + li;                                  \
        br[S][u][q] = fma(ik_, A.p2re[(long)kc_ * TP + t_], rre[ro_ + t_]);                \
        bi[S][u][q] = fma(ik_, A.p2im[(long)kc_ * TP + t_], rim[ro_ + t_]);                \
      }                                                                                    \
    }                                                                                      \
  }
#define HPX_FL_MMA(S)                                                                      \
  _Pragma("unroll") for (int u = 0; u < KCH; ++u) {                                        \
    const long xo_ = (long)max(kq[S][u], 0) * XW;                                          \
    const double a_r = gr[S][u], a_i = -gi[S][u];   /* A[m = li][k] = conj(G[k][m]) */     \
    if (with_s) {                            /* B[k][m' = li] = Dinv_k G[k][m'] */         \
      const double b_r = dk[S][u] * gr[S][u], b_i = dk[S][u] * gi[S][u];                   \
      if (XIN && kq[S][u] >= 0) {                                                          \
        if (li < M) { xin_r[xo_ + 1 + li] = b_r; xin_i[xo_ + 1 + li] = -b_i; }             \
        if (li == 0) { xin_r[xo_] = dk[S][u]; xin_i[xo_] = 0.0; }                          \
      }                                                                                    \
      ar[0] = mfma64(a_r, b_r, ar[0]);                                                     \
      ar[0] = mfma64(-a_i, b_i, ar[0]);                                                    \
      ai[0] = mfma64(a_r, b_i, ai[0]);                                                     \
      ai[0] = mfma64(a_i, b_r, ai[0]);                                                     \
    }                                                                                      \
    _Pragma("unroll") for (int q = 0; q < FT_MAX; ++q) {                                   \
      if (q < nt) {                          /* B[k][t] = Dinv_k r1[k][t] */               \
        const double b_r = dk[S][u] * br[S][u][q], b_i = dk[S][u] * bi[S][u][q];           \
        if (XIN && kq[S][u] >= 0) {                                                        \
          xin_r[xo_ + A.CP + ((tb + q) << 4) + li] = b_r;                                  \
          xin_i[xo_ + A.CP + ((tb + q) << 4) + li] = b_i;                                  \
        }                                                                                  \
        ar[1 + q] = mfma64(a_r, b_r, ar[1 + q]);                                           \
        ar[1 + q] = mfma64(-a_i, b_i, ar[1 + q]);                                          \
        ai[1 + q] = mfma64(a_r, b_i, ai[1 + q]);                                           \
        ai[1 + q] = mfma64(a_i, b_r, ai[1 + q]);                                           \
      }                                                                                    \
    }                                                                                      \
  }
    if (nmy > 0 && !(HPX_FLAT_DIAG & 4)) {
      HPX_FL_LOAD(0, 0)
      for (int c = 0; c < nchk; c += 2) {
        HPX_FL_LOAD(1, c + 1)
        __builtin_amdgcn_sched_barrier(0);
        HPX_FL_MMA(0)
        __builtin_amdgcn_sched_barrier(0);
        HPX_FL_LOAD(0, c + 2)
        __builtin_amdgcn_sched_barrier(0);
        HPX_FL_MMA(1)
        __builtin_amdgcn_sched_barrier(0);
      }
    }
#undef HPX_FL_LOAD
#undef HPX_FL_MMA
    // fixed-order sum of the four waves' partial tiles through ONE slab (waves 3, 2, 1, 0 in
    // turn): a slab per wave would cost 64 KB of LDS and leave one workgroup per CU
    for (int w = 3; w >= 0; --w) {
      if (wave == w) {
#pragma unroll
        for (int q = 0; q < 1 + FT_MAX; ++q)
#pragma unroll
          for (int v = 0; v < 4; ++v) {
            if (w == 3) {
              slab[q * 512 + v * 64 + lane] = ar[q][v];
              slab[q * 512 + 256 + v * 64 + lane] = ai[q][v];
            } else {
              slab[q * 512 + v * 64 + lane] += ar[q][v];
              slab[q * 512 + 256 + v * 64 + lane] += ai[q][v];
            }
          }
      }
      __syncthreads();
    }
    // element (q, v, l) is row m = (l>>4)+4v, column l&15 of tile q
    for (int e = tid; e < (1 + nt) * 256; e += 256) {
      const int q = e >> 8, v = (e >> 6) & 3, l = e & 63;
      if (q == 0 && !with_s) continue;
      const double s_r = slab[q * 512 + v * 64 + l], s_i = slab[q * 512 + 256 + v * 64 + l];
      const int m = HPX_ACC_ROW(l >> 4, v), col = l & 15;
      double o_r, o_i;
      if (q == 0) {                               // S, identity padding beyond M
        if (m < M && col < M) {
          o_r = A.hre[(long)b * M * M + m * M + col] - s_r;
          o_i = A.him[(long)b * M * M + m * M + col] - s_i;
        } else {
          o_r = (m == col) ? 1.0 : 0.0;
          o_i = 0.0;
        }
        sre[m * SW + col] = o_r;
        sim[m * SW + col] = o_i;
      } else {
        const int t = ((tb + q - 1) << 4) + col;
        if (m < M) {
          o_r = A.p4re[(long)b * M * TP + m * TP + t] - s_r;
          o_i = A.p4im[(long)b * M * TP + m * TP + t] - s_i;
        } else {
          o_r = 0.0;
          o_i = 0.0;
        }
        sre[m * SW + 16 + t] = o_r;
        sim[m * SW + 16 + t] = o_i;
      }
    }
    __syncthreads();
  }

}

__global__ __launch_bounds__(256, HPX_FLAT_WGS) void k_solve_flat(const FlatArgs A) {
  extern __shared__ double lds[];
  const int b = blockIdx.x, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int li = lane & 15, g = lane >> 4;
  const int N = A.N, M = A.M, NP = A.NP, TP = A.TP, ncol = A.ncol, TT = TP >> 4;
  const int SW = 16 + TP;                         // row length of the augmented M x M system
  double* dinv = lds;                             // [NP]
  double* iav = dinv + NP;                        // [NP]
  double* slab = iav + NP;                        // [1 + FT_MAX][2][4][64]
  double* sre = slab + (1 + FT_MAX) * 2 * 256;    // [16][SW]
  double* sim = sre + 16 * SW;
  __shared__ int bad_s;
  const double* rre = A.rre + (long)b * NP * ncol;
  const double* rim = A.rim + (long)b * NP * ncol;
  if (tid == 0) bad_s = 0;
  flat_blocks<false>(A, b, dinv, iav, slab, sre, sim, A.cre[(long)b * N]);

  // ---- f = S^-1 Rf: Gauss-Jordan on the augmented 16 x (16 + TP) system (S Hermitian
  // positive definite: no pivoting)
  double* prow_r = slab;                           // scaled pivot row / pivot column of a step
  double* prow_i = prow_r + SW;
  double* pcol_r = prow_i + SW;
  double* pcol_i = pcol_r + 16;
  for (int k = (HPX_FLAT_DIAG & 1) ? 16 : 0; k < 16; ++k) {
    const double piv = sre[k * SW + k];
    if (tid == 0 && !(piv > 0.0)) bad_s = 1;
    const double rinv = 1.0 / piv;
    for (int j = tid; j < SW; j += 256) {
      prow_r[j] = sre[k * SW + j] * rinv;
      prow_i[j] = sim[k * SW + j] * rinv;
    }
    if (tid < 16) {
      pcol_r[tid] = sre[tid * SW + k];
      pcol_i[tid] = sim[tid * SW + k];
    }
    __syncthreads();
    for (int e = tid; e < 16 * SW; e += 256) {
      const int i = e / SW, j = e - i * SW;
      const double pr = prow_r[j], pi = prow_i[j];
      if (i == k) {
        sre[e] = pr;
        sim[e] = pi;
      } else {
        const double fr = pcol_r[i], fi = pcol_i[i];
        sre[e] -= fr * pr - fi * pi;
        sim[e] -= fr * pi + fi * pr;
      }
    }
    __syncthreads();
  }

  // ---- z = Dinv (r1 - G f), written with f and the zero padding as X = [z; f; 0]
  double* Xre = A.Xre + (long)b * A.npad * TP;
  double* Xim = A.Xim + (long)b * A.npad * TP;
  for (int kt = (HPX_FLAT_DIAG & 2) ? (NP >> 4) : wave; kt < (NP >> 4); kt += 4) {
    const int k0 = kt << 4;
    // A[k = k0 + li][m = 4 ks + g] = -G[k][m]
    double ga_r[4], ga_i[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const long o = (long)min(k0 + li, N - 1) * ncol + TP + 4 * ks + g;
      ga_r[ks] = -rre[o];                          // columns M .. 15 of G are zero padding (k_prep)
      ga_i[ks] = -rim[o];
    }
    // two t-tiles at a time: all loads of the pair are issued before the first use, and none sits
    // under a runtime condition (P2 is zero without omega)
    for (int tt = 0; tt < TT; tt += 2) {
      d4 zr[2], zi[2];
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const int t = (min(tt + q, TT - 1) << 4) + li;
#pragma unroll
        for (int v = 0; v < 4; ++v) {             // start from r1[k][t], k = k0 + g + 4v
          const int k = min(k0 + HPX_ACC_ROW(g, v), N - 1);
          zr[q][v] = fma(iav[k], A.p2re[(long)k * TP + t], rre[(long)k * ncol + t]);
          zi[q][v] = fma(iav[k], A.p2im[(long)k * TP + t], rim[(long)k * ncol + t]);
        }
      }
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const int t = (min(tt + q, TT - 1) << 4) + li;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {          // B[m = 4 ks + g][t] = f[m][t]
          const int m = 4 * ks + g;
          const double f_r = sre[m * SW + 16 + t], f_i = sim[m * SW + 16 + t];
          zr[q] = mfma64(ga_r[ks], f_r, zr[q]);
          zr[q] = mfma64(-ga_i[ks], f_i, zr[q]);
          zi[q] = mfma64(ga_r[ks], f_i, zi[q]);
          zi[q] = mfma64(ga_i[ks], f_r, zi[q]);
        }
      }
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        if (tt + q >= TT) continue;
        const int t = ((tt + q) << 4) + li;
#pragma unroll
        for (int v = 0; v < 4; ++v) {
          const int k = k0 + HPX_ACC_ROW(g, v);
          if (k < N) {
            Xre[(long)k * TP + t] = zr[q][v] * dinv[k];
            Xim[(long)k * TP + t] = zi[q][v] * dinv[k];
          }
        }
      }
    }
  }
  for (int e = tid; e < (A.npad - N) * TP; e += 256) {
    const int m = e / TP, t = e - m * TP;
    Xre[(long)(N + m) * TP + t] = (m < M) ? sre[m * SW + 16 + t] : 0.0;
    Xim[(long)(N + m) * TP + t] = (m < M) ? sim[m * SW + 16 + t] : 0.0;
  }
  if (tid == 0 && bad_s && A.info) atomicCAS(&A.info[b], 0, A.iter_tag);
}

// The blocks alone, to global memory: out[b][16][16 + TP] planar (low-rank solver, FFT form).
__global__ __launch_bounds__(256, HPX_FLAT_WGS) void k_flat_blocks(const FlatArgs A, const double* __restrict__ cval,
                                                     double* __restrict__ ore, double* __restrict__ oim) {
  extern __shared__ double lds[];
  const int b = blockIdx.x, NP = A.NP, SW = 16 + A.TP;
  double* dinv = lds;
  double* iav = dinv + NP;
  double* slab = iav + NP;
  double* sre = slab + (1 + FT_MAX) * 2 * 256;
  double* sim = sre + 16 * SW;
  flat_blocks<true>(A, b, dinv, iav, slab, sre, sim, cval[b]);
  for (int e = threadIdx.x; e < 16 * SW; e += 256) {
    ore[(long)b * 16 * SW + e] = sre[e];
    oim[(long)b * 16 * SW + e] = sim[e];
  }
}

}  // namespace

size_t hpx_flat_lds_bytes(const hpx_plan* p) {
  return ((size_t)2 * p->NP + (size_t)(1 + FT_MAX) * 512 + (size_t)2 * 16 * (16 + p->TP)) * sizeof(double);
}

static void flat_args(hpx_plan* p, FlatArgs& A, int iter_tag) {
  A.ia = p->ia; A.cre = p->Cre; A.rre = p->Rre; A.rim = p->Rim; A.p2re = p->P2re; A.p2im = p->P2im;
  A.hre = p->Hre; A.him = p->Him; A.p4re = p->P4re; A.p4im = p->P4im;
  A.Xre = p->Xre; A.Xim = p->Xim; A.info = p->info;
  A.N = p->N; A.M = p->M; A.NP = p->NP; A.TP = p->TP; A.ncol = p->ncolR; A.npad = p->npad;
  A.has_omega = p->has_omega; A.iter_tag = iter_tag;
  A.xin_re = nullptr; A.xin_im = nullptr; A.CP = 0;
}

// foreground blocks of the Schur complement, [nbl][16][16 + TP] planar, with the per-baseline
// noise level cval[b] (low-rank solver, FFT form; needs M <= 16)
int hpx_launch_flat_blocks(hpx_plan* p, const double* cval, double* ore, double* oim, double* xin_re,
                           double* xin_im, int cp, hipStream_t st) {
  FlatArgs A;
  flat_args(p, A, 0);
  A.xin_re = xin_re; A.xin_im = xin_im; A.CP = cp;
  const size_t lds = hpx_flat_lds_bytes(p);
  static hpx_lds_limit limit;
  HPX_TRY(limit.ensure(reinterpret_cast<const void*>(&k_flat_blocks), lds));
  hipLaunchKernelGGL(k_flat_blocks, dim3(p->nbl), dim3(256), lds, st, A, cval, ore, oim);
  HPX_HIP(hipGetLastError());
  return HPX_OK;
}

int hpx_launch_solve_flat(hpx_plan* p, int iter_tag, hipStream_t st) {
  FlatArgs A;
  flat_args(p, A, iter_tag);
  const size_t lds = hpx_flat_lds_bytes(p);
  static hpx_lds_limit limit;
  HPX_TRY(limit.ensure(reinterpret_cast<const void*>(&k_solve_flat), lds));
  hipLaunchKernelGGL(k_solve_flat, dim3(p->nbl), dim3(256), lds, st, A);
  HPX_HIP(hipGetLastError());
  return HPX_OK;
}
