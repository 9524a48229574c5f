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
// k_lr_schur forms S and Rf on the f64 MFMA and writes them in the factor layout; the small dense
// system goes through the batched Cholesky / back substitution of hpx_factor.hip; k_lr_back forms
// z.  X = [z; f] comes out in the layout k_backsolve produces, everything downstream is shared
// with the other solvers.
//
// k_lr_schur streams the operand matrix Xc = [Bd | r1] (N x (npadS + TP), the border laid out once
// per plan, the r1 columns refreshed every iteration by k_lr_r1) through LDS in chunks of 8
// channels with global_load_lds (no VGPR staging, two buffers), and every wave keeps up to 11
// output tiles in accumulators: a baseline's ~90 tiles are spread over the waves of 1-3
// workgroups, so the 3 MB operand is read once per workgroup and iteration instead of once per
// tile block (nine times at the C5 shape, which made the register-blocked version HBM-bound).
//
// FFT form (power-of-two N, M <= 16; hpx_plan.lr_fft).  The columns of Vf are columns of the Fourier
// operator, so every product with Vf is a DFT evaluated at (or scattered from) the flagged channels
// and the border never has to be laid out:
//     (Vf^H Dinv Vf)[i][j] = dhat[(x_j - x_i + N/2) mod N] / N,   dhat = Fop^T dinv      (1 transform)
//     (G^H Dinv Vf)[m][j]  = (Fop^T (dinv . conj G_m))[x_j] / sqrt N                     (M transforms)
//     (Vf^H Dinv r1)[j][t] = (conj(Fop)^T (dinv . r1_t))[x_j] / sqrt N                   (T transforms)
//     (Vf y)[k][t]         = (Fop^T w_t)[k] / sqrt N,  w_t[x_j] = y_j[t], zero elsewhere  (T transforms)
// i.e. 1 + M + 2T length-N transforms, an f x f gather, the M x M / M x T foreground blocks the
// flat-noise solver already forms (hpx_flat.hip), and the same small Cholesky: O(N log N (M + T))
// instead of O(N (M + f)(M + f + T)).
#include "hpx_internal.h"

namespace {

struct LrArgs {
  const double *ia, *cre, *rre, *rim, *p2re, *p2im, *hre, *him, *p4re, *p4im, *fopre, *fopim;
  const int32_t *flist, *fcount;     // [nbl][fmax] flagged channels, [nbl] their number
  const double* cval;                // [nbl] inverse noise variance of the unflagged channels
  const double *bre, *bim;           // [nbl][NP][npadX] Xc = [G | sqrt(c) Vf | 0 | r1], planar, npadX = npadS + TP
  const double *tre, *tim;           // [nbl][npadS][NP] its transpose
  double* Ls;                        // [nbl] small system in the factor layout (npadS, ldS)
  const double *Yre, *Yim;           // [nbl][npadS][TP] its solution
  double *Xre, *Xim;
  int N, M, NP, TP, ncol, npad, has_omega, fmax, npadS, ldS, npadX, nbl;
  // FFT form: transform input / output [nbl][NP][XW] planar (columns [0, CP): dinv, dinv conj(G);
  // columns [CP, CP + TP): dinv r1, later the scattered y), the foreground blocks [nbl][16][16 + TP],
  // channel -> index in the flagged list (or -1)
  double *ire, *iim;
  const double *ore, *oim, *sre, *sim;
  const int32_t* finv;
  int CP, XW;
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

// The border does not change along the chain: it is laid out once, planar, [NP][npadX] per baseline
// (columns npadS.. are the r1 block k_lr_r1 rewrites every iteration), plus a transposed copy.
__global__ void k_lr_border(const LrArgs A, double* __restrict__ bre, double* __restrict__ bim,
                            double* __restrict__ tre, double* __restrict__ tim) {
  const int b = blockIdx.y, N = A.N, NP = A.NP, npadS = A.npadS, npadX = A.npadX;
  const double* rre = A.rre + (long)b * NP * A.ncol;
  const double* rim = A.rim + (long)b * NP * A.ncol;
  const int* fl = A.flist + (long)b * A.fmax;
  const int fcnt = A.fcount[b];
  const double sc = sqrt(A.cval[b]) * A.isn;
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < (long)NP * npadS; e += (long)gridDim.x * blockDim.x) {
    const int k = (int)(e / npadS), col = (int)(e % npadS);
    double vr = 0.0, vi = 0.0;
    if (k < N) border(A, rre, rim, fl, fcnt, sc, k, col, vr, vi);
    bre[(long)b * NP * npadX + (long)k * npadX + col] = vr;
    bim[(long)b * NP * npadX + (long)k * npadX + col] = vi;
    tre[(long)b * NP * npadS + (long)col * NP + k] = vr;      // transposed copy [npadS][NP] for k_lr_back
    tim[(long)b * NP * npadS + (long)col * NP + k] = vi;
  }
}

// r1 = Q + diag(1/a) P2 into columns npadS.. of Xc (rows >= N stay zero)
__global__ void k_lr_r1(const LrArgs A, double* __restrict__ bre, double* __restrict__ bim) {
  const int b = blockIdx.y, N = A.N, NP = A.NP, TP = A.TP, npadX = A.npadX;
  const double* rre = A.rre + (long)b * NP * A.ncol;
  const double* rim = A.rim + (long)b * NP * A.ncol;
  const double* ia = A.ia + (long)b * N;
  double* ore = bre + (long)b * NP * npadX + A.npadS;
  double* oim = bim + (long)b * NP * npadX + A.npadS;
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < (long)NP * TP; e += (long)gridDim.x * blockDim.x) {
    const int k = (int)(e / TP), t = (int)(e % TP);
    double vr = 0.0, vi = 0.0;
    if (k < N) {
      vr = rre[(long)k * A.ncol + t];
      vi = rim[(long)k * A.ncol + t];
      if (A.has_omega) {
        const double a = ia[k];
        vr = fma(a, A.p2re[(long)k * TP + t], vr);
        vi = fma(a, A.p2im[(long)k * TP + t], vi);
      }
    }
    ore[(long)k * npadX + t] = vr;
    oim[(long)k * npadX + t] = vi;
  }
}

constexpr int LR_KC = 8;               // channels per staged chunk (two MFMA k-steps)

static __device__ __forceinline__ void lr_glds16(const double* src, double* dst) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                   (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
}

// S = E - Bd^H Dinv Bd (lower tiles) and Rf = [P4; 0] - Bd^H Dinv r1, written in the factor layout.
// Output tiles in row-major order (row ri: S tiles 0..ri, then the TT tiles of Rf) are dealt out
// evenly to the 4 nwg waves of a baseline's workgroups, at most TPW each.  Per chunk: wait for
// the own global_load_lds of this chunk, barrier (everybody's pieces have landed, everybody is
// done with the other buffer), issue the next chunk into the other buffer, multiply.  A tile is
// acc[m][c] += sum_k conj(Xc[k][m]) dinv_k Xc[k][c]: both operands come from the staged rows.
template <int TPW>
__global__ __launch_bounds__(256, 2) void k_lr_schur(const LrArgs A, const int nwg) {
  extern __shared__ double lds[];
  // the workgroups of a baseline share an XCD (and its L2)
  const int id = blockIdx.x, xcd = id & 7, slot = id >> 3;
  const int b = (slot / nwg) * 8 + xcd, wg = slot % nwg;
  if (b >= A.nbl) return;
  const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  const int li = lane & 15, g = lane >> 4;
  const int N = A.N, M = A.M, NP = A.NP, TP = A.TP, TT = TP >> 4;
  const int npadS = A.npadS, npadX = A.npadX, mt = npadS >> 4;
  double* dinv = lds;
  double* stage = lds + NP;
  const int bufd = 2 * LR_KC * npadX;                 // doubles per buffer: [re | im][8][npadX]
  {
    const double* ia = A.ia + (long)b * N;
    const double c0 = A.cval[b];
    for (int k = tid; k < NP; k += 256) {
      const double v = (k < N) ? ia[k] : 0.0;
      dinv[k] = (k < N) ? 1.0 / fma(v, v, c0) : 0.0;
    }
  }
  __syncthreads();
  // this wave's tiles
  const int ntile = mt * (mt + 1) / 2 + mt * TT, nwv = 4 * nwg, q = wg * 4 + wave;
  const int base = ntile / nwv, extra = ntile % nwv;
  const int cnt = base + (q < extra ? 1 : 0), j0 = q * base + min(q, extra);
  int tri[TPW], tcx[TPW];
  {
    int ri = 0, start = 0;
    while (ri < mt - 1 && j0 >= start + ri + 1 + TT) { start += ri + 1 + TT; ++ri; }
    int pos = j0 - start;
#pragma unroll
    for (int j = 0; j < TPW; ++j) {
      const bool live = j < cnt;
      tri[j] = live ? ri : 0;
      tcx[j] = live ? (pos <= ri ? pos : mt + (pos - ri - 1)) : 0;
      if (live && ++pos == ri + 1 + TT) { pos = 0; ++ri; }
    }
  }
  d4 ar[TPW], ai[TPW];
#pragma unroll
  for (int j = 0; j < TPW; ++j) {
    ar[j] = (d4){0., 0., 0., 0.};
    ai[j] = (d4){0., 0., 0., 0.};
  }
  const double* bre = A.bre + (long)b * NP * npadX;
  const double* bim = A.bim + (long)b * NP * npadX;
  const int np = npadX >> 4;                          // 1 KB pieces per plane and chunk
#define HPX_LR_STAGE(chunk, bufi)                                                       \
  for (int p_ = wave; p_ < 2 * np; p_ += 4) {                                            \
    const int pl_ = p_ >= np ? 1 : 0, ix_ = p_ - pl_ * np;                               \
    lr_glds16((pl_ ? bim : bre) + (long)(chunk) * LR_KC * npadX + ix_ * 128 + 2 * lane,  \
              stage + (bufi) * bufd + pl_ * LR_KC * npadX + ix_ * 128);                  \
  }
  const int nch = NP / LR_KC;
  const int lanepart = g * npadX + li;
  typedef __attribute__((address_space(3))) double lds_f64;
  HPX_LR_STAGE(0, 0)
  for (int ch = 0; ch < nch; ++ch) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (ch + 1 < nch) { HPX_LR_STAGE(ch + 1, (ch + 1) & 1) }
    const lds_f64* Bc = (const lds_f64*)(stage + (ch & 1) * bufd + lanepart);
    const lds_f64* dv = (const lds_f64*)(dinv + ch * LR_KC + g);
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      const double dk = dv[4 * s];
      const lds_f64* Bs = Bc + s * 4 * npadX;
#pragma unroll
      for (int j = 0; j < TPW; ++j) {
        const double a_r = Bs[16 * tri[j]], a_i = Bs[LR_KC * npadX + 16 * tri[j]];
        const double b_r = Bs[16 * tcx[j]], b_i = Bs[LR_KC * npadX + 16 * tcx[j]];
        const double x_r = dk * a_r, x_i = dk * a_i;
        ar[j] = mfma64(x_r, b_r, ar[j]);
        ar[j] = mfma64(x_i, b_i, ar[j]);
        ai[j] = mfma64(x_r, b_i, ai[j]);
        ai[j] = mfma64(-x_i, b_r, ai[j]);
      }
    }
  }
#undef HPX_LR_STAGE
  // lane (li, g), register v holds row m = 16 ri + g + 4v of the output, column 16 cx + li
  double* L = A.Ls + (long)b * npadS * A.ldS * 2;
#pragma unroll
  for (int j = 0; j < TPW; ++j) {
    const int ri = tri[j], cx = tcx[j];
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      if (j >= cnt) continue;                      // (no break: the loop must unroll, acc stays in registers)
      const int m = 16 * ri + HPX_ACC_ROW(g, v);
      if (cx < mt) {                               // S = E - acc
        const int mc = 16 * cx + li;
        double e_r = 0.0, e_i = 0.0;
        if (m < M && mc < M) {
          e_r = A.hre[(long)b * M * M + m * M + mc];
          e_i = A.him[(long)b * M * M + m * M + mc];
        } else if (m == mc) {
          e_r = 1.0;
        }
        const long o = HPX_LIDX(m, mc, npadS);
        L[o] = e_r - ar[j][v];
        L[o + 16] = e_i - ai[j][v];
      } else {                                     // row npadS + t of the factor buffer = conj(Rf[m][t])
        const int tc = ((cx - mt) << 4) + li;
        double e_r = 0.0, e_i = 0.0;
        if (m < M) {
          e_r = A.p4re[(long)b * M * TP + m * TP + tc];
          e_i = A.p4im[(long)b * M * TP + m * TP + tc];
        }
        const long o = HPX_LIDX(npadS + tc, m, npadS);
        L[o] = e_r - ar[j][v];
        L[o + 16] = -(e_i - ai[j][v]);
      }
    }
  }
}

// z = Dinv (r1 - Bd [f; y]) for 16 channels x TTG t-tiles at a time per wave: the border tile is
// fetched once for all the t-tiles of a group, operands of the next k-step are in flight while
// the current one is multiplied; r1 comes from the columns k_lr_r1 wrote.
template <int TTG>
__global__ __launch_bounds__(256) void k_lr_back(const LrArgs A) {
  extern __shared__ double lds[];
  const int b = blockIdx.x, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int li = lane & 15, g = lane >> 4;
  const int N = A.N, M = A.M, NP = A.NP, TP = A.TP, TT = TP >> 4, npadX = A.npadX;
  double* dinv = lds;
  const double* ia = A.ia + (long)b * N;
  const double c0 = A.cval[b];
  const int fcnt = A.fcount[b];
  const double* tre = A.tre + (long)b * NP * A.npadS;
  const double* tim = A.tim + (long)b * NP * A.npadS;
  const double* xre = A.bre + (long)b * NP * npadX + A.npadS;     // r1[k][t] at xre[k * npadX + t]
  const double* xim = A.bim + (long)b * NP * npadX + A.npadS;
  for (int k = tid; k < NP; k += 256) {
    const double v = (k < N) ? ia[k] : 0.0;
    dinv[k] = (k < N) ? 1.0 / fma(v, v, c0) : 0.0;
  }
  __syncthreads();
  const double* yre = A.Yre + (long)b * A.npadS * TP;
  const double* yim = A.Yim + (long)b * A.npadS * TP;
  double* Xre = A.Xre + (long)b * A.npad * TP;
  double* Xim = A.Xim + (long)b * A.npad * TP;
  const int nms = (M + fcnt + 3) >> 2;                // k-steps over the live border columns (>= 1)
  for (int kt = wave; kt < (NP >> 4); kt += 4) {
    const int k0 = kt << 4;
    for (int tg = 0; tg < TT; tg += TTG) {
      d4 zr[TTG], zi[TTG];
#pragma unroll
      for (int q = 0; q < TTG; ++q) {
        const int t = (min(tg + q, TT - 1) << 4) + li;
#pragma unroll
        for (int v = 0; v < 4; ++v) {                 // r1[k][t], k = k0 + g + 4v (rows >= N are zero)
          const long o = (long)(k0 + HPX_ACC_ROW(g, v)) * npadX + t;
          zr[q][v] = xre[o];
          zi[q][v] = xim[o];
        }
      }
      double a0r, a0i, a1r, a1i, f0r[TTG], f0i[TTG], f1r[TTG], f1i[TTG];
#define HPX_LB_LOAD(ar_, ai_, fr_, fi_, ms_)                                                   \
  {                                                                                            \
    const int m_ = 4 * (ms_) + g;                                                              \
    ar_ = tre[(long)m_ * NP + k0 + li];              /* A[k = k0 + li][m] = Bd[k][m], unit stride in k */ \
    ai_ = tim[(long)m_ * NP + k0 + li];                                                        \
    _Pragma("unroll") for (int q = 0; q < TTG; ++q) {                                          \
      const int t_ = (min(tg + q, TT - 1) << 4) + li;                                          \
      fr_[q] = yre[(long)m_ * TP + t_];              /* B[m][t] = Y[m][t] */                    \
      fi_[q] = yim[(long)m_ * TP + t_];                                                        \
    }                                                                                          \
  }
#define HPX_LB_MMA(ar_, ai_, fr_, fi_)                                                         \
  {                                                                                            \
    const double nr_ = -ar_, ni_ = -ai_;                                                       \
    _Pragma("unroll") for (int q = 0; q < TTG; ++q) {                                          \
      zr[q] = mfma64(nr_, fr_[q], zr[q]);                                                      \
      zr[q] = mfma64(ai_, fi_[q], zr[q]);                                                      \
      zi[q] = mfma64(nr_, fi_[q], zi[q]);                                                      \
      zi[q] = mfma64(ni_, fr_[q], zi[q]);                                                      \
    }                                                                                          \
  }
      HPX_LB_LOAD(a0r, a0i, f0r, f0i, 0)
      for (int ms = 0; ms < nms; ms += 2) {
        HPX_LB_LOAD(a1r, a1i, f1r, f1i, min(ms + 1, nms - 1))
        __builtin_amdgcn_sched_barrier(0);
        HPX_LB_MMA(a0r, a0i, f0r, f0i)
        __builtin_amdgcn_sched_barrier(0);
        HPX_LB_LOAD(a0r, a0i, f0r, f0i, min(ms + 2, nms - 1))
        __builtin_amdgcn_sched_barrier(0);
        if (ms + 1 < nms) HPX_LB_MMA(a1r, a1i, f1r, f1i)
        __builtin_amdgcn_sched_barrier(0);
      }
#undef HPX_LB_LOAD
#undef HPX_LB_MMA
#pragma unroll
      for (int q = 0; q < TTG; ++q) {
        if (tg + q >= TT) continue;
        const int t = ((tg + q) << 4) + li;
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
    Xre[(long)(N + m) * TP + t] = (m < M) ? yre[(long)m * TP + t] : 0.0;
    Xim[(long)(N + m) * TP + t] = (m < M) ? yim[(long)m * TP + t] : 0.0;
  }
}

// ---------------------------------------------------------------------------------------------
// FFT form
// S and Rf in the factor layout from the transforms (ore/oim) and the foreground blocks (sre/sim).
// Elements are visited in storage order (16 consecutive rows of a panel, then the column): a
// wave's stores are 128-byte runs.  dhat (column 0 of the transform output, stride XW in memory)
// is staged in LDS once: every entry of the f x f block is a lookup in it.
__global__ __launch_bounds__(256) void k_lrf_gather(const LrArgs A) {
  extern __shared__ double lds[];
  const int b = blockIdx.x, tid = threadIdx.x;
  const int N = A.N, M = A.M, NP = A.NP, TP = A.TP, npadS = A.npadS, XW = A.XW, CP = A.CP, SW = 16 + TP;
  const int* fl = A.flist + (long)b * A.fmax;
  const int fcnt = A.fcount[b];
  const double c0 = A.cval[b];
  const double sc = sqrt(c0) * A.isn, cn = c0 * A.isn * A.isn;      // sqrt(c / N), c / N
  const double* ore = A.ore + (long)b * NP * XW;
  const double* oim = A.oim + (long)b * NP * XW;
  const double* sre = A.sre + (long)b * 16 * SW;
  const double* sim = A.sim + (long)b * 16 * SW;
  double* L = A.Ls + (long)b * npadS * A.ldS * 2;
  double* dhr = lds;                                // [N]
  double* dhi = dhr + N;
  int* xf = reinterpret_cast<int*>(dhi + N);        // [npadS] channel of border column M + i (or -1)
  for (int x = tid; x < N; x += 256) {
    dhr[x] = ore[(long)x * XW];
    dhi[x] = oim[(long)x * XW];
  }
  for (int m = tid; m < npadS; m += 256) xf[m] = (m >= M && m - M < fcnt) ? fl[m - M] : -1;
  __syncthreads();
  const int h = N / 2;
  for (int e = tid; e < npadS * npadS; e += 256) {
    const int rr = e & 15, q = e >> 4, c = q % npadS, r = ((q / npadS) << 4) + rr;
    if (c > r) continue;
    double vr = (r == c) ? 1.0 : 0.0, vi = 0.0;              // identity padding
    if (r < M) {                                              // foreground block (c <= r < M)
      vr = sre[r * SW + c];
      vi = sim[r * SW + c];
    } else if (xf[r] >= 0) {
      const int xi = xf[r];
      if (c < M) {                                            // conj of -(G^H Dinv sqrt(c) Vf)[c][i]
        vr = -sc * ore[(long)xi * XW + 1 + c];
        vi = sc * oim[(long)xi * XW + 1 + c];
      } else {                                                // delta - c (Vf^H Dinv Vf)[i][j]
        int x = xf[c] - xi + h;
        x += (x < 0) ? N : 0;
        x -= (x >= N) ? N : 0;
        vr -= cn * dhr[x];
        vi = -cn * dhi[x];
      }
    }
    const long o = HPX_LIDX(r, c, npadS);
    L[o] = vr;
    L[o + 16] = vi;
  }
  for (int e = tid; e < npadS * TP; e += 256) {               // row npadS + t = conj(Rf[m][t])
    const int tt = e & 15, q = e >> 4, m = q % npadS, t = ((q / npadS) << 4) + tt;
    double vr = 0.0, vi = 0.0;
    if (m < M) {
      vr = sre[m * SW + 16 + t];
      vi = sim[m * SW + 16 + t];
    } else if (xf[m] >= 0) {                                  // -(sqrt(c) Vf^H Dinv r1)[i][t]
      vr = -sc * ore[(long)xf[m] * XW + CP + t];
      vi = -sc * oim[(long)xf[m] * XW + CP + t];
    }
    const long o = HPX_LIDX(npadS + t, m, npadS);
    L[o] = vr;
    L[o + 16] = -vi;
  }
}

// w[x][t] = y_j[t] at the flagged channels x = x_j, zero elsewhere (input of the last transform)
__global__ void k_lrf_fill(const LrArgs A) {
  const int b = blockIdx.y, N = A.N, NP = A.NP, TP = A.TP, M = A.M, CP = A.CP, XW = A.XW;
  const int32_t* finv = A.finv + (long)b * N;
  const double* yre = A.Yre + (long)b * A.npadS * TP;
  const double* yim = A.Yim + (long)b * A.npadS * TP;
  double* ire = A.ire + (long)b * NP * XW + CP;
  double* iim = A.iim + (long)b * NP * XW + CP;
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < (long)NP * TP; e += (long)gridDim.x * blockDim.x) {
    const int k = (int)(e / TP), t = (int)(e % TP);
    const int j = (k < N) ? finv[k] : -1;
    ire[(long)k * XW + t] = (j >= 0) ? yre[(long)(M + j) * TP + t] : 0.0;
    iim[(long)k * XW + t] = (j >= 0) ? yim[(long)(M + j) * TP + t] : 0.0;
  }
}

// z = Dinv (r1 - G f - sqrt(c/N) What), X = [z; f; 0]
__global__ __launch_bounds__(256) void k_lrf_back(const LrArgs A) {
  extern __shared__ double lds[];
  const int b = blockIdx.x, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int li = lane & 15, g = lane >> 4;
  const int N = A.N, M = A.M, NP = A.NP, TP = A.TP, ncol = A.ncol, TT = TP >> 4, XW = A.XW, CP = A.CP;
  double* dinv = lds;
  double* iav = dinv + NP;
  double* fre = iav + NP;                          // f[m][t], 16 x TP
  double* fim = fre + 16 * TP;
  const double* ia = A.ia + (long)b * N;
  const double* rre = A.rre + (long)b * NP * ncol;
  const double* rim = A.rim + (long)b * NP * ncol;
  const double c0 = A.cval[b];
  const double sc = sqrt(c0) * A.isn;
  const double* wre = A.ore + (long)b * NP * XW + CP;
  const double* wim = A.oim + (long)b * NP * XW + CP;
  const double* yre = A.Yre + (long)b * A.npadS * TP;
  const double* yim = A.Yim + (long)b * A.npadS * TP;
  for (int k = tid; k < NP; k += 256) {
    const double v = (k < N) ? ia[k] : 0.0;
    iav[k] = v;
    dinv[k] = (k < N) ? 1.0 / fma(v, v, c0) : 0.0;
  }
  for (int e = tid; e < 16 * TP; e += 256) {
    const int m = e / TP;
    fre[e] = (m < M) ? yre[e] : 0.0;
    fim[e] = (m < M) ? yim[e] : 0.0;
  }
  __syncthreads();
  double* Xre = A.Xre + (long)b * A.npad * TP;
  double* Xim = A.Xim + (long)b * A.npad * TP;
  for (int kt = wave; kt < (NP >> 4); kt += 4) {
    const int k0 = kt << 4;
    double ga_r[4], ga_i[4];                       // A[k = k0 + li][m = 4 ks + g] = -G[k][m]
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const long o = (long)min(k0 + li, N - 1) * ncol + TP + 4 * ks + g;
      ga_r[ks] = -rre[o];                          // columns M .. 15 of G are zero padding (k_prep)
      ga_i[ks] = -rim[o];
    }
    // two t-tiles at a time, every load issued before the first use and none under a runtime
    // condition (P2 is zero without omega)
    for (int tt = 0; tt < TT; tt += 2) {
      d4 zr[2], zi[2];
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const int t = (min(tt + q, TT - 1) << 4) + li;
#pragma unroll
        for (int v = 0; v < 4; ++v) {              // r1[k][t] - sqrt(c/N) What[k][t], k = k0 + g + 4v
          const int kk = k0 + HPX_ACC_ROW(g, v), k = min(kk, N - 1);
          const double r_r = fma(iav[k], A.p2re[(long)k * TP + t], rre[(long)k * ncol + t]);
          const double r_i = fma(iav[k], A.p2im[(long)k * TP + t], rim[(long)k * ncol + t]);
          zr[q][v] = fma(-sc, wre[(long)kk * XW + t], r_r);
          zi[q][v] = fma(-sc, wim[(long)kk * XW + t], r_i);
        }
      }
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const int t = (min(tt + q, TT - 1) << 4) + li;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {           // B[m = 4 ks + g][t] = f[m][t]
          const int m = 4 * ks + g;
          const double f_r = fre[m * TP + t], f_i = fim[m * TP + t];
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
    Xre[(long)(N + m) * TP + t] = (m < M) ? fre[m * TP + t] : 0.0;
    Xim[(long)(N + m) * TP + t] = (m < M) ? fim[m * TP + t] : 0.0;
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
  A.npadX = p->lr_npad + p->TP; A.nbl = p->nbl;
  A.ire = p->lr_Ire; A.iim = p->lr_Iim; A.ore = p->lr_Ore; A.oim = p->lr_Oim; A.sre = p->lr_Sre; A.sim = p->lr_Sim;
  A.finv = p->lr_finv; A.CP = p->lr_cp; A.XW = p->lr_cp + p->TP;
  A.isn = 1.0 / sqrt((double)p->N);
}

// the iteration-invariant border of every baseline (after hpx_plan_set_solver filled the lists)
int hpx_lowrank_prepare(hpx_plan* p, hipStream_t st) {
  if (p->lr_fft) return HPX_OK;                       // the FFT form never lays the border out
  LrArgs A;
  lr_args(p, A);
  hipLaunchKernelGGL(k_lr_border, dim3(64, p->nbl), dim3(256), 0, st, A, p->lr_Bre, p->lr_Bim, p->lr_Tre, p->lr_Tim);
  HPX_HIP(hipGetLastError());
  HPX_HIP(hipStreamSynchronize(st));
  return HPX_OK;
}

template <int TPW>
static int launch_schur(hpx_plan* p, const LrArgs& A, int nwg, size_t lds, hipStream_t st) {
  static hpx_lds_limit limit;      // per instantiation
  HPX_TRY(limit.ensure(reinterpret_cast<const void*>(&k_lr_schur<TPW>), lds));
  const int grid = ((p->nbl + 7) / 8) * 8 * nwg;
  hipLaunchKernelGGL((k_lr_schur<TPW>), dim3(grid), dim3(256), lds, st, A, nwg);
  HPX_HIP(hipGetLastError());
  return HPX_OK;
}

// LDS bytes of k_lr_schur: 1/(c + N/ps) of every channel + two staging buffers
size_t hpx_lowrank_lds_bytes(const hpx_plan* p) {
  return ((size_t)p->NP + (size_t)2 * 2 * LR_KC * (p->lr_npad + p->TP)) * sizeof(double);
}

static int solve_lowrank_fft(hpx_plan* p, const LrArgs& A, int iter_tag, hipStream_t st) {
  const int NP = p->NP, TP = p->TP, CP = p->lr_cp, XW = CP + TP;
  const long bs = (long)NP * XW;
  // foreground blocks; the same pass writes the transform input [dinv | dinv conj(G) | dinv r1]
  HPX_TRY(hpx_launch_flat_blocks(p, p->lr_c, p->lr_Sre, p->lr_Sim, p->lr_Ire, p->lr_Iim, CP, st));
  HPX_TRY(hpx_launch_dft(p->nbl, NP, CP, p->Fopre, p->Fopim, 0, p->lr_Ire, p->lr_Iim, bs, XW, nullptr, 0,
                         p->lr_Ore, p->lr_Oim, bs, XW, 1.0, st, 1));
  HPX_TRY(hpx_launch_dft(p->nbl, NP, TP, p->Fopre, p->Fopim, 1, p->lr_Ire + CP, p->lr_Iim + CP, bs, XW, nullptr, 0,
                         p->lr_Ore + CP, p->lr_Oim + CP, bs, XW, 1.0, st, 1));
  const size_t lds_g = (size_t)2 * p->N * sizeof(double) + (size_t)p->lr_npad * sizeof(int);
  static hpx_lds_limit limit_g;
  HPX_TRY(limit_g.ensure(reinterpret_cast<const void*>(&k_lrf_gather), lds_g));
  hipLaunchKernelGGL(k_lrf_gather, dim3(p->nbl), dim3(256), lds_g, st, A);
  HPX_HIP(hipGetLastError());
  HPX_TRY(hpx_launch_factor(p->nbl, p->lr_npad, p->lr_npad + TP, p->lr_L, p->lr_Wre, p->lr_Wim, p->lr_Vt, p->info,
                            iter_tag, nullptr, st, 0));
  HPX_TRY(hpx_launch_backsolve(p->nbl, p->lr_npad, TP, p->lr_npad + TP, p->lr_L, p->lr_Wre, p->lr_Wim,
                               p->lr_Yre, p->lr_Yim, st));
  hipLaunchKernelGGL(k_lrf_fill, dim3(32, p->nbl), dim3(256), 0, st, A);
  HPX_HIP(hipGetLastError());
  HPX_TRY(hpx_launch_dft(p->nbl, NP, TP, p->Fopre, p->Fopim, 0, p->lr_Ire + CP, p->lr_Iim + CP, bs, XW, nullptr, 0,
                         p->lr_Ore + CP, p->lr_Oim + CP, bs, XW, 1.0, st, 1));
  const size_t lds = ((size_t)2 * NP + (size_t)2 * 16 * TP) * sizeof(double);
  static hpx_lds_limit limit;
  HPX_TRY(limit.ensure(reinterpret_cast<const void*>(&k_lrf_back), lds));
  hipLaunchKernelGGL(k_lrf_back, dim3(p->nbl), dim3(256), lds, st, A);
  HPX_HIP(hipGetLastError());
  return HPX_OK;
}

int hpx_launch_solve_lowrank(hpx_plan* p, int iter_tag, hipStream_t st) {
  LrArgs A;
  lr_args(p, A);
  if (p->lr_fft) return solve_lowrank_fft(p, A, iter_tag, st);
  hipLaunchKernelGGL(k_lr_r1, dim3(32, p->nbl), dim3(256), 0, st, A, p->lr_Bre, p->lr_Bim);
  HPX_HIP(hipGetLastError());
  // output tiles per baseline over the waves of nwg workgroups, at most 11 per wave
  const int mt = p->lr_npad >> 4, TT = p->TP >> 4, ntile = mt * (mt + 1) / 2 + mt * TT;
  const int nwg = (ntile + 43) / 44, per = (ntile + 4 * nwg - 1) / (4 * nwg);
  const size_t lds_s = hpx_lowrank_lds_bytes(p);
  if (per <= 3) HPX_TRY(launch_schur<3>(p, A, nwg, lds_s, st));
  else if (per <= 6) HPX_TRY(launch_schur<6>(p, A, nwg, lds_s, st));
  else if (per <= 8) HPX_TRY(launch_schur<8>(p, A, nwg, lds_s, st));
  else HPX_TRY(launch_schur<11>(p, A, nwg, lds_s, st));
  HPX_TRY(hpx_launch_factor(p->nbl, p->lr_npad, p->lr_npad + p->TP, p->lr_L, p->lr_Wre, p->lr_Wim, p->lr_Vt, p->info,
                            iter_tag, nullptr, st, 0));
  HPX_TRY(hpx_launch_backsolve(p->nbl, p->lr_npad, p->TP, p->lr_npad + p->TP, p->lr_L, p->lr_Wre, p->lr_Wim,
                               p->lr_Yre, p->lr_Yim, st));
  const size_t lds = (size_t)p->NP * sizeof(double);
  if (TT >= 2) hipLaunchKernelGGL(k_lr_back<2>, dim3(p->nbl), dim3(256), lds, st, A);
  else hipLaunchKernelGGL(k_lr_back<1>, dim3(p->nbl), dim3(256), lds, st, A);
  HPX_HIP(hipGetLastError());
  return HPX_OK;
}
