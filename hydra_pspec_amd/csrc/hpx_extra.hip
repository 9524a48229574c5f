// DPSS weighted fit (closed form), OQE helpers, and the MFMA lane-map probe.
#include <math.h>
#include "hpx_internal.h"

namespace {

// ---- MFMA probe -------------------------------------------------------------
__global__ void k_probe(const double* __restrict__ a, const double* __restrict__ b,
                        double* __restrict__ d) {
  const int l = threadIdx.x;
  // lane l supplies A[l&15][l>>4] and B[l>>4][l&15]
  const double av = a[(l & 15) * 4 + (l >> 4)];
  const double bv = b[(l >> 4) * 16 + (l & 15)];
  d4 c = {0., 0., 0., 0.};
  c = mfma64(av, bv, c);
  for (int v = 0; v < 4; ++v) d[l * 4 + v] = c[v];
}

// back-to-back v_mfma_f64_16x16x4_f64 issue rate: 4 independent accumulators per wave
__global__ __launch_bounds__(256, 2) void k_mfma_peak(double* __restrict__ sink, const int iters) {
  d4 c0 = {0., 0., 0., 0.}, c1 = c0, c2 = c0, c3 = c0;
  const double a = 1.0 + 1e-9 * threadIdx.x, b = 1.0 - 1e-9 * threadIdx.x;
  for (int i = 0; i < iters; ++i) {
    c0 = mfma64(a, b, c0);
    c1 = mfma64(b, a, c1);
    c2 = mfma64(a, a, c2);
    c3 = mfma64(b, b, c3);
  }
  const d4 s = c0 + c1 + c2 + c3;
  if (s[0] == -1.0) sink[blockIdx.x * 256 + threadIdx.x] = s[0] + s[1] + s[2] + s[3];
}

// ---- DPSS ---------------------------------------------------------------------
// Closed form of hydra_pspec/dpss.py:7-94 for GROUPS of spectra that share their weights (the
// time samples of one baseline: flags are time independent on the Gibbs path):
//   c_b = A_g^-1 P_g d_b,   A_g = Mw^H Ah Mw (nm x nm),   P_g = Mw^H Ah diag(tw_g)  (nm x N),
//   Mw = diag(tw_g) modes^T,  Ah = Hermitian part of inv(cov).
// Per group (k_dpss_in -> dense MFMA product Ah Mw -> k_dpss_group): the weighted normal matrix, its
// inverse and the projector P_g.  Per spectrum (k_dpss_apply): the tall-skinny projection
// P_g d_b and the multiplication by A_g^-1, both on the f64 MFMA, with the visibility cube read once,
// coalesced, through LDS tiles.

// in[g][j][col] = tw_g[j] * modes[col][j]  (col < nm), zero padded to NP x ncol
__global__ void k_dpss_in(const double* __restrict__ tw, const double* __restrict__ modes,
                          double* __restrict__ ire, double* __restrict__ iim, const int N, const int nm,
                          const int NP, const int ncol) {
  const int b = blockIdx.y;
  const long tot = (long)NP * ncol;
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < tot;
       e += (long)gridDim.x * blockDim.x) {
    const int j = (int)(e / ncol), col = (int)(e % ncol);
    double vr = 0.0;
    if (j < N && col < nm) vr = tw[(long)b * N + j] * modes[(long)col * N + j];
    ire[(long)b * tot + e] = vr;
    iim[(long)b * tot + e] = 0.0;
  }
}

// Hermitian part of icov, planar, stored so that W[k*NP + x] = conj(Ah[x][k]) = Ah[k][x]
__global__ void k_herm_planar(const double* __restrict__ icov, double* __restrict__ re,
                              double* __restrict__ im, const int N, const int NP) {
  const long tot = (long)NP * NP;
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < tot;
       e += (long)gridDim.x * blockDim.x) {
    const int k = (int)(e / NP), x = (int)(e % NP);
    double vr = 0.0, vi = 0.0;
    if (k < N && x < N) {
      const long a = ((long)k * N + x) * 2, t = ((long)x * N + k) * 2;
      vr = 0.5 * (icov[a] + icov[t]);
      vi = 0.5 * (icov[a + 1] - icov[t + 1]);
    }
    re[e] = vr;
    im[e] = vi;
  }
}

#define HPX_DPSS_MAXM 32
// One workgroup per group: A = Mw^T O (O = Ah Mw from the dense product, [NP][ncol] planar), its
// inverse by Gauss-Jordan in LDS (Hermitian positive definite: no pivoting), stored TRANSPOSED
// planar AiT[k][k'] = Ainv[k'][k]; projector PT[j][k] = conj(O[j][k]) tw[j] (zero for j >= N, k >= nm).
// info[g] = 1 if a pivot is not positive.
__global__ __launch_bounds__(256) void k_dpss_group(const double* __restrict__ tw,
                                                    const double* __restrict__ modes,
                                                    const double* __restrict__ ore,
                                                    const double* __restrict__ oim,
                                                    double* __restrict__ ptre, double* __restrict__ ptim,
                                                    double* __restrict__ aitre, double* __restrict__ aitim,
                                                    int32_t* __restrict__ info, const int N, const int nm,
                                                    const int NP, const int ncol) {
  __shared__ double Are[HPX_DPSS_MAXM][2 * HPX_DPSS_MAXM + 1], Aim[HPX_DPSS_MAXM][2 * HPX_DPSS_MAXM + 1];
  __shared__ int bad;
  const int gI = blockIdx.x, tid = threadIdx.x;
  const double* pr = ore + (long)gI * NP * ncol;
  const double* pi = oim + (long)gI * NP * ncol;
  const double* twg = tw + (long)gI * N;
  if (tid == 0) bad = 0;
  // projector (coalesced along k) -- rows j >= N and columns k >= nm are zero in O already
  for (int e = tid; e < NP * ncol; e += 256) {
    const int j = e / ncol;
    const double w = (j < N) ? twg[j] : 0.0;
    ptre[(long)gI * NP * ncol + e] = pr[e] * w;
    ptim[(long)gI * NP * ncol + e] = -pi[e] * w;
  }
  // A[k][k'] = sum_j tw[j] modes[k][j] O[j][k']   (threads along k': unit-stride reads of O)
  for (int e = tid; e < nm * ncol; e += 256) {
    const int k = e / ncol, k2 = e % ncol;
    double sr = 0.0, si = 0.0;
    if (k2 < nm)
      for (int j = 0; j < N; ++j) {
        const double w = twg[j] * modes[(long)k * N + j];
        sr = fma(w, pr[(long)j * ncol + k2], sr);
        si = fma(w, pi[(long)j * ncol + k2], si);
      }
    if (k2 < nm) {
      Are[k][k2] = sr;
      Aim[k][k2] = (k == k2) ? 0.0 : si;
      Are[k][nm + k2] = (k == k2) ? 1.0 : 0.0;
      Aim[k][nm + k2] = 0.0;
    }
  }
  __syncthreads();
  for (int k = 0; k < nm; ++k) {           // Gauss-Jordan on [A | I]
    const double piv = Are[k][k];
    if (tid == 0 && !(piv > 0.0)) bad = 1;
    const double rp = 1.0 / piv;
    __syncthreads();
    for (int c = tid; c < 2 * nm; c += 256) {   // scale the pivot row (the pivot is real)
      Are[k][c] *= rp;
      Aim[k][c] *= rp;
    }
    __syncthreads();
    for (int e = tid; e < nm * 2 * nm; e += 256) {
      const int r = e / (2 * nm), c = e % (2 * nm);
      if (r == k || c == k) continue;             // column k is handled after the sweep
      const double fr = Are[r][k], fi = Aim[r][k];
      Are[r][c] -= fr * Are[k][c] - fi * Aim[k][c];
      Aim[r][c] -= fr * Aim[k][c] + fi * Are[k][c];
    }
    __syncthreads();
    for (int r = tid; r < nm; r += 256)
      if (r != k) { Are[r][k] = 0.0; Aim[r][k] = 0.0; }
    __syncthreads();
  }
  for (int e = tid; e < ncol * ncol; e += 256) {  // AiT[k][k'] = Ainv[k'][k]
    const int k = e / ncol, k2 = e % ncol;
    double vr = 0.0, vi = 0.0;
    if (k < nm && k2 < nm && !bad) { vr = Are[k2][nm + k]; vi = Aim[k2][nm + k]; }   // singular group: zero amplitudes
    aitre[(long)gI * ncol * ncol + e] = vr;
    aitim[(long)gI * ncol * ncol + e] = vi;
  }
  if (tid == 0 && info) info[gI] = bad;
}

// the visibility cube is read exactly once: non-temporal loads keep it from displacing the projector,
// which every wave of the group re-reads (HPX_DPSS_NT=0: plain loads, for A/B)
#ifndef HPX_DPSS_NT
#define HPX_DPSS_NT 1
#endif
typedef double hpx_v2d_ __attribute__((ext_vector_type(2)));
__device__ __forceinline__ double2 hpx_ld2_stream(const double* p) {
#if HPX_DPSS_NT
  const hpx_v2d_ v = __builtin_nontemporal_load(reinterpret_cast<const hpx_v2d_*>(p));
  return make_double2(v.x, v.y);
#else
  return *reinterpret_cast<const double2*>(p);
#endif
}

// Tall-skinny projection + nm x nm solve for 16 spectra per wave (4 waves per workgroup):
//   rhs[k][b] = sum_j PT[j][k] d_b[j]   (A operand = projector, unit stride along k; B operand = the
//   spectra, fetched as 16 spectra x 16 channels tiles with 256-byte rows and turned through a
//   wave-private LDS tile),   c[k'][b] = sum_k Ainv[k'][k] rhs[k][b]  (the accumulator tile is the B
//   operand of the second product as it stands).   MT = modes tiles of 16.
template <int MT>
__global__ __launch_bounds__(256) void k_dpss_apply(const double* __restrict__ d,
                                                    const double* __restrict__ ptre,
                                                    const double* __restrict__ ptim,
                                                    const double* __restrict__ aitre,
                                                    const double* __restrict__ aitim,
                                                    double* __restrict__ amps, const int per,
                                                    const int N, const int nm, const int NP,
                                                    const int ncol) {
  constexpr int PITCH = 34;                       // doubles per LDS tile row: 16 complex + pad
  __shared__ double tile_all[4][16 * PITCH];
  const int gI = blockIdx.y, wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int li = lane & 15, g = lane >> 4;
  const int b0 = (blockIdx.x * 4 + wave) * 16;    // first spectrum of this wave within the group
  if (b0 >= per) return;                          // (no workgroup barrier below)
  double* tile = tile_all[wave];
  const double* pre = ptre + (long)gI * NP * ncol;
  const double* pim = ptim + (long)gI * NP * ncol;
  const double* dg = d + (long)gI * per * N * 2;
  d4 ar[MT], ai[MT];
#pragma unroll
  for (int m = 0; m < MT; ++m) {
    ar[m] = (d4){0., 0., 0., 0.};
    ai[m] = (d4){0., 0., 0., 0.};
  }
  // tile fetch: lane l, piece i -> spectrum row (l >> 4) + 4 i, channel j0 + (l & 15): 16 lanes read
  // 256 contiguous bytes of one spectrum
  const int nchunk = NP >> 4;
  double2 cur[4], nxt[4];
#define HPX_DP_FETCH(dst, ch_)                                                          \
  _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                       \
    const int row_ = g + 4 * i, j_ = ((ch_) << 4) + li;                                 \
    const int bb_ = min(b0 + row_, per - 1);                                            \
    dst[i] = (j_ < N) ? hpx_ld2_stream(dg + ((long)bb_ * N + j_) * 2)                    \
                      : make_double2(0.0, 0.0);                                         \
  }
  HPX_DP_FETCH(cur, 0)
  for (int ch = 0; ch < nchunk; ++ch) {
    const int chn = min(ch + 1, nchunk - 1);
    HPX_DP_FETCH(nxt, chn)                        // next tile in flight while this one is used
#pragma unroll
    for (int i = 0; i < 4; ++i)
      *reinterpret_cast<double2*>(tile + (g + 4 * i) * PITCH + 2 * li) = cur[i];
    __builtin_amdgcn_s_waitcnt(0xc07f);           // lgkmcnt(0): the tile is written (same wave reads it)
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4) {
      const int j = (ch << 4) + 4 * s4 + g;
      const double2 x = *reinterpret_cast<const double2*>(tile + li * PITCH + 2 * (4 * s4 + g));
#pragma unroll
      for (int m = 0; m < MT; ++m) {
        const double pr_ = pre[(long)j * ncol + 16 * m + li], pi_ = pim[(long)j * ncol + 16 * m + li];
        ar[m] = mfma64(pr_, x.x, ar[m]);
        ar[m] = mfma64(-pi_, x.y, ar[m]);
        ai[m] = mfma64(pr_, x.y, ai[m]);
        ai[m] = mfma64(pi_, x.x, ai[m]);
      }
    }
    __builtin_amdgcn_s_waitcnt(0xc07f);           // the tile's reads are done before it is rewritten
#pragma unroll
    for (int i = 0; i < 4; ++i) cur[i] = nxt[i];
  }
#undef HPX_DP_FETCH
  // c = Ainv rhs
  const double* are_ = aitre + (long)gI * ncol * ncol;
  const double* aim_ = aitim + (long)gI * ncol * ncol;
#pragma unroll
  for (int mo = 0; mo < MT; ++mo) {
    d4 cr = {0., 0., 0., 0.}, ci = {0., 0., 0., 0.};
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        const int k = 16 * m + HPX_ACC_ROW(g, v);
        const double wr = are_[(long)k * ncol + 16 * mo + li], wi = aim_[(long)k * ncol + 16 * mo + li];
        cr = mfma64(wr, ar[m][v], cr);
        cr = mfma64(-wi, ai[m][v], cr);
        ci = mfma64(wr, ai[m][v], ci);
        ci = mfma64(wi, ar[m][v], ci);
      }
    if (b0 + li < per) {
      double* out = amps + ((long)gI * per + b0 + li) * 2 * nm;
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        const int k2 = 16 * mo + HPX_ACC_ROW(g, v);
        if (k2 < nm) {
          out[2 * k2] = cr[v];
          out[2 * k2 + 1] = ci[v];
        }
      }
    }
  }
}

// ---- OQE ------------------------------------------------------------------------
// Every O(s^3) piece of hydra_pspec/oqe.py is a product with the (un-centred) DFT matrix
// M[a][j] = exp(-2 pi i a j / s) (oqe.py:7-10: m_tau = fft(e_tau)) or with the weighting R, and runs
// as a dense contraction on the f64 MFMA (k_dft, hpx_transform.hip: out[x][c] = sum_k Wbuf[k][x]
// in[k][c], all operands planar and zero padded to SP = ceil16(s)).  Q_tau = outer(conj m_tau, m_tau)
// is rank one, so the trace formulas collapse:
//   F_ab  = 1/2 conj(Wm_ba) X_ab,  X = M R M^H,  Wm = conj(M) R M^T        (oqe.py:43-50)
//   Ft_ab = 1/2 |X_ab|^2                                                     (oqe.py:53-66)
//   q_h   = 1/2 conj(M R x1) (M R x2)                                        (oqe.py:33-40, 104-114)
//   bias_tau = 1/2 diag(M (R C conj R) M^H)_tau                              (oqe.py:23-24)
//   Sig_QEN_i = 1/2 norm^2 n_i^2,  Sig_QESN_i = 1/2 norm^2 (n_i^2 + 2 s_i n_i),
//       n = diag(M (R C_noise R) M^H), s = diag(M (R C_S R) M^H)            (oqe.py:161-186)

// M planar [SP][SP] (symmetric), zero padded
__global__ void k_oqe_mmatrix(double* __restrict__ re, double* __restrict__ im, const int s, const int SP) {
  const long tot = (long)SP * SP;
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < tot; e += (long)gridDim.x * blockDim.x) {
    const int a = (int)(e / SP), j = (int)(e % SP);
    double c = 0.0, sn = 0.0;
    if (a < s && j < s) {
      const long q = ((long)a * j) % s;
      sincospi(-2.0 * (double)q / (double)s, &sn, &c);
    }
    re[e] = c;
    im[e] = sn;
  }
}

// interleaved (nb, nr, nc) c128 -> planar [nb][RP][CP] zero padded; tr: dst[c][r] = src[r][c];
// cj: conjugate
__global__ void k_oqe_to_planar(const double* __restrict__ src, double* __restrict__ re,
                                double* __restrict__ im, const int nr, const int nc, const int RP,
                                const int CP, const int tr, const int cj) {
  const int b = blockIdx.y;
  const long tot = (long)RP * CP;
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < tot; e += (long)gridDim.x * blockDim.x) {
    const int i = (int)(e / CP), j = (int)(e % CP);
    const int r = tr ? j : i, c = tr ? i : j;
    double vr = 0.0, vi = 0.0;
    if (r < nr && c < nc) {
      const long o = (((long)b * nr + r) * nc + c) * 2;
      vr = src[o];
      vi = cj ? -src[o + 1] : src[o + 1];
    }
    re[(long)b * tot + e] = vr;
    im[(long)b * tot + e] = vi;
  }
}

// planar [nb][SP][SP] transpose through a 32 x 33 LDS tile (both sides unit stride)
__global__ __launch_bounds__(256) void k_oqe_transpose(const double* __restrict__ ire,
                                                       const double* __restrict__ iim,
                                                       double* __restrict__ ore, double* __restrict__ oim,
                                                       const int SP) {
  __shared__ double tr_[32][33], ti_[32][33];
  const int b = blockIdx.z, x0 = blockIdx.x * 32, y0 = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const long base = (long)b * SP * SP;
  for (int r = ty; r < 32; r += 8)
    if (y0 + r < SP && x0 + tx < SP) {
      tr_[r][tx] = ire[base + (long)(y0 + r) * SP + x0 + tx];
      ti_[r][tx] = iim[base + (long)(y0 + r) * SP + x0 + tx];
    }
  __syncthreads();
  for (int r = ty; r < 32; r += 8)
    if (x0 + r < SP && y0 + tx < SP) {
      ore[base + (long)(x0 + r) * SP + y0 + tx] = tr_[tx][r];
      oim[base + (long)(x0 + r) * SP + y0 + tx] = ti_[tx][r];
    }
}

// variant 0: F_ab = 1/2 conj(Wm[b][a]) X[a][b] with XT[b][a] = X[a][b], WmT[a][b] = Wm[b][a];
// variant 1: Ft_ab = 1/2 |X[a][b]|^2.  Output interleaved (nb, s, s).
__global__ void k_oqe_combine(const double* __restrict__ xtre, const double* __restrict__ xtim,
                              const double* __restrict__ wtre, const double* __restrict__ wtim,
                              double* __restrict__ F, const int s, const int SP, const int variant) {
  const int b = blockIdx.y;
  for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < s * s; e += gridDim.x * blockDim.x) {
    const int a = e / s, b2 = e % s;
    const long ox = (long)b * SP * SP + (long)b2 * SP + a;      // XT[b2][a]
    const double xr = xtre[ox], xi = xtim[ox];
    double fr, fi;
    if (variant == 1) {
      fr = 0.5 * (xr * xr + xi * xi);
      fi = 0.0;
    } else {
      const long ow = (long)b * SP * SP + (long)a * SP + b2;    // WmT[a][b2] = Wm[b2][a]
      const double wr = wtre[ow], wi = -wtim[ow];
      fr = 0.5 * (wr * xr - wi * xi);
      fi = 0.5 * (wr * xi + wi * xr);
    }
    F[((long)b * s * s + e) * 2] = fr;
    F[((long)b * s * s + e) * 2 + 1] = fi;
  }
}

// q[b][p][t] = 1/2 conj(Z[t][c1]) Z[t][c2], Z planar [SP][ncv]; pairs (c1, c2) = (2p, 2p+1)
__global__ void k_oqe_qpairs(const double* __restrict__ zre, const double* __restrict__ zim,
                             double* __restrict__ q, const int s, const int SP, const int ncv,
                             const int npair) {
  const int b = blockIdx.y;
  for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < npair * s; e += gridDim.x * blockDim.x) {
    const int pidx = e / s, t = e % s;
    const long o = (long)b * SP * ncv + (long)t * ncv + 2 * pidx;
    const double ar = zre[o], ai = zim[o], br = zre[o + 1], bi = zim[o + 1];
    q[((long)b * npair * s + e) * 2] = 0.5 * (ar * br + ai * bi);
    q[((long)b * npair * s + e) * 2 + 1] = 0.5 * (ar * bi - ai * br);
  }
}

// q[b][v][t] = 1/2 conj(Z[t][v]) Z[t][h + v]  (auto estimator: columns v = R^T x, h + v = R x)
__global__ void k_oqe_qsplit(const double* __restrict__ zr, const double* __restrict__ zi,
                             double* __restrict__ q, const int s, const int SP, const int ncv, const int h,
                             const int nv) {
  const int b = blockIdx.y;
  for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < nv * s; e += gridDim.x * blockDim.x) {
    const int v = e / s, t = e % s;
    const long o = (long)b * SP * ncv + (long)t * ncv + v;
    const double ar = zr[o], ai = zi[o], br = zr[o + h], bi = zi[o + h];
    q[((long)b * nv * s + e) * 2] = 0.5 * (ar * br + ai * bi);
    q[((long)b * nv * s + e) * 2 + 1] = 0.5 * (ar * bi - ai * br);
  }
}

// out[b][tau] = sum_c T[tau][c] conj(M[tau][c])   (diag of T M^H); one wave per row
__global__ __launch_bounds__(256) void k_oqe_rowdot(const double* __restrict__ tre,
                                                    const double* __restrict__ tim,
                                                    const double* __restrict__ mre,
                                                    const double* __restrict__ mim,
                                                    double* __restrict__ out, const int s, const int SP) {
  const int b = blockIdx.y, wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int tau = blockIdx.x * 4 + wave;
  if (tau >= s) return;
  const long o = (long)b * SP * SP + (long)tau * SP, om = (long)tau * SP;
  double sr = 0.0, si = 0.0;
  for (int c = lane; c < s; c += 64) {
    const double ar = tre[o + c], ai = tim[o + c], mr = mre[om + c], mi = mim[om + c];
    sr += ar * mr + ai * mi;
    si += ai * mr - ar * mi;
  }
#pragma unroll
  for (int k = 32; k > 0; k >>= 1) {
    sr += __shfl_down(sr, k);
    si += __shfl_down(si, k);
  }
  if (lane == 0) {
    out[((long)b * s + tau) * 2] = sr;
    out[((long)b * s + tau) * 2 + 1] = si;
  }
}

// M_opt (oqe.py:77-84): M = diag(1 / F_aa), row a divided by sum_b (M F)_ab = sum_b F_ab / F_aa
__global__ __launch_bounds__(256) void k_oqe_mopt(const double* __restrict__ F, double* __restrict__ M,
                                                  const int s) {
  __shared__ double rr[4], ri[4];
  const int b = blockIdx.y, a = blockIdx.x, tid = threadIdx.x;
  const double* Fb = F + (long)b * s * s * 2;
  double sr = 0.0, si = 0.0;
  for (int c = tid; c < s; c += 256) {
    sr += Fb[((long)a * s + c) * 2];
    si += Fb[((long)a * s + c) * 2 + 1];
  }
#pragma unroll
  for (int k = 32; k > 0; k >>= 1) {
    sr += __shfl_down(sr, k);
    si += __shfl_down(si, k);
  }
  if ((tid & 63) == 0) { rr[tid >> 6] = sr; ri[tid >> 6] = si; }
  __syncthreads();
  sr = rr[0] + rr[1] + rr[2] + rr[3];
  si = ri[0] + ri[1] + ri[2] + ri[3];
  // m = (1 / F_aa) / ((1 / F_aa) * rowsum) in the reference's operation order
  const double dr = Fb[((long)a * s + a) * 2], di = Fb[((long)a * s + a) * 2 + 1];
  const double dd = dr * dr + di * di;
  const double ir = dr / dd, ii = -di / dd;                 // 1 / F_aa
  const double wr = ir * sr - ii * si, wi = ir * si + ii * sr;   // sum_b W_ab
  const double wd = wr * wr + wi * wi;
  const double mr = (ir * wr + ii * wi) / wd, mi = (ii * wr - ir * wi) / wd;
  for (int c = tid; c < s; c += 256) {
    M[((long)b * s * s + (long)a * s + c) * 2] = (c == a) ? mr : 0.0;
    M[((long)b * s * s + (long)a * s + c) * 2 + 1] = (c == a) ? mi : 0.0;
  }
}

__global__ void k_lincomb(const long n, const double a, const double* __restrict__ x, const double bcoef,
                          const double* __restrict__ y, double* __restrict__ out) {
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (long)gridDim.x * blockDim.x)
    out[e] = a * x[e] + bcoef * y[e];
}

// bump allocator over a caller workspace or an internal buffer
struct Arena {
  hpx_devbuf own;
  double* base = nullptr;
  size_t cap = 0, used = 0;
  int init(void* work, int64_t work_bytes, size_t need_doubles) {
    if (work) {
      if ((size_t)work_bytes < need_doubles * sizeof(double)) {
        hpx_set_error("workspace too small: %lld bytes given, %zu needed", (long long)work_bytes,
                      need_doubles * sizeof(double));
        return HPX_EINVAL;
      }
      base = (double*)work;
    } else {
      HPX_TRY(own.alloc(need_doubles));
      base = own.p;
    }
    cap = need_doubles;
    return HPX_OK;
  }
  double* take(size_t n) { double* q = base + used; used += n; return q; }
};

static size_t oqe_need(int nb, int s, int nvis) {
  const size_t SP = ceil16(s), ncv = 2 * (size_t)ceil16(nvis > 0 ? nvis : 1);
  return 2 * SP * SP + (size_t)nb * (14 * SP * SP + 6 * SP * ncv) + 64;
}

// P = A B for planar operands: Wbuf = A^T (index [k][x] = A[x][k])
static int gemm(int nb, int SP, int ncol, const double* atre, const double* atim, long a_bstride, int conjA,
                const double* bre, const double* bim, double* ore, double* oim, hipStream_t st) {
  return hpx_launch_dft(nb, SP, ncol, atre, atim, conjA, bre, bim, (long)SP * ncol, ncol, nullptr, 0, ore, oim,
                        (long)SP * ncol, ncol, 1.0, st, 0, a_bstride);
}

static int finish(Arena& A, hipStream_t st, const char* who) {
  hipError_t e = hipGetLastError();
  if (e == hipSuccess && A.own.p) e = hipStreamSynchronize(st);     // internal workspace dies on return
  if (e != hipSuccess) { hpx_set_error("%s: %s", who, hipGetErrorString(e)); return HPX_EHIP; }
  return HPX_OK;
}

}  // namespace

extern "C" int hpx_mfma_probe(const double* a_host, const double* b_host, double* d_host) {
  HPX_REQUIRE(a_host && b_host && d_host, "hpx_mfma_probe: null argument");
  hpx_devbuf buf;
  HPX_TRY(buf.alloc(64 + 64 + 256));
  double *a = buf.p, *b = a + 64, *d = b + 64;
  HPX_HIP(hipMemcpy(a, a_host, 64 * 8, hipMemcpyHostToDevice));
  HPX_HIP(hipMemcpy(b, b_host, 64 * 8, hipMemcpyHostToDevice));
  hipLaunchKernelGGL(k_probe, dim3(1), dim3(64), 0, 0, a, b, d);
  HPX_HIP(hipGetLastError());
  HPX_HIP(hipMemcpy(d_host, d, 256 * 8, hipMemcpyDeviceToHost));
  return HPX_OK;
}

extern "C" int hpx_mfma_f64_peak(int iters, double* tflops_host) {
  HPX_REQUIRE(iters > 0 && tflops_host, "hpx_mfma_f64_peak: bad argument");
  int dev = 0, ncu = 0;
  HPX_HIP(hipGetDevice(&dev));
  HPX_HIP(hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev));
  const int nblk = ncu * 2;   // 2 workgroups of 4 waves per CU = 2 waves per SIMD
  hpx_devbuf sink;
  HPX_TRY(sink.alloc((size_t)nblk * 256));
  hpx_event e0, e1;
  HPX_TRY(e0.create());
  HPX_TRY(e1.create());
  hipLaunchKernelGGL(k_mfma_peak, dim3(nblk), dim3(256), 0, 0, sink.p, iters);   // warm-up
  HPX_HIP(hipEventRecord(e0.e, 0));
  hipLaunchKernelGGL(k_mfma_peak, dim3(nblk), dim3(256), 0, 0, sink.p, iters);
  HPX_HIP(hipEventRecord(e1.e, 0));
  HPX_HIP(hipEventSynchronize(e1.e));
  float ms = 0.f;
  HPX_HIP(hipEventElapsedTime(&ms, e0.e, e1.e));
  *tflops_host = (double)nblk * 4.0 * iters * 4.0 * 2048.0 / (ms * 1e-3) / 1e12;
  return HPX_OK;
}

extern "C" int64_t hpx_dpss_workspace_bytes(int ngroups, int per, int N, int nm) {
  if (ngroups <= 0 || per <= 0 || N <= 0 || nm <= 0) return 0;
  const size_t NP = ceil16(N), ncol = ceil16(nm);
  // icov planar (2 NP^2) + per group: in, out, projector (3 x 2 NP ncol) + inverse (2 ncol^2) + info
  return (int64_t)((2 * NP * NP + (size_t)ngroups * (6 * NP * ncol + 2 * ncol * ncol)) * sizeof(double) +
                   (size_t)ngroups * sizeof(int32_t) + 256);
}

extern "C" int hpx_dpss_group_info(const void* work, int64_t work_bytes, int ngroups, int per, int N, int nm,
                                   int32_t* info_host, void* stream) {
  HPX_REQUIRE(work && info_host && ngroups > 0 && per > 0 && N > 0 && nm > 0, "hpx_dpss_group_info: bad argument");
  HPX_REQUIRE(work_bytes >= hpx_dpss_workspace_bytes(ngroups, per, N, nm), "hpx_dpss_group_info: workspace too small");
  const size_t NP = ceil16(N), ncol = ceil16(nm);
  const double* base = (const double*)work;      // (layout of hpx_dpss_project_grouped)
  const int32_t* info = (const int32_t*)(base + 2 * NP * NP + (size_t)ngroups * (6 * NP * ncol + 2 * ncol * ncol));
  hipStream_t st = (hipStream_t)stream;
  HPX_HIP(hipMemcpyAsync(info_host, info, (size_t)ngroups * sizeof(int32_t), hipMemcpyDeviceToHost, st));
  HPX_HIP(hipStreamSynchronize(st));
  return HPX_OK;
}

extern "C" int hpx_dpss_project_grouped(int ngroups, int per, int N, int nm, const double* d,
                                        const double* tw, const double* modes, const double* icov,
                                        double* amps, void* work, int64_t work_bytes, int reuse_projector,
                                        void* stream) {
  HPX_REQUIRE(ngroups > 0 && per > 0 && N > 0 && nm > 0 && nm <= HPX_DPSS_MAXM && d && amps,
              "hpx_dpss_project_grouped: bad argument (need 0 < nmodes <= 32)");
  HPX_REQUIRE(reuse_projector ? (work != nullptr) : (tw && modes && icov),
              "hpx_dpss_project_grouped: reuse_projector needs the caller workspace of the call that built it; "
              "otherwise tw, modes and icov are required");
  hipStream_t st = (hipStream_t)stream;
  const int NP = ceil16(N), ncol = ceil16(nm);
  const int64_t need = hpx_dpss_workspace_bytes(ngroups, per, N, nm);
  hpx_devbuf own;
  double* base = (double*)work;
  if (!base) {
    HPX_TRY(own.alloc((size_t)need / sizeof(double) + 1));
    base = own.p;
  } else {
    HPX_REQUIRE(work_bytes >= need, "hpx_dpss_project_grouped: workspace too small (hpx_dpss_workspace_bytes)");
  }
  const size_t wsz = (size_t)NP * NP, gsz = (size_t)ngroups * NP * ncol, asz = (size_t)ngroups * ncol * ncol;
  double *wre = base, *wim = wre + wsz, *ire = wim + wsz, *iim = ire + gsz, *ore = iim + gsz, *oim = ore + gsz,
         *ptre = oim + gsz, *ptim = ptre + gsz, *aitre = ptim + gsz, *aitim = aitre + asz;
  int32_t* info = (int32_t*)(aitim + asz);
  if (!reuse_projector) {
    hipLaunchKernelGGL(k_herm_planar, dim3(256), dim3(256), 0, st, icov, wre, wim, N, NP);
    hipLaunchKernelGGL(k_dpss_in, dim3(16, ngroups), dim3(256), 0, st, tw, modes, ire, iim, N, nm, NP, ncol);
    HPX_HIP(hipGetLastError());
    // O = Ah Mw: the stored planar matrix is Ah^T = conj(Ah), hence conjW = 1
    HPX_TRY(hpx_launch_dft(ngroups, NP, ncol, wre, wim, 1, ire, iim, (long)NP * ncol, ncol, nullptr, 0,
                           ore, oim, (long)NP * ncol, ncol, 1.0, st, 0));
    hipLaunchKernelGGL(k_dpss_group, dim3(ngroups), dim3(256), 0, st, tw, modes, ore, oim, ptre, ptim, aitre,
                       aitim, info, N, nm, NP, ncol);
    HPX_HIP(hipGetLastError());
  }
  dim3 grid((per + 63) / 64, ngroups);
  if (ncol == 16)
    hipLaunchKernelGGL(k_dpss_apply<1>, grid, dim3(256), 0, st, d, ptre, ptim, aitre, aitim, amps, per, N, nm, NP, ncol);
  else
    hipLaunchKernelGGL(k_dpss_apply<2>, grid, dim3(256), 0, st, d, ptre, ptim, aitre, aitim, amps, per, N, nm, NP, ncol);
  HPX_HIP(hipGetLastError());
  if (!work) {     // internal workspace: it must outlive the kernels; a caller workspace keeps the call asynchronous
    std::vector<int32_t> h(ngroups);
    HPX_HIP(hipMemcpyAsync(h.data(), info, (size_t)ngroups * sizeof(int32_t), hipMemcpyDeviceToHost, st));
    HPX_HIP(hipStreamSynchronize(st));
    for (int g = 0; g < ngroups; ++g)
      if (h[g]) {
        hpx_set_error("hpx_dpss_project_grouped: weighted normal matrix of group %d is not positive definite", g);
        return HPX_ENOTPD;
      }
  }
  return HPX_OK;
}

extern "C" int hpx_dpss_project(int nb, int N, int nm, const double* d, const double* tw,
                                const double* modes, const double* icov, double* amps,
                                void* stream) {
  // every spectrum its own weights: groups of one
  return hpx_dpss_project_grouped(nb, 1, N, nm, d, tw, modes, icov, amps, nullptr, 0, 0, stream);
}

extern "C" int64_t hpx_oqe_workspace_bytes(int nb, int s, int nvis) {
  if (nb <= 0 || s <= 0) return 0;
  return (int64_t)(oqe_need(nb, s, nvis) * sizeof(double));
}

extern "C" int hpx_oqe_fisher(int nb, int s, const double* R, double* F_out, int variant,
                              void* work, int64_t work_bytes, void* stream) {
  HPX_REQUIRE(nb > 0 && s > 0 && R && F_out && (variant == 0 || variant == 1),
              "hpx_oqe_fisher: bad argument");
  hipStream_t st = (hipStream_t)stream;
  const int SP = ceil16(s);
  const size_t m2 = (size_t)SP * SP, b2 = (size_t)nb * m2;
  Arena A;
  HPX_TRY(A.init(work, work_bytes, oqe_need(nb, s, 0)));
  double *mre = A.take(m2), *mim = A.take(m2), *rre = A.take(b2), *rim = A.take(b2), *t1re = A.take(b2),
         *t1im = A.take(b2), *ttre = A.take(b2), *ttim = A.take(b2), *xtre = A.take(b2), *xtim = A.take(b2),
         *wtre = A.take(b2), *wtim = A.take(b2);
  const dim3 tgrid((SP + 31) / 32, (SP + 31) / 32, nb);
  hipLaunchKernelGGL(k_oqe_mmatrix, dim3(256), dim3(256), 0, st, mre, mim, s, SP);
  hipLaunchKernelGGL(k_oqe_to_planar, dim3(64, nb), dim3(256), 0, st, R, rre, rim, s, s, SP, SP, 0, 0);
  HPX_HIP(hipGetLastError());
  // X = M R M^H:  T1 = M R;  XT = conj(M) T1^T  (XT[b][a] = X[a][b])
  HPX_TRY(gemm(nb, SP, SP, mre, mim, 0, 0, rre, rim, t1re, t1im, st));
  hipLaunchKernelGGL(k_oqe_transpose, tgrid, dim3(256), 0, st, t1re, t1im, ttre, ttim, SP);
  HPX_TRY(gemm(nb, SP, SP, mre, mim, 0, 1, ttre, ttim, xtre, xtim, st));
  if (variant == 0) {   // Wm = conj(M) R M^T:  T1' = conj(M) R;  WmT = M T1'^T
    HPX_TRY(gemm(nb, SP, SP, mre, mim, 0, 1, rre, rim, t1re, t1im, st));
    hipLaunchKernelGGL(k_oqe_transpose, tgrid, dim3(256), 0, st, t1re, t1im, ttre, ttim, SP);
    HPX_TRY(gemm(nb, SP, SP, mre, mim, 0, 0, ttre, ttim, wtre, wtim, st));
  }
  hipLaunchKernelGGL(k_oqe_combine, dim3((s * s + 255) / 256, nb), dim3(256), 0, st, xtre, xtim, wtre, wtim,
                     F_out, s, SP, variant);
  return finish(A, st, "hpx_oqe_fisher");
}

// Z = M (Rop V^T) for the rows of V (nb, nv, s): the two transforms of the estimators.
// which = 0: Rop = R for every column;  which = 1: columns interleaved (R^T x, R x) per visibility
static int oqe_estimate(int nb, int nv, int s, const double* R, const double* V, double* q_out, int which,
                        void* work, int64_t work_bytes, hipStream_t st, const char* who) {
  const int SP = ceil16(s);
  const int ncv = which ? 2 * ceil16(nv) : ceil16(nv);
  const size_t m2 = (size_t)SP * SP, b2 = (size_t)nb * m2, v2 = (size_t)nb * SP * ncv;
  Arena A;
  HPX_TRY(A.init(work, work_bytes, oqe_need(nb, s, nv)));
  double *mre = A.take(m2), *mim = A.take(m2), *rtre = A.take(b2), *rtim = A.take(b2), *vre = A.take(v2),
         *vim = A.take(v2), *yre = A.take(v2), *yim = A.take(v2), *zre = A.take(v2), *zim = A.take(v2);
  hipLaunchKernelGGL(k_oqe_mmatrix, dim3(256), dim3(256), 0, st, mre, mim, s, SP);
  // V^T planar [k][v] (zero padded)
  hipLaunchKernelGGL(k_oqe_to_planar, dim3(64, nb), dim3(256), 0, st, V, vre, vim, nv, s, SP, which ? ncv / 2 : ncv, 1, 0);
  HPX_HIP(hipGetLastError());
  if (!which) {
    // Y = R V^T: Wbuf = R^T
    hipLaunchKernelGGL(k_oqe_to_planar, dim3(64, nb), dim3(256), 0, st, R, rtre, rtim, s, s, SP, SP, 1, 0);
    HPX_TRY(gemm(nb, SP, ncv, rtre, rtim, (long)m2, 0, vre, vim, yre, yim, st));
    HPX_TRY(gemm(nb, SP, ncv, mre, mim, 0, 0, yre, yim, zre, zim, st));
    hipLaunchKernelGGL(k_oqe_qpairs, dim3((nv / 2 * s + 255) / 256, nb), dim3(256), 0, st, zre, zim, q_out, s, SP,
                       ncv, nv / 2);
  } else {
    // columns [0, h): R^T x (Wbuf = R), columns [h, 2h): R x (Wbuf = R^T); h = ncv / 2
    const int h = ncv / 2;
    hipLaunchKernelGGL(k_oqe_to_planar, dim3(64, nb), dim3(256), 0, st, R, rtre, rtim, s, s, SP, SP, 0, 0);
    HPX_TRY(hpx_launch_dft(nb, SP, h, rtre, rtim, 0, vre, vim, (long)SP * h, h, nullptr, 0, yre, yim,
                           (long)SP * ncv, ncv, 1.0, st, 0, (long)m2));
    hipLaunchKernelGGL(k_oqe_to_planar, dim3(64, nb), dim3(256), 0, st, R, rtre, rtim, s, s, SP, SP, 1, 0);
    HPX_TRY(hpx_launch_dft(nb, SP, h, rtre, rtim, 0, vre, vim, (long)SP * h, h, nullptr, 0, yre + h, yim + h,
                           (long)SP * ncv, ncv, 1.0, st, 0, (long)m2));
    HPX_TRY(gemm(nb, SP, ncv, mre, mim, 0, 0, yre, yim, zre, zim, st));
    // q[v][t] = 1/2 conj(Z[t][v]) Z[t][h + v]
    hipLaunchKernelGGL(k_oqe_qsplit, dim3((nv * s + 255) / 256, nb), dim3(256), 0, st, zre, zim, q_out, s, SP, ncv, h, nv);
  }
  return finish(A, st, who);
}

extern "C" int hpx_oqe_qh(int nb, int npair, int s, const double* R, const double* V,
                          double* q_out, void* work, int64_t work_bytes, void* stream) {
  HPX_REQUIRE(nb > 0 && npair > 0 && s > 0 && R && V && q_out, "hpx_oqe_qh: bad argument");
  return oqe_estimate(nb, 2 * npair, s, R, V, q_out, 0, work, work_bytes, (hipStream_t)stream, "hpx_oqe_qh");
}

extern "C" int hpx_oqe_qauto(int nb, int nvis, int s, const double* R, const double* V,
                             double* q_out, void* work, int64_t work_bytes, void* stream) {
  HPX_REQUIRE(nb > 0 && nvis > 0 && s > 0 && R && V && q_out, "hpx_oqe_qauto: bad argument");
  return oqe_estimate(nb, nvis, s, R, V, q_out, 1, work, work_bytes, (hipStream_t)stream, "hpx_oqe_qauto");
}

extern "C" int hpx_oqe_sandwich_diag(int nb, int s, const double* R, const double* Cm, int conj_right,
                                     double* out, void* work, int64_t work_bytes, void* stream) {
  HPX_REQUIRE(nb > 0 && s > 0 && R && Cm && out, "hpx_oqe_sandwich_diag: bad argument");
  hipStream_t st = (hipStream_t)stream;
  const int SP = ceil16(s);
  const size_t m2 = (size_t)SP * SP, b2 = (size_t)nb * m2;
  Arena A;
  HPX_TRY(A.init(work, work_bytes, oqe_need(nb, s, 0)));
  double *mre = A.take(m2), *mim = A.take(m2), *rtre = A.take(b2), *rtim = A.take(b2), *cre = A.take(b2),
         *cim = A.take(b2), *t2re = A.take(b2), *t2im = A.take(b2), *ttre = A.take(b2), *ttim = A.take(b2),
         *rrre = A.take(b2), *rrim = A.take(b2), *bre = A.take(b2), *bim = A.take(b2);
  const dim3 tgrid((SP + 31) / 32, (SP + 31) / 32, nb);
  hipLaunchKernelGGL(k_oqe_mmatrix, dim3(256), dim3(256), 0, st, mre, mim, s, SP);
  hipLaunchKernelGGL(k_oqe_to_planar, dim3(64, nb), dim3(256), 0, st, R, rtre, rtim, s, s, SP, SP, 1, 0);      // R^T
  hipLaunchKernelGGL(k_oqe_to_planar, dim3(64, nb), dim3(256), 0, st, Cm, cre, cim, s, s, SP, SP, 0, 0);       // C
  hipLaunchKernelGGL(k_oqe_to_planar, dim3(64, nb), dim3(256), 0, st, R, rrre, rrim, s, s, SP, SP, 0, conj_right);   // R or conj R
  HPX_HIP(hipGetLastError());
  HPX_TRY(gemm(nb, SP, SP, rtre, rtim, (long)m2, 0, cre, cim, t2re, t2im, st));          // T2 = R C
  hipLaunchKernelGGL(k_oqe_transpose, tgrid, dim3(256), 0, st, t2re, t2im, ttre, ttim, SP);
  HPX_TRY(gemm(nb, SP, SP, ttre, ttim, (long)m2, 0, rrre, rrim, bre, bim, st));          // B = T2 R'
  HPX_TRY(gemm(nb, SP, SP, mre, mim, 0, 0, bre, bim, t2re, t2im, st));                   // T = M B
  hipLaunchKernelGGL(k_oqe_rowdot, dim3((s + 3) / 4, nb), dim3(256), 0, st, t2re, t2im, mre, mim, out, s, SP);
  return finish(A, st, "hpx_oqe_sandwich_diag");
}

extern "C" int hpx_oqe_mopt(int nb, int s, const double* F, double* M_out, void* stream) {
  HPX_REQUIRE(nb > 0 && s > 0 && F && M_out, "hpx_oqe_mopt: bad argument");
  hipLaunchKernelGGL(k_oqe_mopt, dim3(s, nb), dim3(256), 0, (hipStream_t)stream, F, M_out, s);
  HPX_HIP(hipGetLastError());
  return HPX_OK;
}

// beta_k = sum_t |sk[t][k]|^2  (sample_S, pspec.py:96-100), (nb, T, N) c128 -> (nb, N) f64
__global__ void k_power_sum(const double* __restrict__ sk, double* __restrict__ out, const int T, const int N) {
  const int b = blockIdx.y;
  for (int k = blockIdx.x * blockDim.x + threadIdx.x; k < N; k += gridDim.x * blockDim.x) {
    double acc = 0.0;
    for (int t = 0; t < T; ++t) {                       // threads along k: unit-stride reads, fixed order
      const long o = (((long)b * T + t) * N + k) * 2;
      acc += sk[o] * sk[o] + sk[o + 1] * sk[o + 1];
    }
    out[(long)b * N + k] = acc;
  }
}

extern "C" int hpx_power_sum(int nb, int T, int N, const double* sk, double* out, void* stream) {
  HPX_REQUIRE(nb > 0 && T > 0 && N > 0 && sk && out, "hpx_power_sum: bad argument");
  hipLaunchKernelGGL(k_power_sum, dim3((N + 255) / 256, nb), dim3(256), 0, (hipStream_t)stream, sk, out, T, N);
  HPX_HIP(hipGetLastError());
  return HPX_OK;
}

extern "C" int hpx_lincomb(int64_t n, double a, const double* x, double b, const double* y, double* out,
                           void* stream) {
  HPX_REQUIRE(n > 0 && x && y && out, "hpx_lincomb: bad argument");
  hipLaunchKernelGGL(k_lincomb, dim3(256), dim3(256), 0, (hipStream_t)stream, (long)n, a, x, b, y, out);
  HPX_HIP(hipGetLastError());
  return HPX_OK;
}
