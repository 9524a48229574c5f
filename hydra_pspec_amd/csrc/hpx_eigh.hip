// Batched Hermitian positive semi-definite eigendecomposition for orders 128 .. 1024: blocked ONE-SIDED Jacobi on
// the Cholesky factor, the rotations of a pair of 8-column blocks applied as one 16 x 16 unitary on the f64 MFMA.
//
// Why: when Ntimes > Nfreqs -- or both are large -- the reference diagonalises an
// n x n covariance per baseline with n up to the channel count (scripts/calc-vis-cov-matrices.py:239-247).  The cyclic
// two-sided Jacobi of hpx_modes.hip (one workgroup per baseline, scalar rotations, the matrices streamed through
// the cache hierarchy n - 1 times per sweep) is right for n = Ntimes = 32 and takes seconds per baseline at n = 512.
//
// Method.  C + eps I = L L^H (the batched Cholesky of hpx_factor*.hip; eps = 8 n 2^-52 trace(C) makes the exactly
// singular Gram matrix of a centred cube factorisable and is subtracted from the eigenvalues again).  The eigenvectors of
// L L^H are the left singular vectors of L: rotate the COLUMNS of W = L from the right until they are mutually
// orthogonal (Hestenes); then lambda_j = |w_j|^2 and u_j = w_j / |w_j|.  Blocked: the n columns form p = n / 8 blocks,
// a step pairs the blocks off by the round-robin schedule (p / 2 disjoint pairs, p - 1 steps per sweep), and one
// workgroup per pair
//   1. forms the 16 x 16 Gram matrix of its 16 columns (K = n on the MFMA, the four waves split the rows),
//   2. runs one sweep of two-sided Jacobi on it in LDS (one entry per thread) accumulating the unitary Q,
//   3. replaces the 16 columns by W Q (MFMA again; second read of the columns from L2).
// The columns are stored block-wise, [block][row][8 columns], planar: both passes read and write whole 64-byte rows
// of a block, 256 contiguous bytes per k-step.  A sweep moves 2 n^2 x 16 B per matrix through HBM per step ... the
// method is bandwidth-bound: 1024 matrices of order 512 take about a second (docs/HISTORY.md section 11.9).
#include "hpx_internal.h"
#include "../../include/hpx.h"

#ifndef HPX_EIGH_MAX_SWEEPS
#define HPX_EIGH_MAX_SWEEPS 30
#endif

namespace {

__device__ __forceinline__ double hj_rsqrt(const double d) {
  double q = __builtin_amdgcn_rsq(d);
  q = fma(q * 0.5, fma(-d * q, q, 1.0), q);
  q = fma(q * 0.5, fma(-d * q, q, 1.0), q);
  return q;
}
__device__ __forceinline__ double hj_rcp(const double d) {
  double r = __builtin_amdgcn_rcp(d);
  r = fma(r, fma(-d, r, 1.0), r);
  r = fma(r, fma(-d, r, 1.0), r);
  return r;
}
__device__ __forceinline__ void rr_pair16(const int n, const int s, const int i, int& p, int& q) {
  const int m = n - 1;
  int a, b;
  if (i == 0) {
    a = m;
    b = s % m;
  } else {
    a = (s + i) % m;
    b = (s - i + m) % m;
  }
  p = min(a, b);
  q = max(a, b);
}

// ridge[b] = 8 n eps trace(G) (>= the rounding error of the Gram matrix and of its factorisation, both ~ n eps ||G||:
// an exactly singular Gram matrix, or one with eigenvalues 1e-16 of the largest, must still factor);
// A (interleaved) = G + ridge I
__global__ __launch_bounds__(256) void k_hj_prep(const double* __restrict__ gr, const double* __restrict__ gi,
                                                 double* __restrict__ a, double* __restrict__ ridge, const int n) {
  __shared__ double red[256];
  const int b = blockIdx.x, tid = threadIdx.x;
  const double* r = gr + (long)b * n * n;
  const double* im = gi + (long)b * n * n;
  double best = 0.0;
  for (int i = tid; i < n; i += 256) best += fabs(r[(long)i * n + i]);
  red[tid] = best;
  __syncthreads();
  for (int k = 128; k > 0; k >>= 1) {
    if (tid < k) red[tid] += red[tid + k];
    __syncthreads();
  }
  const double eps = red[0] > 0.0 ? 8.0 * n * 2.220446049250313e-16 * red[0] : 1e-300;
  if (tid == 0) ridge[b] = eps;
  double* ab = a + (long)b * n * n * 2;
  for (long e = tid; e < (long)n * n; e += 256) {
    const int i = (int)(e / n), j = (int)(e % n);
    // (the Hermitian part: the Gram kernel's two triangles agree to rounding only)
    const long t = (long)j * n + i;
    ab[e * 2] = 0.5 * (r[e] + r[t]) + (i == j ? eps : 0.0);
    ab[e * 2 + 1] = (i == j) ? 0.0 : 0.5 * (im[e] - im[t]);
  }
}

// W = L (lower triangle of the interleaved factor, zero above) in block layout [n / NB][n rows][NB]
__global__ void k_hj_pack(const double* __restrict__ l, double* __restrict__ wre, double* __restrict__ wim, const int n,
                          const int NB) {
  const int b = blockIdx.y;
  const double* lb = l + (long)b * n * n * 2;
  double* wr = wre + (long)b * n * n;
  double* wi = wim + (long)b * n * n;
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < (long)n * n; e += (long)gridDim.x * blockDim.x) {
    const int row = (int)(e / n), col = (int)(e % n);
    const long o = ((long)(col / NB) * n + row) * NB + (col % NB);
    const bool low = col <= row;
    wr[o] = low ? lb[e * 2] : 0.0;
    wi[o] = low ? lb[e * 2 + 1] : 0.0;
  }
}

// one step of the round-robin schedule: workgroup (pair, baseline).  NB = columns per block (8 or 16): a pair is
// M = 2 NB columns, its Gram matrix M x M.  With NB = 16 a sweep has half as many steps -- every step streams the
// whole factor through HBM once in and once out, which is what bounds the method -- for the same MFMA work.
template <int NB>
__global__ __launch_bounds__(256, 2) void k_hj_step(double* __restrict__ wre, double* __restrict__ wim, const int n,
                                                    const int p, const int s, unsigned long long* __restrict__ meas,
                                                    const int inner_sweeps) {
  constexpr int M = 2 * NB, MM = M * M, MT = M / 16;      // MT x MT tiles of 16 x 16
  constexpr int EPT = MM / 256;                           // entries per thread of the small problem
  __shared__ double Ga[2][2][MM], Qa[2][2][MM];
  __shared__ double rot[M / 2][4];
  __shared__ double partbuf[(MT == 1) ? 2048 : 1];        // NB = 8: the four waves' partial Gram matrices
  __shared__ double red[4];
  __shared__ int skip;
  const int b = blockIdx.y, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, li = lane & 15, g = lane >> 4;
  int bi, bj;
  rr_pair16(p, s, blockIdx.x, bi, bj);
  double* wr = wre + (long)b * n * n;
  double* wi = wim + (long)b * n * n;
  const long offA = (long)bi * n * NB, offB = (long)bj * n * NB;
  const int ntile = n >> 4;                         // 16-row tiles
  // column c (0 .. M-1) of the pair, row r: offset of the element
#define HPX_HJ_OFF(c_, r_) ((((c_) < NB) ? offA : offB) + (long)(r_) * NB + ((c_) & (NB - 1)))
  // ---- 1. Gram matrix of the pair's M columns: G[i][j] = sum_k conj(W[k][i]) W[k][j]
  if (MT == 1) {          // one tile: the four waves split the rows, partial sums through LDS
    d4 grr = {0., 0., 0., 0.}, gii = grr, gri = grr, gir = grr;
    for (int t = wave; t < ntile; t += 4) {
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        const long o = HPX_HJ_OFF(li, 16 * t + 4 * ks + g);
        const double xr = wr[o], xi = wi[o];
        grr = mfma64(xr, xr, grr);
        gii = mfma64(xi, xi, gii);
        gri = mfma64(xr, xi, gri);
        gir = mfma64(xi, xr, gir);
      }
    }
    double* part = partbuf;                         // [wave][re | im][256]
#pragma unroll
    for (int v = 0; v < 4; ++v) {                   // accumulator: row g + 4 v, column li
      part[(wave * 2 + 0) * 256 + HPX_ACC_ROW(g, v) * 16 + li] = grr[v] + gii[v];
      part[(wave * 2 + 1) * 256 + HPX_ACC_ROW(g, v) * 16 + li] = gri[v] - gir[v];
    }
    __syncthreads();
    const double sre = part[0 * 256 + tid] + part[2 * 256 + tid] + part[4 * 256 + tid] + part[6 * 256 + tid];
    const double sim = part[1 * 256 + tid] + part[3 * 256 + tid] + part[5 * 256 + tid] + part[7 * 256 + tid];
    __syncthreads();
    Ga[0][0][tid] = sre;
    Ga[0][1][tid] = ((tid >> 4) == (tid & 15)) ? 0.0 : sim;
  } else {                // 2 x 2 tiles: wave w forms tile (w >> 1, w & 1) over all the rows, nothing to reduce
    // conj(a) b = (ar br + ai bi) + i (ar bi - ai br) from three products: P1 = ar br, P2 = ai bi,
    // P3 = (ar + ai)(br - bi) = P1 - ar bi + ai br - P2  ->  im = P1 - P2 - P3.  Operands one row tile ahead.
    const int ti = wave >> 1, tj = wave & 1;
    d4 p1 = {0., 0., 0., 0.}, p2 = p1, p3 = p1;
    double ar[2][4], ai[2][4], br[2][4], bm[2][4];
#define HPX_HJ_GLOAD(S_, t_)                                                         \
    _Pragma("unroll") for (int ks = 0; ks < 4; ++ks) {                               \
      const int row = 16 * (t_) + 4 * ks + g;                                        \
      const long oa = HPX_HJ_OFF(16 * ti + li, row), ob = HPX_HJ_OFF(16 * tj + li, row); \
      ar[S_][ks] = wr[oa]; ai[S_][ks] = wi[oa]; br[S_][ks] = wr[ob]; bm[S_][ks] = wi[ob]; \
    }
#define HPX_HJ_GMMA(S_)                                                              \
    _Pragma("unroll") for (int ks = 0; ks < 4; ++ks) {                               \
      p1 = mfma64(ar[S_][ks], br[S_][ks], p1);                                       \
      p2 = mfma64(ai[S_][ks], bm[S_][ks], p2);                                       \
      p3 = mfma64(ar[S_][ks] + ai[S_][ks], br[S_][ks] - bm[S_][ks], p3);             \
    }
    HPX_HJ_GLOAD(0, 0)
    for (int t = 0; t < ntile; t += 2) {
      HPX_HJ_GLOAD(1, min(t + 1, ntile - 1))
      __builtin_amdgcn_sched_barrier(0);
      HPX_HJ_GMMA(0)
      __builtin_amdgcn_sched_barrier(0);
      HPX_HJ_GLOAD(0, min(t + 2, ntile - 1))
      __builtin_amdgcn_sched_barrier(0);
      if (t + 1 < ntile) { HPX_HJ_GMMA(1) }
      __builtin_amdgcn_sched_barrier(0);
    }
#undef HPX_HJ_GLOAD
#undef HPX_HJ_GMMA
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      const int rr_ = 16 * ti + HPX_ACC_ROW(g, v), cc_ = 16 * tj + li;
      Ga[0][0][rr_ * M + cc_] = p1[v] + p2[v];
      Ga[0][1][rr_ * M + cc_] = (rr_ == cc_) ? 0.0 : p1[v] - p2[v] - p3[v];
    }
  }
  __syncthreads();
#pragma unroll
  for (int q = 0; q < EPT; ++q) {
    const int e = tid + 256 * q;
    Qa[0][0][e] = ((e / M) == (e % M)) ? 1.0 : 0.0;
    Qa[0][1][e] = 0.0;
  }
  {   // how far from orthogonal the M columns were: max |G_rc|^2 / (G_rr G_cc)
    double m = 0.0;
#pragma unroll
    for (int q = 0; q < EPT; ++q) {
      const int e = tid + 256 * q, r = e / M, c = e % M;
      if (r < c) {
        const double off = Ga[0][0][e] * Ga[0][0][e] + Ga[0][1][e] * Ga[0][1][e];
        const double dd = Ga[0][0][r * (M + 1)] * Ga[0][0][c * (M + 1)];
        m = fmax(m, (dd > 0.0) ? off / dd : (off > 0.0 ? 1.0 : 0.0));
        // fmax drops a NaN and the comparisons above turn one into 0: a non-finite Gram entry must not read as
        // "orthogonal already" (+inf has the largest bit pattern of all non-negative doubles: it wins the atomicMax)
        if (!(off <= 1.7e308) || !(fabs(dd) <= 1.7e308)) m = __builtin_inf();
      }
    }
    for (int o = 32; o > 0; o >>= 1) m = fmax(m, __shfl_xor(m, o, 64));
    if (lane == 0) red[wave] = m;
    __syncthreads();
    if (tid == 0) {
      const double w = fmax(fmax(red[0], red[1]), fmax(red[2], red[3]));
      atomicMax(&meas[b], (unsigned long long)__double_as_longlong(w));
      skip = w < 1e-31;                             // orthogonal to rounding already: nothing to rotate, nothing to write
    }
    __syncthreads();
    if (skip) return;
  }
  // ---- 2. two-sided Jacobi on G, Q accumulates the rotations.  A step rotates M / 2 disjoint pairs; entry (r, c) of
  // G' = J^H G J needs G at {r, partner r} x {c, partner c} -- a 2 x 2 block that maps to itself.  One THREAD per
  // block (pair of rows i, pair of columns j: (M/2)^2 = 256 blocks for M = 32): it reads its four entries of G and of
  // Q once and the two rotations from the step's table.  (Round 4: one entry per thread -- eight reads of G, four of Q
  // and four table look-ups per entry.)  The rotations themselves stay on 16 lanes of ONE wave: their square roots
  // and divisions are fp64 vector work, which shares the pipe with the co-resident workgroup's MFMAs -- formed by
  // every thread for itself (no table, one barrier less) the kernel took 2.2 - 2.7 s instead of 1.6.  Same operations
  // per entry, in the same order, as before.
  // (rotating only the pairs (column of block I, column of block J), M / 2 steps instead of M - 1, does not
  // converge: the columns inside a block have to meet each other too)
  int cur = 0;
  {
    const int bi_ = tid / (M / 2), bj_ = tid % (M / 2);
    const bool act = tid < (M / 2) * (M / 2);
    for (int sw = 0; sw < inner_sweeps; ++sw)
      for (int st = 0; st < M - 1; ++st) {
        if (tid < M / 2) {
          int pp, qq;
          rr_pair16(M, st, tid, pp, qq);
          const double a = Ga[cur][0][pp * (M + 1)], bq = Ga[cur][0][qq * (M + 1)];
          const double cr = Ga[cur][0][pp * M + qq], ci = Ga[cur][1][pp * M + qq];
          const double ac2 = cr * cr + ci * ci;
          double cs = 1.0, sn = 0.0, cp = 1.0, sp = 0.0;
          if (ac2 > 1e-34 * fabs(a * bq) && ac2 > 0.0) {
            // reciprocal square roots / reciprocal by the hardware seed + two Newton steps (2e-16) instead of three
            // IEEE square roots and four divisions: this chain, on one wave, is what a step waits for
            const double iac = hj_rsqrt(ac2);                      // 1 / |c|
            const double tau = (bq - a) * 0.5 * iac;
            const double h2 = 1.0 + tau * tau;
            // (tau^2 overflowing makes h2 = inf and rsq(inf) * inf a NaN: the rotation angle is 0 there)
            const double t = (h2 <= 1.7e308) ? ((tau >= 0.0) ? 1.0 : -1.0) * hj_rcp(fabs(tau) + h2 * hj_rsqrt(h2)) : 0.0;
            cs = hj_rsqrt(1.0 + t * t);
            sn = t * cs;
            cp = cr * iac;
            sp = ci * iac;
          }
          rot[tid][0] = cs; rot[tid][1] = sn; rot[tid][2] = cp; rot[tid][3] = sp;
        }
        __syncthreads();
        if (act) {
          const double* gre = Ga[cur][0];
          const double* gim = Ga[cur][1];
          const double* qre = Qa[cur][0];
          const double* qim = Qa[cur][1];
          int rw[2], cl[2];
          rr_pair16(M, st, bi_, rw[0], rw[1]);
          rr_pair16(M, st, bj_, cl[0], cl[1]);
          const double csr = rot[bi_][0], snr = rot[bi_][1], cpr = rot[bi_][2], spr = rot[bi_][3];
          const double csc = rot[bj_][0], snc = rot[bj_][1], cpc = rot[bj_][2], spc = rot[bj_][3];
          // x_a' = own_a x_a + part_a x_partner(a):  a = p: (cs, -sn e^{-i phi});  a = q: (cs, sn e^{i phi})
          double g_r[2][2], g_i[2][2], q_r[2][2], q_i[2][2];
#pragma unroll
          for (int x = 0; x < 2; ++x)
#pragma unroll
            for (int y = 0; y < 2; ++y) {
              g_r[x][y] = gre[rw[x] * M + cl[y]]; g_i[x][y] = gim[rw[x] * M + cl[y]];
              q_r[x][y] = qre[rw[x] * M + cl[y]]; q_i[x][y] = qim[rw[x] * M + cl[y]];
            }
#pragma unroll
          for (int x = 0; x < 2; ++x)
#pragma unroll
            for (int y = 0; y < 2; ++y) {
              const double own_c = csc, pcr = y ? snc * cpc : -snc * cpc, pci = snc * spc;
              const double own_r = csr, prr = x ? snr * cpr : -snr * cpr, pri = snr * spr;
              // T = G J (columns), at rows r and partner(r)
              const double t1r = g_r[x][y] * own_c + g_r[x][1 - y] * pcr - g_i[x][1 - y] * pci;
              const double t1i = g_i[x][y] * own_c + g_r[x][1 - y] * pci + g_i[x][1 - y] * pcr;
              const double t2r = g_r[1 - x][y] * own_c + g_r[1 - x][1 - y] * pcr - g_i[1 - x][1 - y] * pci;
              const double t2i = g_i[1 - x][y] * own_c + g_r[1 - x][1 - y] * pci + g_i[1 - x][1 - y] * pcr;
              // G' = J^H T (rows): conj(part_r) = (prr, -pri)
              const double nr = own_r * t1r + prr * t2r + pri * t2i;
              const double ni = own_r * t1i + prr * t2i - pri * t2r;
              const double qr_ = q_r[x][y] * own_c + q_r[x][1 - y] * pcr - q_i[x][1 - y] * pci;
              const double qi_ = q_i[x][y] * own_c + q_r[x][1 - y] * pci + q_i[x][1 - y] * pcr;
              const int e = rw[x] * M + cl[y];
              Ga[cur ^ 1][0][e] = nr;
              Ga[cur ^ 1][1][e] = (rw[x] == cl[y]) ? 0.0 : ni;
              Qa[cur ^ 1][0][e] = qr_;
              Qa[cur ^ 1][1][e] = qi_;
            }
        }
        __syncthreads();
        cur ^= 1;
      }
  }
  // ---- 3. W <- W Q on the pair's columns, 16 rows at a time: out[r][j] = sum_i W[r][i] Q[i][j]
  {
    double qr[MT][4 * MT], qi[MT][4 * MT];
#pragma unroll
    for (int jt = 0; jt < MT; ++jt)
#pragma unroll
      for (int ks = 0; ks < 4 * MT; ++ks) {         // B[k = 4 ks + g][n = li] = Q[k][16 jt + li]
        qr[jt][ks] = Qa[cur][0][(4 * ks + g) * M + 16 * jt + li];
        qi[jt][ks] = Qa[cur][1][(4 * ks + g) * M + 16 * jt + li];
      }
    // x q = (xr qr - xi qi) + i (xr qi + xi qr) from three products: P1 = xr qr, P2 = xi qi,
    // P3 = (xr + xi)(qr + qi)  ->  im = P3 - P1 - P2.  The tile's operands are loaded one tile ahead.
    double qs[MT][4 * MT];
#pragma unroll
    for (int jt = 0; jt < MT; ++jt)
#pragma unroll
      for (int ks = 0; ks < 4 * MT; ++ks) qs[jt][ks] = qr[jt][ks] + qi[jt][ks];
    double xr[2][4 * MT], xi[2][4 * MT];
#define HPX_HJ_ULOAD(S_, t_)                                                         \
    _Pragma("unroll") for (int ks = 0; ks < 4 * MT; ++ks) {                          \
      const long o = HPX_HJ_OFF(4 * ks + g, 16 * (t_) + li);                         \
      xr[S_][ks] = wr[o]; xi[S_][ks] = wi[o];                                        \
    }
#define HPX_HJ_UTILE(S_, t_)                                                         \
    {                                                                                \
      d4 u1[MT], u2[MT], u3[MT];                                                     \
      _Pragma("unroll") for (int jt = 0; jt < MT; ++jt) { u1[jt] = (d4){0., 0., 0., 0.}; u2[jt] = u1[jt]; u3[jt] = u1[jt]; } \
      _Pragma("unroll") for (int ks = 0; ks < 4 * MT; ++ks) {                        \
        const double xs_ = xr[S_][ks] + xi[S_][ks];                                  \
        _Pragma("unroll") for (int jt = 0; jt < MT; ++jt) {                          \
          u1[jt] = mfma64(xr[S_][ks], qr[jt][ks], u1[jt]);                           \
          u2[jt] = mfma64(xi[S_][ks], qi[jt][ks], u2[jt]);                           \
          u3[jt] = mfma64(xs_, qs[jt][ks], u3[jt]);                                  \
        }                                                                            \
      }                                                                              \
      _Pragma("unroll") for (int jt = 0; jt < MT; ++jt)                              \
        _Pragma("unroll") for (int v = 0; v < 4; ++v) {                              \
          const long o = HPX_HJ_OFF(16 * jt + li, 16 * (t_) + HPX_ACC_ROW(g, v));    \
          wr[o] = u1[jt][v] - u2[jt][v];                                             \
          wi[o] = u3[jt][v] - u1[jt][v] - u2[jt][v];                                 \
        }                                                                            \
    }
    if (wave < ntile) { HPX_HJ_ULOAD(0, wave) }
    for (int t = wave; t < ntile; t += 8) {
      HPX_HJ_ULOAD(1, min(t + 4, ntile - 1))
      __builtin_amdgcn_sched_barrier(0);
      HPX_HJ_UTILE(0, t)
      __builtin_amdgcn_sched_barrier(0);
      HPX_HJ_ULOAD(0, min(t + 8, ntile - 1))
      __builtin_amdgcn_sched_barrier(0);
      if (t + 4 < ntile) { HPX_HJ_UTILE(1, t + 4) }
      __builtin_amdgcn_sched_barrier(0);
    }
#undef HPX_HJ_ULOAD
#undef HPX_HJ_UTILE
  }
#undef HPX_HJ_OFF
}

// lambda_j = |w_j|^2 - ridge on the diagonal of gr, V = the normalised columns ([row][column], pitch n)
__global__ __launch_bounds__(256) void k_hj_finish(const double* __restrict__ wre, const double* __restrict__ wim,
                                                   const double* __restrict__ ridge, double* __restrict__ gr,
                                                   double* __restrict__ vr, double* __restrict__ vi, const int n,
                                                   const int NB) {
  __shared__ double acc[32][8];
  __shared__ double inv[8];
  // workgroup = 8 consecutive columns (inside one block of NB = 8 or 16)
  const int b = blockIdx.y, blk = blockIdx.x, tid = threadIdx.x, cc = tid & 7, rr = tid >> 3;
  const int col0 = blk * 8;
  const double* wr = wre + (long)b * n * n + (long)(col0 / NB) * n * NB + (col0 % NB);
  const double* wi = wim + (long)b * n * n + (long)(col0 / NB) * n * NB + (col0 % NB);
  double sum = 0.0;
  for (int k = rr; k < n; k += 32) {
    const double xr = wr[(long)k * NB + cc], xi = wi[(long)k * NB + cc];
    sum += xr * xr + xi * xi;
  }
  acc[rr][cc] = sum;
  __syncthreads();
  if (tid < 8) {
    double t = 0.0;
    for (int k = 0; k < 32; ++k) t += acc[k][tid];
    const int j = blk * 8 + tid;
    gr[(long)b * n * n + (long)j * n + j] = t - ridge[b];
    inv[tid] = t > 0.0 ? 1.0 / sqrt(t) : 0.0;
  }
  __syncthreads();
  for (int k = rr; k < n; k += 32) {
    const long o = (long)b * n * n + (long)k * n + blk * 8 + cc;
    vr[o] = wr[(long)k * NB + cc] * inv[cc];
    vi[o] = wi[(long)k * NB + cc] * inv[cc];
  }
}

// interleaved (n0 x n0) -> planar inside an n x n slot (zero padding)
__global__ void k_hj_in(const double* __restrict__ a, double* __restrict__ gr, double* __restrict__ gi, const int n0,
                        const int n) {
  const int b = blockIdx.y;
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < (long)n * n; e += (long)gridDim.x * blockDim.x) {
    const int i = (int)(e / n), j = (int)(e % n);
    const bool in = i < n0 && j < n0;
    gr[(long)b * n * n + e] = in ? a[(((long)b * n0 + i) * n0 + j) * 2] : 0.0;
    gi[(long)b * n * n + e] = in ? a[(((long)b * n0 + i) * n0 + j) * 2 + 1] : 0.0;
  }
}
// eigenvalues (the diagonal of gr) and eigenvectors (planar, pitch n) -> w (n0), v (n0 x n0 interleaved): the columns
// of the padding coordinates are dropped by the caller's sort (their eigenvalues are zero), here all n0 leading ones
// are copied in the solver's order
__global__ void k_hj_out(const double* __restrict__ gr, const double* __restrict__ vr, const double* __restrict__ vi,
                         double* __restrict__ w, double* __restrict__ v, const int n0, const int n) {
  const int b = blockIdx.y;
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < (long)n * n; e += (long)gridDim.x * blockDim.x) {
    const int i = (int)(e / n), j = (int)(e % n);
    if (i == 0) w[(long)b * n + j] = gr[(long)b * n * n + (long)j * n + j];
    if (i < n0) {
      v[(((long)b * n0 + i) * n + j) * 2] = vr[(long)b * n * n + e];
      v[(((long)b * n0 + i) * n + j) * 2 + 1] = vi[(long)b * n * n + e];
    }
  }
}

}  // namespace

// the order a matrix of order n0 is padded to (zero rows / columns): a multiple of 16, of 32 from 256 on
int hpx_eigh_padded_order(int n0) { return n0 >= 241 ? ((n0 + 31) & ~31) : ((n0 + 15) & ~15); }

// Eigendecomposition of nb Hermitian positive semi-definite matrices given planar (gr, gi: [nb][n][n], n a multiple
// of 16): on return the diagonal of gr holds the eigenvalues (unsorted) and vr, vi the unit eigenvectors as columns.
// gi and the off-diagonal of gr are left as they were.  sweeps_out (host, optional): outer sweeps taken.
// HPX_OPT_EIGH_INNER_SWEEPS: sweeps of the small two-sided problem per visit (1; measured at order 512, 256 matrices:
// 0.39 s with one, 0.51 s with two, 0.56 s with three -- the outer sweep count does not change); HPX_OPT_EIGH_TRACE:
// the convergence measure of every sweep on stderr
static int g_eigh_inner = 1, g_eigh_trace = 0;
int hpx_eigh_set_option(int key, int value) {
  if (key == HPX_OPT_EIGH_INNER_SWEEPS && value >= 1 && value <= 8) g_eigh_inner = value;
  else if (key == HPX_OPT_EIGH_TRACE) g_eigh_trace = value != 0;
  else return HPX_EINVAL;
  return HPX_OK;
}

int hpx_eigh_psd_planar(int nb, int n, double* gr, const double* gi, double* vr, double* vi, int* sweeps_out,
                        hipStream_t st) {
  HPX_REQUIRE(nb > 0 && n >= 16 && (n & 15) == 0, "hpx_eigh_psd_planar: the order must be a multiple of 16");
  const size_t m = (size_t)n * n;
  hpx_devbuf abuf, lbuf, wbuf, sbuf;
  HPX_TRY(abuf.alloc(2 * nb * m));
  HPX_TRY(lbuf.alloc(2 * nb * m));
  HPX_TRY(wbuf.alloc(2 * nb * m));
  HPX_TRY(sbuf.alloc((size_t)3 * nb));
  double *wre = wbuf.p, *wim = wbuf.p + nb * m;
  double* ridge = sbuf.p;
  unsigned long long* meas = (unsigned long long*)(sbuf.p + nb);
  int32_t* info = (int32_t*)(sbuf.p + 2 * nb);
  hipLaunchKernelGGL(k_hj_prep, dim3(nb), dim3(256), 0, st, gr, gi, abuf.p, ridge, n);
  HPX_HIP(hipGetLastError());
  HPX_TRY(hpx_zpotrf_batched(nb, n, abuf.p, lbuf.p, info, (void*)st));
  {
    std::vector<int32_t> hinfo(nb);
    HPX_HIP(hipMemcpyAsync(hinfo.data(), info, (size_t)nb * sizeof(int32_t), hipMemcpyDeviceToHost, st));
    HPX_HIP(hipStreamSynchronize(st));
    for (int b = 0; b < nb; ++b)
      if (hinfo[b]) {
        hpx_set_error("hpx_eigh_psd_planar: matrix %d is not positive semi-definite", b);
        return HPX_EINVAL;
      }
  }
  // 16-column blocks when the order allows an even number of them (n a multiple of 32: hpx_eigh_padded_order)
  const int NB = (n >= 256 && (n & 31) == 0) ? 16 : 8;
  hipLaunchKernelGGL(k_hj_pack, dim3(64, nb), dim3(256), 0, st, lbuf.p, wre, wim, n, NB);
  HPX_HIP(hipGetLastError());
  const int p = n / NB;
  const bool trace = g_eigh_trace != 0;
  // sweeps of the 16 x 16 problem per visit: one (measured at order 512, 256 matrices: 0.39 s with one, 0.51 s
  // with two, 0.56 s with three -- the outer sweep count, 10 - 11, does not change)
  const int inner = g_eigh_inner;
  std::vector<double> hm(nb);
  int sweeps = 0;
  bool converged = false;
  double worst = 0.0;
  for (; sweeps < HPX_EIGH_MAX_SWEEPS; ++sweeps) {
    HPX_HIP(hipMemsetAsync(meas, 0, (size_t)nb * sizeof(double), st));
    for (int s = 0; s < p - 1; ++s)
      if (NB == 16) hipLaunchKernelGGL(k_hj_step<16>, dim3(p / 2, nb), dim3(256), 0, st, wre, wim, n, p, s, meas, inner);
      else hipLaunchKernelGGL(k_hj_step<8>, dim3(p / 2, nb), dim3(256), 0, st, wre, wim, n, p, s, meas, inner);
    HPX_HIP(hipGetLastError());
    HPX_HIP(hipMemcpyAsync(hm.data(), meas, (size_t)nb * sizeof(double), hipMemcpyDeviceToHost, st));
    HPX_HIP(hipStreamSynchronize(st));
    worst = 0.0;
    for (int b = 0; b < nb; ++b)
      if (!(hm[b] <= worst)) worst = hm[b];
    if (!std::isfinite(worst)) {
      if (sweeps_out) *sweeps_out = -1;
      hpx_set_error("hpx_eigh_psd_planar: non-finite Gram matrix at sweep %d (order %d): the input holds NaN / inf or "
                    "overflows", sweeps, n);
      return HPX_EINVAL;
    }
    if (trace) fprintf(stderr, "hpx_eigh: sweep %d  max |G_ij|^2 / (G_ii G_jj) before its rotations = %.3e\n", sweeps, worst);
    // the measure was taken BEFORE this sweep's rotations, and the convergence is quadratic by then (measured:
    // 3e-7 -> 8e-13 -> 1e-25): below 1e-10 the sweep just done leaves the columns orthogonal to rounding
    if (worst < 1e-10) { ++sweeps; converged = true; break; }
  }
  if (!converged) {
    // (the eigenpairs of the sweep limit are not handed out as if they were converged)
    if (sweeps_out) *sweeps_out = -1;
    hpx_set_error("hpx_eigh_psd_planar: no convergence in %d sweeps (order %d; max |G_ij|^2 / (G_ii G_jj) = %.3e before "
                  "the last one, needs < 1e-10)", HPX_EIGH_MAX_SWEEPS, n, worst);
    return HPX_EINVAL;
  }
  hipLaunchKernelGGL(k_hj_finish, dim3(n / 8, nb), dim3(256), 0, st, wre, wim, ridge, gr, vr, vi, n, NB);
  HPX_HIP(hipGetLastError());
  HPX_HIP(hipStreamSynchronize(st));
  if (sweeps_out) *sweeps_out = sweeps;
  return HPX_OK;
}

// C-ABI: a (nb,n0,n0) c128 Hermitian positive semi-definite -> w (nb,n) f64 and v (nb,n0,n) c128 with
// n = hpx_eigh_padded_order(n0) (hpx_zheev_psd_order): all n eigenpairs of the matrix padded with zeros, in the solver's order (the caller sorts; the
// n - n0 pairs of the padding have eigenvalue 0 and zero vectors in the first n0 coordinates).
extern "C" int hpx_zheev_psd_batched(int nb, int n0, const double* a, double* w, double* v, int* sweeps_out,
                                     void* stream) {
  HPX_REQUIRE(nb > 0 && n0 > 0 && n0 <= 2048 && a && w && v, "hpx_zheev_psd_batched: bad argument");
  hipStream_t st = (hipStream_t)stream;
  const int n = hpx_eigh_padded_order(n0);
  hpx_devbuf g;
  HPX_TRY(g.alloc((size_t)4 * nb * n * n));
  double *gr = g.p, *gi = gr + (size_t)nb * n * n, *vr = gi + (size_t)nb * n * n, *vi = vr + (size_t)nb * n * n;
  hipLaunchKernelGGL(k_hj_in, dim3(64, nb), dim3(256), 0, st, a, gr, gi, n0, n);
  HPX_HIP(hipGetLastError());
  HPX_TRY(hpx_eigh_psd_planar(nb, n, gr, gi, vr, vi, sweeps_out, st));
  hipLaunchKernelGGL(k_hj_out, dim3(64, nb), dim3(256), 0, st, gr, vr, vi, w, v, n0, n);
  HPX_HIP(hipGetLastError());
  HPX_HIP(hipStreamSynchronize(st));
  return HPX_OK;
}

extern "C" int hpx_zheev_psd_order(int n0) { return hpx_eigh_padded_order(n0); }
