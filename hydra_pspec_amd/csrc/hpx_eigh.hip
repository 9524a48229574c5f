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
// method is bandwidth-bound: 1024 matrices of order 512 take about a second.
#include "hpx_internal.h"
#include "../../include/hpx.h"

namespace {

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

// W = L (lower triangle of the interleaved factor, zero above) in block layout [n / 8][n rows][8]
__global__ void k_hj_pack(const double* __restrict__ l, double* __restrict__ wre, double* __restrict__ wim, const int n) {
  const int b = blockIdx.y;
  const double* lb = l + (long)b * n * n * 2;
  double* wr = wre + (long)b * n * n;
  double* wi = wim + (long)b * n * n;
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < (long)n * n; e += (long)gridDim.x * blockDim.x) {
    const int row = (int)(e / n), col = (int)(e % n);
    const long o = ((long)(col >> 3) * n + row) * 8 + (col & 7);
    const bool low = col <= row;
    wr[o] = low ? lb[e * 2] : 0.0;
    wi[o] = low ? lb[e * 2 + 1] : 0.0;
  }
}

// one step of the round-robin schedule: workgroup (pair, baseline)
__global__ __launch_bounds__(256, 2) void k_hj_step(double* __restrict__ wre, double* __restrict__ wim, const int n,
                                                    const int p, const int s, unsigned long long* __restrict__ meas,
                                                    const int inner_sweeps) {
  __shared__ double part[4][2][256];
  __shared__ double Ga[2][2][256], Qa[2][2][256];
  __shared__ double rot[8][4];
  __shared__ int partner[16], isq[16], pidx[16];
  __shared__ double red[4];
  __shared__ int skip;
  const int b = blockIdx.y, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, li = lane & 15, g = lane >> 4;
  int bi, bj;
  rr_pair16(p, s, blockIdx.x, bi, bj);
  double* wr = wre + (long)b * n * n;
  double* wi = wim + (long)b * n * n;
  const long offA = (long)bi * n * 8, offB = (long)bj * n * 8;
  const int ntile = n >> 4;                         // 16-row tiles, dealt to the waves round-robin
  // ---- 1. Gram matrix of the pair's 16 columns: G[i][j] = sum_k conj(W[k][i]) W[k][j]
  {
    const long cb = ((li < 8) ? offA : offB) + (li & 7);
    d4 grr = {0., 0., 0., 0.}, gii = grr, gri = grr, gir = grr;
    for (int t = wave; t < ntile; t += 4) {
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        const long o = cb + (long)(16 * t + 4 * ks + g) * 8;
        const double xr = wr[o], xi = wi[o];
        grr = mfma64(xr, xr, grr);
        gii = mfma64(xi, xi, gii);
        gri = mfma64(xr, xi, gri);
        gir = mfma64(xi, xr, gir);
      }
    }
#pragma unroll
    for (int v = 0; v < 4; ++v) {                   // accumulator: row g + 4 v, column li
      part[wave][0][HPX_ACC_ROW(g, v) * 16 + li] = grr[v] + gii[v];
      part[wave][1][HPX_ACC_ROW(g, v) * 16 + li] = gri[v] - gir[v];
    }
  }
  __syncthreads();
  const int r = tid >> 4, c = tid & 15;
  {
    const double sre = part[0][0][tid] + part[1][0][tid] + part[2][0][tid] + part[3][0][tid];
    const double sim = part[0][1][tid] + part[1][1][tid] + part[2][1][tid] + part[3][1][tid];
    Ga[0][0][tid] = sre;
    Ga[0][1][tid] = (r == c) ? 0.0 : sim;
    Qa[0][0][tid] = (r == c) ? 1.0 : 0.0;
    Qa[0][1][tid] = 0.0;
  }
  __syncthreads();
  {   // how far from orthogonal the 16 columns were: max |G_rc|^2 / (G_rr G_cc)
    double m = 0.0;
    if (r < c) {
      const double off = Ga[0][0][tid] * Ga[0][0][tid] + Ga[0][1][tid] * Ga[0][1][tid];
      const double dd = Ga[0][0][r * 17] * Ga[0][0][c * 17];
      m = (dd > 0.0) ? off / dd : (off > 0.0 ? 1.0 : 0.0);
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
  // ---- 2. two-sided Jacobi on G (one entry per thread), Q accumulates the rotations
  int cur = 0;
  for (int sw = 0; sw < inner_sweeps; ++sw)
    for (int st = 0; st < 15; ++st) {
      if (tid < 8) {
        int pp, qq;
        rr_pair16(16, st, tid, pp, qq);
        const double a = Ga[cur][0][pp * 17], bq = Ga[cur][0][qq * 17];
        const double cr = Ga[cur][0][pp * 16 + qq], ci = Ga[cur][1][pp * 16 + qq];
        const double ac2 = cr * cr + ci * ci;
        double cs = 1.0, sn = 0.0, cp = 1.0, sp = 0.0;
        if (ac2 > 1e-34 * fabs(a * bq) && ac2 > 0.0) {
          const double ac = sqrt(ac2);
          const double tau = (bq - a) / (2.0 * ac);
          const double t = ((tau >= 0.0) ? 1.0 : -1.0) / (fabs(tau) + sqrt(1.0 + tau * tau));
          cs = 1.0 / sqrt(1.0 + t * t);
          sn = t * cs;
          cp = cr / ac;
          sp = ci / ac;
        }
        rot[tid][0] = cs; rot[tid][1] = sn; rot[tid][2] = cp; rot[tid][3] = sp;
        partner[pp] = qq; partner[qq] = pp;
        isq[pp] = 0; isq[qq] = 1;
        pidx[pp] = tid; pidx[qq] = tid;
      }
      __syncthreads();
      // x_a' = own_a x_a + part_a x_partner(a):  a = p: (cs, -sn e^{-i phi});  a = q: (cs, sn e^{i phi})
      const int rp = partner[r], cq = partner[c];
      const double* rc_ = rot[pidx[c]];
      const double own_c = rc_[0];
      const double pcr = isq[c] ? rc_[1] * rc_[2] : -rc_[1] * rc_[2];
      const double pci = rc_[1] * rc_[3];
      const double* rr_ = rot[pidx[r]];
      const double own_r = rr_[0];
      const double prr = isq[r] ? rr_[1] * rr_[2] : -rr_[1] * rr_[2];
      const double pri = rr_[1] * rr_[3];
      const double* gre = Ga[cur][0];
      const double* gim = Ga[cur][1];
      // T = G J (columns), at rows r and partner(r)
      const double t1r = gre[r * 16 + c] * own_c + gre[r * 16 + cq] * pcr - gim[r * 16 + cq] * pci;
      const double t1i = gim[r * 16 + c] * own_c + gre[r * 16 + cq] * pci + gim[r * 16 + cq] * pcr;
      const double t2r = gre[rp * 16 + c] * own_c + gre[rp * 16 + cq] * pcr - gim[rp * 16 + cq] * pci;
      const double t2i = gim[rp * 16 + c] * own_c + gre[rp * 16 + cq] * pci + gim[rp * 16 + cq] * pcr;
      // G' = J^H T (rows): conj(part_r) = (prr, -pri)
      const double nr = own_r * t1r + prr * t2r + pri * t2i;
      const double ni = own_r * t1i + prr * t2i - pri * t2r;
      const double* qre = Qa[cur][0];
      const double* qim = Qa[cur][1];
      const double qr_ = qre[r * 16 + c] * own_c + qre[r * 16 + cq] * pcr - qim[r * 16 + cq] * pci;
      const double qi_ = qim[r * 16 + c] * own_c + qre[r * 16 + cq] * pci + qim[r * 16 + cq] * pcr;
      Ga[cur ^ 1][0][tid] = nr;
      Ga[cur ^ 1][1][tid] = (r == c) ? 0.0 : ni;
      Qa[cur ^ 1][0][tid] = qr_;
      Qa[cur ^ 1][1][tid] = qi_;
      __syncthreads();
      cur ^= 1;
    }
  // ---- 3. W <- W Q on the pair's columns, 16 rows at a time
  {
    double qr[4], qi[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {                // B[k = 4 ks + g][n = li] = Q[k][li]
      qr[ks] = Qa[cur][0][(4 * ks + g) * 16 + li];
      qi[ks] = Qa[cur][1][(4 * ks + g) * 16 + li];
    }
    const long sb = ((li < 8) ? offA : offB) + (li & 7);
    for (int t = wave; t < ntile; t += 4) {
      const int rt = 16 * t;
      d4 dre = {0., 0., 0., 0.}, dim = dre;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {              // A[m = li][k = 4 ks + g] = W[rt + li][column 4 ks + g of the pair]
        const int i = 4 * ks + g;
        const long o = ((i < 8) ? offA : offB) + (long)(rt + li) * 8 + (i & 7);
        const double xr = wr[o], xi = wi[o];
        dre = mfma64(xr, qr[ks], dre);
        dre = mfma64(-xi, qi[ks], dre);
        dim = mfma64(xr, qi[ks], dim);
        dim = mfma64(xi, qr[ks], dim);
      }
#pragma unroll
      for (int v = 0; v < 4; ++v) {                 // accumulator: row rt + g + 4 v, column li of the pair
        const long o = sb + (long)(rt + HPX_ACC_ROW(g, v)) * 8;
        wr[o] = dre[v];
        wi[o] = dim[v];
      }
    }
  }
}

// lambda_j = |w_j|^2 - ridge on the diagonal of gr, V = the normalised columns ([row][column], pitch n)
__global__ __launch_bounds__(256) void k_hj_finish(const double* __restrict__ wre, const double* __restrict__ wim,
                                                   const double* __restrict__ ridge, double* __restrict__ gr,
                                                   double* __restrict__ vr, double* __restrict__ vi, const int n) {
  __shared__ double acc[32][8];
  __shared__ double inv[8];
  const int b = blockIdx.y, blk = blockIdx.x, tid = threadIdx.x, cc = tid & 7, rr = tid >> 3;
  const double* wr = wre + (long)b * n * n + (long)blk * n * 8;
  const double* wi = wim + (long)b * n * n + (long)blk * n * 8;
  double sum = 0.0;
  for (int k = rr; k < n; k += 32) {
    const double xr = wr[(long)k * 8 + cc], xi = wi[(long)k * 8 + cc];
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
    vr[o] = wr[(long)k * 8 + cc] * inv[cc];
    vi[o] = wi[(long)k * 8 + cc] * inv[cc];
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

// Eigendecomposition of nb Hermitian positive semi-definite matrices given planar (gr, gi: [nb][n][n], n a multiple
// of 16): on return the diagonal of gr holds the eigenvalues (unsorted) and vr, vi the unit eigenvectors as columns.
// gi and the off-diagonal of gr are left as they were.  sweeps_out (host, optional): outer sweeps taken.
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
  hipLaunchKernelGGL(k_hj_pack, dim3(64, nb), dim3(256), 0, st, lbuf.p, wre, wim, n);
  HPX_HIP(hipGetLastError());
  const int p = n >> 3;
  static const bool trace = getenv("HPX_EIGH_TRACE") != nullptr;
  // sweeps of the 16 x 16 problem per visit: one (measured at order 512, 256 matrices: 0.39 s with one, 0.51 s
  // with two, 0.56 s with three -- the outer sweep count, 10 - 11, does not change)
  static const int inner = getenv("HPX_EIGH_INNER") ? atoi(getenv("HPX_EIGH_INNER")) : 1;
  std::vector<double> hm(nb);
  int sweeps = 0;
  for (; sweeps < 30; ++sweeps) {
    HPX_HIP(hipMemsetAsync(meas, 0, (size_t)nb * sizeof(double), st));
    for (int s = 0; s < p - 1; ++s)
      hipLaunchKernelGGL(k_hj_step, dim3(p / 2, nb), dim3(256), 0, st, wre, wim, n, p, s, meas, inner);
    HPX_HIP(hipGetLastError());
    HPX_HIP(hipMemcpyAsync(hm.data(), meas, (size_t)nb * sizeof(double), hipMemcpyDeviceToHost, st));
    HPX_HIP(hipStreamSynchronize(st));
    double worst = 0.0;
    for (int b = 0; b < nb; ++b) worst = hm[b] > worst ? hm[b] : worst;
    if (trace) fprintf(stderr, "hpx_eigh: sweep %d  max |G_ij|^2 / (G_ii G_jj) before its rotations = %.3e\n", sweeps, worst);
    // the measure was taken BEFORE this sweep's rotations, and the convergence is quadratic by then (measured:
    // 3e-8 -> 3e-16 -> 2e-30): below 1e-13 the sweep just done leaves the columns orthogonal to rounding
    if (worst < 1e-13) { ++sweeps; break; }
  }
  hipLaunchKernelGGL(k_hj_finish, dim3(p, nb), dim3(256), 0, st, wre, wim, ridge, gr, vr, vi, n);
  HPX_HIP(hipGetLastError());
  HPX_HIP(hipStreamSynchronize(st));
  if (sweeps_out) *sweeps_out = sweeps;
  return HPX_OK;
}

// C-ABI: a (nb,n0,n0) c128 Hermitian positive semi-definite -> w (nb,n) f64 and v (nb,n0,n) c128 with
// n = ceil16(n0): all n eigenpairs of the matrix padded with zeros, in the solver's order (the caller sorts; the
// n - n0 pairs of the padding have eigenvalue 0 and zero vectors in the first n0 coordinates).
extern "C" int hpx_zheev_psd_batched(int nb, int n0, const double* a, double* w, double* v, int* sweeps_out,
                                     void* stream) {
  HPX_REQUIRE(nb > 0 && n0 > 0 && n0 <= 2048 && a && w && v, "hpx_zheev_psd_batched: bad argument");
  hipStream_t st = (hipStream_t)stream;
  const int n = (n0 + 15) & ~15;
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
