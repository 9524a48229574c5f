// Foreground modes from the data: leading eigenvectors of the per-baseline frequency-frequency
// covariance over time, cov = np.cov(bl_data.T)  (reference scripts/calc-vis-cov-matrices.py:
// 235-249; the driver then keeps the first Nfgmodes columns, run-hydra-pspec.py:453).
//
// The covariance C = A A^H, A = Xc^T / sqrt(T-1) (Xc = data minus its time mean), has rank at
// most T-1, so for T <= N its eigenvectors come from the T x T Gram matrix A^H A:
// (lambda, v) -> u = A v / sqrt(lambda).  Either way a Hermitian n x n problem with
// n = min(T, N) <= 1024 is diagonalised per baseline: below order 128 (the usual case: the Gram path with
// n = Ntimes) by the cyclic two-sided Jacobi method of this file with the round-robin parallel ordering -- n/2
// disjoint rotations per step, applied as a column phase and a row phase by the whole workgroup, one workgroup
// per baseline, the matrices in LDS up to n = 64; from order 128 on by the blocked one-sided Jacobi of
// hpx_eigh.hip (16 x 16 rotations on the MFMA, a workgroup per pair of column blocks: round 4).
#include "hpx_internal.h"

#ifndef HPX_EIGH_BLOCKED_MIN
#define HPX_EIGH_BLOCKED_MIN 128
#endif

namespace {

// Xc[b][t][k] = vis - mean over t, interleaved complex in, planar out
__global__ void k_center(const double* __restrict__ vis, double* __restrict__ xr,
                         double* __restrict__ xi, const int T, const int N) {
  const int b = blockIdx.y;
  for (int k = blockIdx.x * blockDim.x + threadIdx.x; k < N; k += gridDim.x * blockDim.x) {
    double mr = 0.0, mi = 0.0;
    for (int t = 0; t < T; ++t) {
      mr += vis[(((long)b * T + t) * N + k) * 2];
      mi += vis[(((long)b * T + t) * N + k) * 2 + 1];
    }
    mr /= T;
    mi /= T;
    for (int t = 0; t < T; ++t) {
      xr[((long)b * T + t) * N + k] = vis[(((long)b * T + t) * N + k) * 2] - mr;
      xi[((long)b * T + t) * N + k] = vis[(((long)b * T + t) * N + k) * 2 + 1] - mi;
    }
  }
}

// gram != 0:  G[t][t'] = sum_k conj(xc_t(k)) xc_t'(k) / (T-1)      (n = T)
// gram == 0:  G[i][j]  = sum_t xc_t(i) conj(xc_t(j)) / (T-1)       (n = N, the covariance itself)
// One workgroup per 32 x 32 tile of G: the two 32-row operand tiles go through LDS in chunks of 32
// along the reduction index (coalesced along the contiguous index of the cube either way), every
// thread accumulates four entries.
__global__ __launch_bounds__(256) void k_gram(const double* __restrict__ xr, const double* __restrict__ xi,
                                              double* __restrict__ gr, double* __restrict__ gi, const int T,
                                              const int N, const int n, const int pitch, const int gram) {
  __shared__ double Ar[32][33], Ai[32][33], Br[32][33], Bi[32][33];     // [row of G][reduction index]
  const int b = blockIdx.z, i0 = blockIdx.y * 32, j0 = blockIdx.x * 32, tid = threadIdx.x;
  const double* ar = xr + (long)b * T * N;
  const double* ai = xi + (long)b * T * N;
  const int nred = gram ? N : T;
  const int tj = tid & 31, ti = tid >> 5;            // entries (i0 + ti + 8 m, j0 + tj), m = 0..3
  double sr[4] = {0., 0., 0., 0.}, si[4] = {0., 0., 0., 0.};
  for (int k0 = 0; k0 < nred; k0 += 32) {
    // gram: element (row r, reduction k) is X[r][k] (contiguous in k); otherwise X[k][r] (contiguous in r)
    for (int e = tid; e < 32 * 32; e += 256) {
      const int fast = e & 31, slow = e >> 5;
      const int r = gram ? slow : fast, k = gram ? fast : slow;
      const bool okk = k0 + k < nred;
      const long oa = gram ? (long)(i0 + r) * N + k0 + k : (long)(k0 + k) * N + i0 + r;
      const long ob = gram ? (long)(j0 + r) * N + k0 + k : (long)(k0 + k) * N + j0 + r;
      const bool oka = okk && i0 + r < n, okb = okk && j0 + r < n;
      Ar[r][k] = oka ? ar[oa] : 0.0;
      Ai[r][k] = oka ? ai[oa] : 0.0;
      Br[r][k] = okb ? ar[ob] : 0.0;
      Bi[r][k] = okb ? ai[ob] : 0.0;
    }
    __syncthreads();
#pragma unroll 8
    for (int k = 0; k < 32; ++k) {
      const double qr = Br[tj][k], qi = Bi[tj][k];
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        const double pr = Ar[ti + 8 * m][k], pi = Ai[ti + 8 * m][k];
        if (gram) {                                   // conj(p) q
          sr[m] += pr * qr + pi * qi;
          si[m] += pr * qi - pi * qr;
        } else {                                      // p conj(q)
          sr[m] += pr * qr + pi * qi;
          si[m] += pi * qr - pr * qi;
        }
      }
    }
    __syncthreads();
  }
  const double sc = 1.0 / (double)(T - 1);
#pragma unroll
  for (int m = 0; m < 4; ++m) {
    const int i = i0 + ti + 8 * m, j = j0 + tj;
    if (i < n && j < n) {                             // order n inside a pitch x pitch slot
      gr[(long)b * pitch * pitch + (long)i * pitch + j] = sr[m] * sc;
      gi[(long)b * pitch * pitch + (long)i * pitch + j] = si[m] * sc;
    }
  }
}

// round-robin pairing of step s: pair i of n/2 (n even) -> (p, q), p < q
__device__ __forceinline__ void rr_pair(const int n, const int s, const int i, int& p, int& q) {
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

// Cyclic two-sided Jacobi: G -> diag(lambda), V -> eigenvectors (columns).  n even (the caller
// pads an odd order with an isolated zero row/column).
__global__ __launch_bounds__(256) void k_jacobi(double* __restrict__ gr_all, double* __restrict__ gi_all,
                                                double* __restrict__ vr_all, double* __restrict__ vi_all,
                                                const int n, const int max_sweeps, const int in_lds) {
  extern __shared__ double rot[];             // [n/2][4]: cs, sn, cos(phi), sin(phi); in_lds: + G and V (4 n^2)
  __shared__ double red[4], red2[4], floor_s;
  __shared__ int done;
  const int b = blockIdx.x, tid = threadIdx.x, half = n >> 1;
  double* const ggr = gr_all + (long)b * n * n;
  double* const ggi = gi_all + (long)b * n * n;
  double* const gvr = vr_all + (long)b * n * n;
  double* const gvi = vi_all + (long)b * n * n;
  // small orders (the usual Gram path, n = Ntimes): the matrices live in LDS for the whole diagonalisation --
  // every rotation step is three dependent read-modify-write passes over them, a cache round trip each otherwise
  // (leading dimension n + 1 there: the column phase walks a column with one thread per row, which at a
  // power-of-two stride is a 32-way bank conflict)
  double *gr = ggr, *gi = ggi, *vr = gvr, *vi = gvi;
  const int ld = in_lds ? n + 1 : n;
  if (in_lds) {
    gr = rot + 4 * half;
    gi = gr + n * ld;
    vr = gi + n * ld;
    vi = vr + n * ld;
    for (int e = tid; e < n * n; e += 256) {
      gr[(e / n) * ld + e % n] = ggr[e];
      gi[(e / n) * ld + e % n] = ggi[e];
    }
  }
  for (int e = tid; e < n * n; e += 256) {
    vr[(e / n) * ld + e % n] = (e / n == e % n) ? 1.0 : 0.0;
    vi[(e / n) * ld + e % n] = 0.0;
  }
  __syncthreads();
  for (int sweep = 0; sweep < max_sweeps; ++sweep) {
    // converged when every off-diagonal entry is negligible against its two diagonal entries,
    // |g_ij|^2 <= 1e-28 g_ii g_jj (eigenvalues exact to ~1e-14 relative, the level rounding allows), or against
    // the matrix itself, |g_ij| <= 1e-16 max_k g_kk: a centred cube has an exact null vector (rank Ntimes - 1),
    // whose diagonal entry is pure rounding -- the relative test alone is never met there and every run would
    // take all max_sweeps
    double gmax = 0.0;
    for (int i = tid; i < n; i += 256) gmax = fmax(gmax, fabs(gr[(long)i * ld + i]));
    for (int o = 32; o > 0; o >>= 1) gmax = fmax(gmax, __shfl_xor(gmax, o, 64));
    if ((tid & 63) == 0) red2[tid >> 6] = gmax;
    __syncthreads();
    gmax = fmax(fmax(red2[0], red2[1]), fmax(red2[2], red2[3]));
    const double floor2 = 1e-32 * gmax * gmax;
    double worst = 0.0;
    for (int e = tid; e < n * n; e += 256) {
      const int i = e / n, j = e % n;
      if (i >= j) continue;
      const double off = gr[i * ld + j] * gr[i * ld + j] + gi[i * ld + j] * gi[i * ld + j];
      const double dd = fabs(gr[(long)i * ld + i] * gr[(long)j * ld + j]);
      if (off > floor2) worst = fmax(worst, (dd > 0.0) ? off / dd : 1.0);
    }
    // block max through LDS
    for (int o = 32; o > 0; o >>= 1) worst = fmax(worst, __shfl_xor(worst, o, 64));
    if ((tid & 63) == 0) red[tid >> 6] = worst;
    __syncthreads();
    if (tid == 0) {
      done = fmax(fmax(red[0], red[1]), fmax(red[2], red[3])) < 1e-28;
      floor_s = floor2;
    }
    __syncthreads();
    if (done) break;
    for (int s = 0; s < n - 1; ++s) {
      for (int i = tid; i < half; i += 256) {      // rotation parameters of the step's pairs
        int p, q;
        rr_pair(n, s, i, p, q);
        const double a = gr[(long)p * ld + p], bq = gr[(long)q * ld + q];
        const double cr = gr[(long)p * ld + q], ci = gi[(long)p * ld + q];
        const double ac = sqrt(cr * cr + ci * ci);
        double cs = 1.0, sn = 0.0, cp = 1.0, sp = 0.0;
        if (ac * ac > floor_s && ac * ac > 1e-34 * fabs(a * bq)) {
          const double tau = (bq - a) / (2.0 * ac);
          const double t = ((tau >= 0.0) ? 1.0 : -1.0) / (fabs(tau) + sqrt(1.0 + tau * tau));
          cs = 1.0 / sqrt(1.0 + t * t);
          sn = t * cs;
          cp = cr / ac;
          sp = ci / ac;
        }
        rot[4 * i] = cs; rot[4 * i + 1] = sn; rot[4 * i + 2] = cp; rot[4 * i + 3] = sp;
      }
      __syncthreads();
      // columns: [x_p, x_q] <- [cs x_p - sn e^{-i phi} x_q, sn e^{i phi} x_p + cs x_q]  (G and V)
      for (int e = tid; e < half * n; e += 256) {
        const int i = e / n, k = e % n;
        int p, q;
        rr_pair(n, s, i, p, q);
        const double cs = rot[4 * i], sn = rot[4 * i + 1], cp = rot[4 * i + 2], sp = rot[4 * i + 3];
#pragma unroll
        for (int which = 0; which < 2; ++which) {
          double* mr = which ? vr : gr;
          double* mi = which ? vi : gi;
          const double pr = mr[(long)k * ld + p], pi = mi[(long)k * ld + p];
          const double qr = mr[(long)k * ld + q], qi = mi[(long)k * ld + q];
          // e^{-i phi} x_q = (cp - i sp)(qr + i qi);  e^{i phi} x_p = (cp + i sp)(pr + i pi)
          const double eqr = cp * qr + sp * qi, eqi = cp * qi - sp * qr;
          const double epr = cp * pr - sp * pi, epi = cp * pi + sp * pr;
          mr[(long)k * ld + p] = cs * pr - sn * eqr;
          mi[(long)k * ld + p] = cs * pi - sn * eqi;
          mr[(long)k * ld + q] = sn * epr + cs * qr;
          mi[(long)k * ld + q] = sn * epi + cs * qi;
        }
      }
      __syncthreads();
      // rows of G: [y_p, y_q] <- [cs y_p - sn e^{i phi} y_q, sn e^{-i phi} y_p + cs y_q]
      for (int e = tid; e < half * n; e += 256) {
        const int i = e / n, k = e % n;
        int p, q;
        rr_pair(n, s, i, p, q);
        const double cs = rot[4 * i], sn = rot[4 * i + 1], cp = rot[4 * i + 2], sp = rot[4 * i + 3];
        const double pr = gr[(long)p * ld + k], pi = gi[(long)p * ld + k];
        const double qr = gr[(long)q * ld + k], qi = gi[(long)q * ld + k];
        const double eqr = cp * qr - sp * qi, eqi = cp * qi + sp * qr;      // e^{i phi} y_q
        const double epr = cp * pr + sp * pi, epi = cp * pi - sp * pr;      // e^{-i phi} y_p
        gr[(long)p * ld + k] = cs * pr - sn * eqr;
        gi[(long)p * ld + k] = cs * pi - sn * eqi;
        gr[(long)q * ld + k] = sn * epr + cs * qr;
        gi[(long)q * ld + k] = sn * epi + cs * qi;
      }
      __syncthreads();
    }
  }
  if (in_lds) {
    __syncthreads();
    for (int e = tid; e < n * n; e += 256) {
      const int o = (e / n) * ld + e % n;
      ggr[e] = gr[o];
      ggi[e] = gi[o];
      gvr[e] = vr[o];
      gvi[e] = vi[o];
    }
  }
}

// Pick the nm largest eigenvalues (descending), build the modes and fix their phase (largest
// component real and positive).  gram: u[k] = sum_t xc_t(k) v[t] / sqrt(lambda (T-1)).
__global__ __launch_bounds__(256) void k_modes_out(const double* __restrict__ gr,
                                                   const double* __restrict__ vr_all,
                                                   const double* __restrict__ vi_all,
                                                   const double* __restrict__ xr,
                                                   const double* __restrict__ xi,
                                                   double* __restrict__ modes,
                                                   double* __restrict__ evals, const int T, const int N,
                                                   const int n, const int nreal, const int nm,
                                                   const int gram) {
  extern __shared__ double sh[];               // lam[n], then u_re[N], u_im[N]
  __shared__ int order[256];                   // (nm <= 256)
  __shared__ double red[4];
  __shared__ int redi[4];
  const int b = blockIdx.x, tid = threadIdx.x;
  double* lam = sh;
  double* ur = sh + n;
  double* ui = ur + N;
  for (int i = tid; i < n; i += 256) lam[i] = (i < nreal) ? gr[(long)b * n * n + (long)i * n + i] : -INFINITY;
  __syncthreads();
  if (tid == 0) {                              // selection of the nm largest (nm n compares: trivial)
    for (int m = 0; m < nm; ++m) {
      int best = 0;
      for (int i = 1; i < n; ++i)
        if (lam[i] > lam[best]) best = i;
      order[m] = best;
      evals[(long)b * nm + m] = lam[best];
      lam[best] = -INFINITY;
    }
  }
  __syncthreads();
  const double* vr = vr_all + (long)b * n * n;
  const double* vi = vi_all + (long)b * n * n;
  const double* ar = xr + (long)b * T * N;
  const double* ai = xi + (long)b * T * N;
  for (int m = 0; m < nm; ++m) {
    const int col = order[m];
    const double lm = evals[(long)b * nm + m];
    for (int k = tid; k < N; k += 256) {
      double sr = 0.0, si = 0.0;
      if (gram) {
        for (int t = 0; t < T; ++t) {
          const double pr = ar[(long)t * N + k], pi = ai[(long)t * N + k];
          const double qr = vr[(long)t * n + col], qi = vi[(long)t * n + col];
          sr += pr * qr - pi * qi;
          si += pr * qi + pi * qr;
        }
      } else {
        sr = vr[(long)k * n + col];
        si = vi[(long)k * n + col];
      }
      ur[k] = sr;
      ui[k] = si;
    }
    __syncthreads();
    // norm and the component of largest magnitude (first one on ties)
    double nn = 0.0, big = -1.0;
    int at = 0;
    for (int k = tid; k < N; k += 256) {
      const double a2 = ur[k] * ur[k] + ui[k] * ui[k];
      nn += a2;
      if (a2 > big) { big = a2; at = k; }
    }
    for (int o = 32; o > 0; o >>= 1) {
      nn += __shfl_xor(nn, o, 64);
      const double ob = __shfl_xor(big, o, 64);
      const int oa = __shfl_xor(at, o, 64);
      if (ob > big || (ob == big && oa < at)) { big = ob; at = oa; }
    }
    if ((tid & 63) == 0) { red[tid >> 6] = nn; redi[tid >> 6] = at; }
    __syncthreads();
    nn = red[0] + red[1] + red[2] + red[3];
    int best = redi[0];
    for (int w = 1; w < 4; ++w) {
      const int c = redi[w];
      const double a2 = ur[c] * ur[c] + ui[c] * ui[c], b2 = ur[best] * ur[best] + ui[best] * ui[best];
      if (a2 > b2 || (a2 == b2 && c < best)) best = c;
    }
    const double mag = sqrt(ur[best] * ur[best] + ui[best] * ui[best]);
    const double scale = (nn > 0.0 && mag > 0.0) ? 1.0 / sqrt(nn) : 0.0;
    const double pr = (mag > 0.0) ? ur[best] / mag : 1.0, pi = (mag > 0.0) ? -ui[best] / mag : 0.0;   // e^{-i arg}
    (void)lm;
    for (int k = tid; k < N; k += 256) {
      double* o = modes + (((long)b * N + k) * nm + m) * 2;
      o[0] = (ur[k] * pr - ui[k] * pi) * scale;
      o[1] = (ur[k] * pi + ui[k] * pr) * scale;
    }
    __syncthreads();
  }
}

}  // namespace

extern "C" int hpx_fgmodes_eig(int nb, int T, int N, int nmodes, const double* vis, double* modes,
                               double* evals, void* stream) {
  HPX_REQUIRE(nb > 0 && T > 1 && N > 0 && vis && modes && evals, "hpx_fgmodes_eig: bad argument");
  const int gram = (T <= N) ? 1 : 0;
  const int nreal = gram ? T : N;
  HPX_REQUIRE(nreal <= 1024, "hpx_fgmodes_eig: min(Ntimes, Nfreqs) must be <= 1024");
  HPX_REQUIRE(nmodes > 0 && nmodes <= nreal && nmodes <= 256, "hpx_fgmodes_eig: bad number of modes");
  // from order 128 on: the blocked one-sided Jacobi of hpx_eigh.hip (order padded to a multiple of 16); below, the
  // cyclic two-sided Jacobi of this file (even order for its round-robin pairing)
  const bool blocked = nreal >= HPX_EIGH_BLOCKED_MIN;
  const int n = blocked ? hpx_eigh_padded_order(nreal) : nreal + (nreal & 1);
  hipStream_t st = (hipStream_t)stream;
  hpx_devbuf xbuf, gbuf;
  HPX_TRY(xbuf.alloc((size_t)2 * nb * T * N));
  HPX_TRY(gbuf.alloc((size_t)4 * nb * n * n));
  double *xr = xbuf.p, *xi = xr + (size_t)nb * T * N;
  double *gr = gbuf.p, *gi = gr + (size_t)nb * n * n, *vr = gi + (size_t)nb * n * n, *vi = vr + (size_t)nb * n * n;
  HPX_HIP(hipMemsetAsync(gbuf.p, 0, (size_t)4 * nb * n * n * sizeof(double), st));
  hipLaunchKernelGGL(k_center, dim3((N + 255) / 256, nb), dim3(256), 0, st, vis, xr, xi, T, N);
  // an odd order is padded with an isolated zero row / column (never rotated, never selected)
  hipLaunchKernelGGL(k_gram, dim3((nreal + 31) / 32, (nreal + 31) / 32, nb), dim3(256), 0, st, xr, xi, gr, gi, T, N,
                     nreal, n, gram);
  HPX_HIP(hipGetLastError());
  if (blocked) {
    HPX_TRY(hpx_eigh_psd_planar(nb, n, gr, gi, vr, vi, nullptr, st));
  } else {
    const int in_lds = n <= 64;             // 4 n^2 doubles: 32 KB at n = 32, 128 KB at n = 64
    const size_t lds = ((size_t)(n / 2) * 4 + (in_lds ? (size_t)4 * n * (n + 1) : 0)) * sizeof(double);
    static hpx_lds_limit limit;
    if (lds > 48 * 1024) HPX_TRY(limit.ensure(reinterpret_cast<const void*>(&k_jacobi), lds));
    hipLaunchKernelGGL(k_jacobi, dim3(nb), dim3(256), lds, st, gr, gi, vr, vi, n, 30, in_lds);
  }
  hipLaunchKernelGGL(k_modes_out, dim3(nb), dim3(256), (size_t)(n + 2 * N) * sizeof(double), st, gr, vr, vi, xr, xi,
                     modes, evals, T, N, n, nreal, nmodes, gram);
  hipError_t e = hipGetLastError();
  if (e == hipSuccess) e = hipStreamSynchronize(st);
  if (e != hipSuccess) { hpx_set_error("hpx_fgmodes_eig: %s", hipGetErrorString(e)); return HPX_EHIP; }
  return HPX_OK;
}
