// Principal square root and inverse square root of a batch of Hermitian positive-definite matrices on the device:
// the coupled Newton-Schulz iteration
//     Y_0 = A / s,  Z_0 = I,      R_k = (3 I - Z_k Y_k) / 2,   Y_{k+1} = Y_k R_k,   Z_{k+1} = R_k Z_k
// (s = the largest absolute row sum of A >= its largest eigenvalue, so the spectrum of Y_0 lies in (0, 1] and the
// iteration converges, quadratically once Z_k Y_k is near I):  Y_k -> (A / s)^1/2,  Z_k -> (A / s)^-1/2.
// All iterates are polynomials in A -- Hermitian, commuting -- so the three products per step are the batched
// complex GEMMs the chain already has (hpx_launch_dft with one W per system, FP64 MFMA).  24 n^3 flops per step and
// system against an eigendecomposition's latency-bound sweeps: 1024 matrices of order 512 take about a second.
//
// Replaces, at set-up of the correlated-noise paths, the host calls scipy.linalg.sqrtm / numpy.linalg.eigh per
// baseline (reference pspec.py:361-362): for a Hermitian Ninv the root itself; with flagged channels
// Ninv diag(w) = P [[A, 0], [B, 0]] P^T (unflagged channels first, A = Ninv[u, u] Hermitian) has the root
// P [[A^1/2, 0], [B A^-1/2, 0]] P^T -- one call on A gives both factors (hydra_pspec_amd/pspec.py).
#include "hpx_internal.h"

namespace {

// s[b] = max_i sum_j |a_ij|   (one workgroup per system)
__global__ __launch_bounds__(256) void k_ns_norm(const double* __restrict__ a, double* __restrict__ s, const int n) {
  __shared__ double red[256];
  const int b = blockIdx.x, tid = threadIdx.x;
  const double* ab = a + (long)b * n * n * 2;
  double best = 0.0;
  for (int i = tid; i < n; i += 256) {
    double acc = 0.0;
    for (int j = 0; j < n; ++j) {
      const double re = ab[((long)i * n + j) * 2], im = ab[((long)i * n + j) * 2 + 1];
      acc += sqrt(re * re + im * im);
    }
    best = fmax(best, acc);
  }
  red[tid] = best;
  __syncthreads();
  for (int k = 128; k > 0; k >>= 1) {
    if (tid < k) red[tid] = fmax(red[tid], red[tid + k]);
    __syncthreads();
  }
  if (tid == 0) s[b] = red[0];
}

// Y = A / s (Hermitian part), Z = I, planar
__global__ void k_ns_init(const double* __restrict__ a, const double* __restrict__ s, double* __restrict__ yr,
                          double* __restrict__ yi, double* __restrict__ zr, double* __restrict__ zi, const int n) {
  const int b = blockIdx.y;
  const double inv = 1.0 / s[b];
  const double* ab = a + (long)b * n * n * 2;
  const long o = (long)b * n * n;
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < (long)n * n; e += (long)gridDim.x * blockDim.x) {
    const int i = (int)(e / n), j = (int)(e % n);
    const long t = (long)j * n + i;
    yr[o + e] = 0.5 * (ab[e * 2] + ab[t * 2]) * inv;
    yi[o + e] = 0.5 * (ab[e * 2 + 1] - ab[t * 2 + 1]) * inv;
    zr[o + e] = (i == j) ? 1.0 : 0.0;
    zi[o + e] = 0.0;
  }
}

// T <- (3 I - T) / 2 in place;  err[b] += || I - T ||_F^2
__global__ __launch_bounds__(256) void k_ns_resid(double* __restrict__ tr, double* __restrict__ ti,
                                                  double* __restrict__ err, const int n) {
  __shared__ double red[256];
  const int b = blockIdx.y, tid = threadIdx.x;
  const long o = (long)b * n * n;
  double acc = 0.0;
  for (long e = (long)blockIdx.x * 256 + tid; e < (long)n * n; e += (long)gridDim.x * 256) {
    const int i = (int)(e / n), j = (int)(e % n);
    const double dr = ((i == j) ? 1.0 : 0.0) - tr[o + e], di = -ti[o + e];
    acc += dr * dr + di * di;
    tr[o + e] = 0.5 * (((i == j) ? 3.0 : 0.0) - tr[o + e]);
    ti[o + e] = -0.5 * ti[o + e];
  }
  red[tid] = acc;
  __syncthreads();
  for (int k = 128; k > 0; k >>= 1) {
    if (tid < k) red[tid] += red[tid + k];
    __syncthreads();
  }
  if (tid == 0) atomicAdd(&err[b], red[0]);
}

// out (interleaved) = scale[b]^(+-1/2) * planar
__global__ void k_ns_out(const double* __restrict__ pr, const double* __restrict__ pi, const double* __restrict__ s,
                         const int inverse, double* __restrict__ out, const int n) {
  const int b = blockIdx.y;
  const double f = inverse ? 1.0 / sqrt(s[b]) : sqrt(s[b]);
  const long o = (long)b * n * n;
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < (long)n * n; e += (long)gridDim.x * blockDim.x) {
    out[(o + e) * 2] = f * pr[o + e];
    out[(o + e) * 2 + 1] = f * pi[o + e];
  }
}

// ---- the masked root's bookkeeping (hpx_sqrtm_masked_batched) ----------------------------------------------------
// per matrix: the permutation "unflagged channels first" (stable), the number of unflagged channels, the mean of the
// unflagged diagonal (the value of the padding's diagonal: it keeps the padded block's condition number that of A)
__global__ __launch_bounds__(256) void k_sm_order(const uint8_t* __restrict__ w, const double* __restrict__ a,
                                                  const long a_bstride, int32_t* __restrict__ perm,
                                                  int32_t* __restrict__ nu, double* __restrict__ dmean, const int n) {
  const int b = blockIdx.x;
  __shared__ int cnt;
  __shared__ double dsum;
  if (threadIdx.x == 0) {
    const uint8_t* wb = w + (long)b * n;
    const double* ab = a + (long)b * a_bstride;
    int32_t* pb = perm + (long)b * n;
    int k = 0;
    double d = 0.0;
    for (int i = 0; i < n; ++i)
      if (wb[i]) { pb[k++] = i; d += ab[((long)i * n + i) * 2]; }
    cnt = k;
    dsum = d;
    for (int i = 0; i < n; ++i)
      if (!wb[i]) pb[k++] = i;
    nu[b] = cnt;
    dmean[b] = cnt > 0 ? dsum / cnt : 1.0;
  }
}
// A[b] (npad x npad, interleaved) = Ninv[u, u] in the permuted order, the padding's diagonal = the mean of A's
__global__ void k_sm_block(const double* __restrict__ a, const long a_bstride, const int32_t* __restrict__ perm,
                           const int32_t* __restrict__ nu, const double* __restrict__ dmean, double* __restrict__ out,
                           const int n, const int npad) {
  const int b = blockIdx.y;
  const double* ab = a + (long)b * a_bstride;
  const int32_t* pb = perm + (long)b * n;
  const int m = nu[b];
  double* ob = out + (long)b * npad * npad * 2;
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < (long)npad * npad; e += (long)gridDim.x * blockDim.x) {
    const int i = (int)(e / npad), j = (int)(e % npad);
    double re = 0.0, im = 0.0;
    if (i < m && j < m) {
      const long o = ((long)pb[i] * n + pb[j]) * 2;
      re = ab[o];
      im = ab[o + 1];
    } else if (i == j) {
      re = dmean[b];
    }
    ob[e * 2] = re;
    ob[e * 2 + 1] = im;
  }
}
// planar in[b][k][c] (npad x ncol) = B^T: B[c][k] = Ninv[perm c][perm k] for a flagged row c >= nu and an unflagged
// column k < nu, zero elsewhere
__global__ void k_sm_bt(const double* __restrict__ a, const long a_bstride, const int32_t* __restrict__ perm,
                        const int32_t* __restrict__ nu, double* __restrict__ inr, double* __restrict__ ini, const int n,
                        const int npad, const int ncol) {
  const int b = blockIdx.y;
  const double* ab = a + (long)b * a_bstride;
  const int32_t* pb = perm + (long)b * n;
  const int m = nu[b];
  const long o = (long)b * npad * ncol;
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < (long)npad * ncol; e += (long)gridDim.x * blockDim.x) {
    const int k = (int)(e / ncol), c = (int)(e % ncol);
    double re = 0.0, im = 0.0;
    if (k < m && c >= m && c < n) {
      const long q = ((long)pb[c] * n + pb[k]) * 2;
      re = ab[q];
      im = ab[q + 1];
    }
    inr[o + e] = re;
    ini[o + e] = im;
  }
}
// out[b][perm r][perm j] (n x n, interleaved): sqrt(s) Y[r][j] for r, j < nu; (B A^-1/2)[r][j] = lowT[j][r] / sqrt(s)
// for r >= nu, j < nu; zero in the flagged columns
__global__ void k_sm_scatter(const double* __restrict__ yr, const double* __restrict__ yi, const double* __restrict__ lr,
                             const double* __restrict__ li, const double* __restrict__ s,
                             const int32_t* __restrict__ perm, const int32_t* __restrict__ nu, double* __restrict__ out,
                             const int n, const int npad, const int ncol) {
  const int b = blockIdx.y;
  const int32_t* pb = perm + (long)b * n;
  const int m = nu[b];
  const double f = sqrt(s[b]), fi = 1.0 / f;
  const long oy = (long)b * npad * npad, ol = (long)b * npad * ncol;
  double* ob = out + (long)b * n * n * 2;
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < (long)n * n; e += (long)gridDim.x * blockDim.x) {
    const int r = (int)(e / n), j = (int)(e % n);
    double re = 0.0, im = 0.0;
    if (j < m) {
      if (r < m) {
        re = f * yr[oy + (long)r * npad + j];
        im = f * yi[oy + (long)r * npad + j];
      } else {
        re = fi * lr[ol + (long)j * ncol + r];
        im = fi * li[ol + (long)j * ncol + r];
      }
    }
    const long q = ((long)pb[r] * n + pb[j]) * 2;
    ob[q] = re;
    ob[q + 1] = im;
  }
}

// The iteration on one chunk of nc systems (interleaved input `ab`): on return (yr, yi) = (A / s)^1/2 and (zr, zi) =
// (A / s)^-1/2 in planar form (the buffers are swapped among the five pairs of `W`), s[b] the scales.
struct NsWork {
  double *yr, *yi, *zr, *zi, *tr, *ti, *y2r, *y2i, *z2r, *z2i, *s, *err;
};
int ns_iterate(const int nc, const int n, const double* ab, NsWork& K, const double tol, const int max_iter,
               std::vector<double>& herr, int* iters, hipStream_t st) {
  const long m = (long)n * n;
  hipLaunchKernelGGL(k_ns_norm, dim3(nc), dim3(256), 0, st, ab, K.s, n);
  hipLaunchKernelGGL(k_ns_init, dim3(64, nc), dim3(256), 0, st, ab, K.s, K.yr, K.yi, K.zr, K.zi, n);
  HPX_HIP(hipGetLastError());
  int it = 0, polish = 1;      // one more step after the stop test is met (it squares the residual once more)
  bool done = false;
  while (it < max_iter + 1) {
    // T = Z Y  (the kernel forms W^H in with W = the stored matrix: the iterates are Hermitian to rounding)
    HPX_TRY(hpx_launch_dft(nc, n, n, K.zr, K.zi, 1, K.yr, K.yi, m, n, nullptr, 0, K.tr, K.ti, m, n, 1.0, st, 0, m));
    HPX_HIP(hipMemsetAsync(K.err, 0, (size_t)nc * sizeof(double), st));
    hipLaunchKernelGGL(k_ns_resid, dim3(64, nc), dim3(256), 0, st, K.tr, K.ti, K.err, n);       // T <- R = (3 I - T) / 2
    HPX_HIP(hipGetLastError());
    // the stable coupling: Y' = Y R (R on the right), Z' = R Z
    HPX_TRY(hpx_launch_dft(nc, n, n, K.yr, K.yi, 1, K.tr, K.ti, m, n, nullptr, 0, K.y2r, K.y2i, m, n, 1.0, st, 0, m));
    HPX_TRY(hpx_launch_dft(nc, n, n, K.tr, K.ti, 1, K.zr, K.zi, m, n, nullptr, 0, K.z2r, K.z2i, m, n, 1.0, st, 0, m));
    std::swap(K.yr, K.y2r); std::swap(K.yi, K.y2i); std::swap(K.zr, K.z2r); std::swap(K.zi, K.z2i);
    ++it;
    if (done) break;           // that was the polishing step
    // the residual measured at the START of this step: the step just taken squares it
    HPX_HIP(hipMemcpyAsync(herr.data(), K.err, (size_t)nc * sizeof(double), hipMemcpyDeviceToHost, st));
    HPX_HIP(hipStreamSynchronize(st));
    double worst = 0.0;
    for (int b = 0; b < nc; ++b)
      if (!(herr[b] <= worst)) worst = herr[b];       // (written so that a NaN residual is kept, not dropped)
    if (!std::isfinite(worst)) {
      hpx_set_error("hpx_sqrtm_hpd_batched: the iteration diverged (a matrix is not positive definite?)");
      return HPX_EINVAL;
    }
    done = sqrt(worst) < tol * n;       // || I - Z Y ||_F below tol * n before the step: below (tol n)^2 after it
    if (done && !polish) break;
    if (!done && it >= max_iter) {
      hpx_set_error("hpx_sqrtm_hpd_batched: no convergence in %d iterations", max_iter);
      return HPX_EINVAL;
    }
  }
  *iters = it;
  return HPX_OK;
}
int ns_alloc(hpx_devbuf& work, hpx_devbuf& sc, const int chunk, const long m, NsWork& K) {
  HPX_TRY(work.alloc((size_t)chunk * m * 10));      // 5 planar work matrices per system: 80 n^2 bytes each
  HPX_TRY(sc.alloc((size_t)chunk * 2));
  double* W = work.p;
  K.yr = W; K.yi = W + chunk * m; K.zr = W + 2 * chunk * m; K.zi = W + 3 * chunk * m; K.tr = W + 4 * chunk * m;
  K.ti = W + 5 * chunk * m; K.y2r = W + 6 * chunk * m; K.y2i = W + 7 * chunk * m; K.z2r = W + 8 * chunk * m;
  K.z2i = W + 9 * chunk * m;
  K.s = sc.p;
  K.err = sc.p + chunk;
  return HPX_OK;
}

}  // namespace

extern "C" int hpx_sqrtm_hpd_batched(int nb, int n, const double* a, double* sq, double* isq, double tol,
                                     int max_iter, int* iters_out, void* stream) {
  HPX_REQUIRE(nb > 0 && n > 0 && (n & 15) == 0 && a && (sq || isq), "hpx_sqrtm_hpd_batched: need nb > 0, n a multiple of 16, "
              "the matrices and at least one output");
  HPX_REQUIRE(tol > 0 && max_iter > 0, "hpx_sqrtm_hpd_batched: need tol > 0 and max_iter > 0");
  hipStream_t st = (hipStream_t)stream;
  const long m = (long)n * n;
  const int chunk = nb < 256 ? nb : 256;
  hpx_devbuf work, sc;
  NsWork K;
  HPX_TRY(ns_alloc(work, sc, chunk, m, K));
  std::vector<double> herr(chunk);
  int worst_iters = 0;
  for (int b0 = 0; b0 < nb; b0 += chunk) {
    const int nc = (nb - b0 < chunk) ? nb - b0 : chunk;
    int it = 0;
    HPX_TRY(ns_iterate(nc, n, a + (long)b0 * m * 2, K, tol, max_iter, herr, &it, st));
    worst_iters = it > worst_iters ? it : worst_iters;
    if (sq) hipLaunchKernelGGL(k_ns_out, dim3(64, nc), dim3(256), 0, st, K.yr, K.yi, K.s, 0, sq + (long)b0 * m * 2, n);
    if (isq) hipLaunchKernelGGL(k_ns_out, dim3(64, nc), dim3(256), 0, st, K.zr, K.zi, K.s, 1, isq + (long)b0 * m * 2, n);
    HPX_HIP(hipGetLastError());
    HPX_HIP(hipStreamSynchronize(st));
  }
  if (iters_out) *iters_out = worst_iters;
  return HPX_OK;
}

// sqrtm(Ninv diag(w)) for nb Hermitian positive-definite Ninv (a: (nb | 1, n, n) c128; shared != 0: one matrix for
// all) and channel masks w (nb, n) u8 (1 = use): with the unflagged channels u first,
// Ninv diag(w) = P [[A, 0], [B, 0]] P^T  ->  P [[A^1/2, 0], [B A^-1/2, 0]] P^T.  Everything on the device, in chunks
// of at most 256 systems: permutation, the A blocks (padded to a common multiple of 16 with mean(diag A) on the
// diagonal), Newton-Schulz, the product B A^-1/2 on the batched FP64-MFMA GEMM, the scatter back to channel order.
extern "C" int hpx_sqrtm_masked_batched(int nb, int n, const double* a, int shared, const uint8_t* w, double* out,
                                        double tol, int max_iter, int* iters_out, void* stream) {
  HPX_REQUIRE(nb > 0 && n > 0 && a && w && out, "hpx_sqrtm_masked_batched: bad argument");
  HPX_REQUIRE(tol > 0 && max_iter > 0, "hpx_sqrtm_masked_batched: need tol > 0 and max_iter > 0");
  hipStream_t st = (hipStream_t)stream;
  const long a_bs = shared ? 0 : (long)n * n * 2;
  const int chunk = nb < 256 ? nb : 256;
  hpx_devbuf ibuf, dbuf;
  HPX_TRY(ibuf.alloc(((size_t)chunk * (n + 1) + 1) / 2 + 1));        // int32 perm [chunk][n] + nu [chunk]
  HPX_TRY(dbuf.alloc((size_t)chunk));
  int32_t* perm = (int32_t*)ibuf.p;
  int32_t* nu = perm + (size_t)chunk * n;
  std::vector<int32_t> hnu(chunk);
  int worst_iters = 0;
  for (int b0 = 0; b0 < nb; b0 += chunk) {
    const int nc = (nb - b0 < chunk) ? nb - b0 : chunk;
    const double* ab = a + (long)b0 * a_bs;
    hipLaunchKernelGGL(k_sm_order, dim3(nc), dim3(256), 0, st, w + (long)b0 * n, ab, a_bs, perm, nu, dbuf.p, n);
    HPX_HIP(hipGetLastError());
    HPX_HIP(hipMemcpyAsync(hnu.data(), nu, (size_t)nc * sizeof(int32_t), hipMemcpyDeviceToHost, st));
    HPX_HIP(hipStreamSynchronize(st));
    int mx = 0, mn = n;
    for (int b = 0; b < nc; ++b) { mx = hnu[b] > mx ? hnu[b] : mx; mn = hnu[b] < mn ? hnu[b] : mn; }
    double* ob = out + (long)b0 * n * n * 2;
    if (mx == 0) {
      HPX_HIP(hipMemsetAsync(ob, 0, (size_t)nc * n * n * 2 * sizeof(double), st));
      continue;
    }
    const int npad = ceil16(mx), ncol = ceil16(n);
    const long m = (long)npad * npad;
    hpx_devbuf work, sc, abuf, lbuf;
    NsWork K;
    HPX_TRY(ns_alloc(work, sc, nc, m, K));
    HPX_TRY(abuf.alloc((size_t)nc * m * 2));
    hipLaunchKernelGGL(k_sm_block, dim3(64, nc), dim3(256), 0, st, ab, a_bs, perm, nu, dbuf.p, abuf.p, n, npad);
    HPX_HIP(hipGetLastError());
    std::vector<double> herr(nc);
    int it = 0;
    HPX_TRY(ns_iterate(nc, npad, abuf.p, K, tol, max_iter, herr, &it, st));
    worst_iters = it > worst_iters ? it : worst_iters;
    double *lr = nullptr, *li = nullptr;
    if (mn < n) {
      // rows of the flagged channels: (B A^-1/2)^T = (A^-1/2)^T B^T  (the kernel's W^T in form, conjW = 0)
      HPX_TRY(lbuf.alloc((size_t)nc * npad * ncol * 4));
      double *inr = lbuf.p, *ini = inr + (size_t)nc * npad * ncol;
      lr = ini + (size_t)nc * npad * ncol;
      li = lr + (size_t)nc * npad * ncol;
      hipLaunchKernelGGL(k_sm_bt, dim3(64, nc), dim3(256), 0, st, ab, a_bs, perm, nu, inr, ini, n, npad, ncol);
      HPX_HIP(hipGetLastError());
      HPX_TRY(hpx_launch_dft(nc, npad, ncol, K.zr, K.zi, 0, inr, ini, (long)npad * ncol, ncol, nullptr, 0, lr, li,
                             (long)npad * ncol, ncol, 1.0, st, 0, m));
    } else {
      HPX_TRY(lbuf.alloc(2));          // (never read: no flagged row)
      lr = lbuf.p;
      li = lbuf.p;
    }
    hipLaunchKernelGGL(k_sm_scatter, dim3(64, nc), dim3(256), 0, st, K.yr, K.yi, lr, li, K.s, perm, nu, ob, n, npad, ncol);
    HPX_HIP(hipGetLastError());
    HPX_HIP(hipStreamSynchronize(st));
  }
  if (iters_out) *iters_out = worst_iters;
  return HPX_OK;
}
