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

}  // namespace

extern "C" int hpx_sqrtm_hpd_batched(int nb, int n, const double* a, double* sq, double* isq, double tol,
                                     int max_iter, int* iters_out, void* stream) {
  HPX_REQUIRE(nb > 0 && n > 0 && (n & 15) == 0 && a && (sq || isq), "hpx_sqrtm_hpd_batched: need nb > 0, n a multiple of 16, "
              "the matrices and at least one output");
  HPX_REQUIRE(tol > 0 && max_iter > 0, "hpx_sqrtm_hpd_batched: need tol > 0 and max_iter > 0");
  hipStream_t st = (hipStream_t)stream;
  const long m = (long)n * n;
  const int chunk = nb < 256 ? nb : 256;       // 5 planar work matrices per system: 80 n^2 bytes each
  hpx_devbuf work, sc;
  HPX_TRY(work.alloc((size_t)chunk * m * 10));
  HPX_TRY(sc.alloc((size_t)chunk * 2));
  double* W = work.p;
  double *yr = W, *yi = W + chunk * m, *zr = W + 2 * chunk * m, *zi = W + 3 * chunk * m, *tr = W + 4 * chunk * m,
         *ti = W + 5 * chunk * m, *y2r = W + 6 * chunk * m, *y2i = W + 7 * chunk * m, *z2r = W + 8 * chunk * m,
         *z2i = W + 9 * chunk * m;
  double* s = sc.p;
  double* err = s + chunk;
  std::vector<double> herr(chunk);
  int worst_iters = 0;
  for (int b0 = 0; b0 < nb; b0 += chunk) {
    const int nc = (nb - b0 < chunk) ? nb - b0 : chunk;
    const double* ab = a + (long)b0 * m * 2;
    hipLaunchKernelGGL(k_ns_norm, dim3(nc), dim3(256), 0, st, ab, s, n);
    hipLaunchKernelGGL(k_ns_init, dim3(64, nc), dim3(256), 0, st, ab, s, yr, yi, zr, zi, n);
    HPX_HIP(hipGetLastError());
    int it = 0;
    bool done = false;
    while (!done && it < max_iter) {
      // T = Z Y  (the kernel forms W^H in with W = the stored matrix: Z is Hermitian)
      HPX_TRY(hpx_launch_dft(nc, n, n, zr, zi, 1, yr, yi, m, n, nullptr, 0, tr, ti, m, n, 1.0, st, 0, m));
      HPX_HIP(hipMemsetAsync(err, 0, (size_t)nc * sizeof(double), st));
      hipLaunchKernelGGL(k_ns_resid, dim3(64, nc), dim3(256), 0, st, tr, ti, err, n);       // T <- R = (3 I - T) / 2
      HPX_HIP(hipGetLastError());
      HPX_TRY(hpx_launch_dft(nc, n, n, tr, ti, 1, yr, yi, m, n, nullptr, 0, y2r, y2i, m, n, 1.0, st, 0, m));   // Y' = R Y
      HPX_TRY(hpx_launch_dft(nc, n, n, tr, ti, 1, zr, zi, m, n, nullptr, 0, z2r, z2i, m, n, 1.0, st, 0, m));   // Z' = R Z
      std::swap(yr, y2r); std::swap(yi, y2i); std::swap(zr, z2r); std::swap(zi, z2i);
      ++it;
      // the residual measured at the START of this step: the step just taken squares it
      HPX_HIP(hipMemcpyAsync(herr.data(), err, (size_t)nc * sizeof(double), hipMemcpyDeviceToHost, st));
      HPX_HIP(hipStreamSynchronize(st));
      double worst = 0.0;
      for (int b = 0; b < nc; ++b) worst = herr[b] > worst ? herr[b] : worst;
      if (!(worst == worst)) {
        hpx_set_error("hpx_sqrtm_hpd_batched: the iteration diverged (a matrix is not positive definite?)");
        return HPX_EINVAL;
      }
      done = sqrt(worst) < tol * n;       // || I - Z Y ||_F below tol * n before the step: below (tol n)^2 after it
    }
    if (!done) {
      hpx_set_error("hpx_sqrtm_hpd_batched: no convergence in %d iterations", max_iter);
      return HPX_EINVAL;
    }
    worst_iters = it > worst_iters ? it : worst_iters;
    if (sq) hipLaunchKernelGGL(k_ns_out, dim3(64, nc), dim3(256), 0, st, yr, yi, s, 0, sq + (long)b0 * m * 2, n);
    if (isq) hipLaunchKernelGGL(k_ns_out, dim3(64, nc), dim3(256), 0, st, zr, zi, s, 1, isq + (long)b0 * m * 2, n);
    HPX_HIP(hipGetLastError());
    HPX_HIP(hipStreamSynchronize(st));
  }
  if (iters_out) *iters_out = worst_iters;
  return HPX_OK;
}
