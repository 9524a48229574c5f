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
// in[b][j][col]: col < nm: tw_b[j] * modes[col][j];  col == nm: tw_b[j] * d_b[j]
__global__ void k_dpss_in(const double* __restrict__ d, const double* __restrict__ tw,
                          const double* __restrict__ modes, double* __restrict__ ire,
                          double* __restrict__ iim, const int N, const int nm, const int NP,
                          const int ncol) {
  const int b = blockIdx.y;
  const long tot = (long)NP * ncol;
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < tot;
       e += (long)gridDim.x * blockDim.x) {
    const int j = (int)(e / ncol), col = (int)(e % ncol);
    double vr = 0.0, vi = 0.0;
    if (j < N) {
      const double w = tw[(long)b * N + j];
      if (col < nm) vr = w * modes[(long)col * N + j];
      else if (col == nm) {
        vr = w * d[((long)b * N + j) * 2];
        vi = w * d[((long)b * N + j) * 2 + 1];
      }
    }
    ire[(long)b * tot + e] = vr;
    iim[(long)b * tot + e] = vi;
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

// projection onto the weighted basis + nm x nm Hermitian solve; one block per spectrum
#define HPX_DPSS_MAXM 32
__global__ __launch_bounds__(256) void k_dpss_solve(const double* __restrict__ tw,
                                                    const double* __restrict__ modes,
                                                    const double* __restrict__ ore,
                                                    const double* __restrict__ oim,
                                                    double* __restrict__ amps, const int N,
                                                    const int nm, const int NP, const int ncol) {
  __shared__ double Are[HPX_DPSS_MAXM][HPX_DPSS_MAXM + 2], Aim[HPX_DPSS_MAXM][HPX_DPSS_MAXM + 2];
  const int b = blockIdx.x, tid = threadIdx.x;
  const double* pr = ore + (long)b * NP * ncol;
  const double* pi = oim + (long)b * NP * ncol;
  // [lhs | rhs][k][col] = sum_j tw[j] modes[k][j] * (Ah in)[j][col]
  for (int e = tid; e < nm * (nm + 1); e += 256) {
    const int k = e / (nm + 1), col = e % (nm + 1);
    double sr = 0.0, si = 0.0;
    for (int j = 0; j < N; ++j) {
      const double w = tw[(long)b * N + j] * modes[(long)k * N + j];
      sr += w * pr[(long)j * ncol + col];
      si += w * pi[(long)j * ncol + col];
    }
    Are[k][col] = sr;
    Aim[k][col] = si;
  }
  __syncthreads();
  // Gaussian elimination with partial pivoting on the augmented nm x (nm+1) system
  for (int k = 0; k < nm; ++k) {
    __shared__ int piv;
    if (tid == 0) {
      int best = k;
      double bv = Are[k][k] * Are[k][k] + Aim[k][k] * Aim[k][k];
      for (int r = k + 1; r < nm; ++r) {
        const double v = Are[r][k] * Are[r][k] + Aim[r][k] * Aim[r][k];
        if (v > bv) { bv = v; best = r; }
      }
      piv = best;
    }
    __syncthreads();
    if (piv != k) {
      for (int c = tid; c <= nm; c += 256) {
        const double tr = Are[k][c], ti = Aim[k][c];
        Are[k][c] = Are[piv][c]; Aim[k][c] = Aim[piv][c];
        Are[piv][c] = tr; Aim[piv][c] = ti;
      }
    }
    __syncthreads();
    const double pr0 = Are[k][k], pi0 = Aim[k][k];
    const double den = pr0 * pr0 + pi0 * pi0;
    for (int e = tid; e < (nm - k - 1) * (nm - k); e += 256) {
      const int r = k + 1 + e / (nm - k), c = k + 1 + e % (nm - k);
      // f = A[r][k] / A[k][k]
      const double fr = (Are[r][k] * pr0 + Aim[r][k] * pi0) / den;
      const double fi = (Aim[r][k] * pr0 - Are[r][k] * pi0) / den;
      Are[r][c] -= fr * Are[k][c] - fi * Aim[k][c];
      Aim[r][c] -= fr * Aim[k][c] + fi * Are[k][c];
    }
    __syncthreads();
  }
  if (tid == 0) {
    for (int k = nm - 1; k >= 0; --k) {
      double sr = Are[k][nm], si = Aim[k][nm];
      for (int c = k + 1; c < nm; ++c) {
        const double xr = amps[(long)b * 2 * nm + 2 * c], xi = amps[(long)b * 2 * nm + 2 * c + 1];
        sr -= Are[k][c] * xr - Aim[k][c] * xi;
        si -= Are[k][c] * xi + Aim[k][c] * xr;
      }
      const double pr0 = Are[k][k], pi0 = Aim[k][k], den = pr0 * pr0 + pi0 * pi0;
      amps[(long)b * 2 * nm + 2 * k] = (sr * pr0 + si * pi0) / den;
      amps[(long)b * 2 * nm + 2 * k + 1] = (si * pr0 - sr * pi0) / den;
    }
  }
}

// ---- OQE ------------------------------------------------------------------------
// m_a[j] = exp(-2 pi i a j / s)  (oqe.py:7-10)
__device__ __forceinline__ void twid(const int a, const int j, const int s, double& c, double& sn) {
  const long q = ((long)a * j) % s;
  sincospi(-2.0 * (double)q / (double)s, &sn, &c);
}

// T1[a][k] = sum_j mm[a][j] R[j][k] with mm = m (conjm=0) or conj(m) (conjm=1)
__global__ void k_oqe_left(const double* __restrict__ R, double* __restrict__ T1, const int s,
                           const int conjm) {
  const int b = blockIdx.y;
  const double* Rb = R + (long)b * s * s * 2;
  for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < s * s; e += gridDim.x * blockDim.x) {
    const int a = e / s, k = e % s;
    double sr = 0.0, si = 0.0;
    for (int j = 0; j < s; ++j) {
      double c, sn;
      twid(a, j, s, c, sn);
      if (conjm) sn = -sn;
      const double rr = Rb[((long)j * s + k) * 2], ri = Rb[((long)j * s + k) * 2 + 1];
      sr += c * rr - sn * ri;
      si += c * ri + sn * rr;
    }
    T1[((long)b * s * s + e) * 2] = sr;
    T1[((long)b * s * s + e) * 2 + 1] = si;
  }
}

// X[a][b2] = sum_k T1[a][k] mm[b2][k], mm = conj(m) (conjm=1) or m (conjm=0)
__global__ void k_oqe_right(const double* __restrict__ T1, double* __restrict__ X, const int s,
                            const int conjm) {
  const int b = blockIdx.y;
  const double* Tb = T1 + (long)b * s * s * 2;
  for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < s * s; e += gridDim.x * blockDim.x) {
    const int a = e / s, b2 = e % s;
    double sr = 0.0, si = 0.0;
    for (int k = 0; k < s; ++k) {
      double c, sn;
      twid(b2, k, s, c, sn);
      if (conjm) sn = -sn;
      const double tr = Tb[((long)a * s + k) * 2], ti = Tb[((long)a * s + k) * 2 + 1];
      sr += tr * c - ti * sn;
      si += tr * sn + ti * c;
    }
    X[((long)b * s * s + e) * 2] = sr;
    X[((long)b * s * s + e) * 2 + 1] = si;
  }
}

// variant 0: F_ab = 1/2 conj(Wm[b][a]) X[a][b];  variant 1: Ft_ab = 1/2 |X[a][b]|^2
__global__ void k_oqe_combine(const double* __restrict__ X, const double* __restrict__ Wm,
                              double* __restrict__ F, const int s, const int variant) {
  const int b = blockIdx.y;
  for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < s * s; e += gridDim.x * blockDim.x) {
    const int a = e / s, b2 = e % s;
    const long o = ((long)b * s * s + e) * 2;
    const double xr = X[o], xi = X[o + 1];
    double fr, fi;
    if (variant == 1) {
      fr = 0.5 * (xr * xr + xi * xi);
      fi = 0.0;
    } else {
      const long ow = ((long)b * s * s + (long)b2 * s + a) * 2;
      const double wr = Wm[ow], wi = -Wm[ow + 1];
      fr = 0.5 * (wr * xr - wi * xi);
      fi = 0.5 * (wr * xi + wi * xr);
    }
    F[o] = fr;
    F[o + 1] = fi;
  }
}

// y[b][v][j] = sum_k R[b][j][k] V[b][v][k]
__global__ void k_oqe_rx(const double* __restrict__ R, const double* __restrict__ V,
                         double* __restrict__ Y, const int s, const int nv) {
  const int b = blockIdx.y;
  const double* Rb = R + (long)b * s * s * 2;
  for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < nv * s; e += gridDim.x * blockDim.x) {
    const int v = e / s, j = e % s;
    const double* x = V + ((long)b * nv + v) * s * 2;
    double sr = 0.0, si = 0.0;
    for (int k = 0; k < s; ++k) {
      const double rr = Rb[((long)j * s + k) * 2], ri = Rb[((long)j * s + k) * 2 + 1];
      sr += rr * x[2 * k] - ri * x[2 * k + 1];
      si += rr * x[2 * k + 1] + ri * x[2 * k];
    }
    Y[((long)b * nv * s + e) * 2] = sr;
    Y[((long)b * nv * s + e) * 2 + 1] = si;
  }
}

// y[b][2v][j] = sum_k R[b][k][j] V[b][v][k] (R^T x),  y[b][2v+1][j] = sum_k R[b][j][k] V[b][v][k] (R x)
__global__ void k_oqe_rx2(const double* __restrict__ R, const double* __restrict__ V,
                          double* __restrict__ Y, const int s, const int nv) {
  const int b = blockIdx.y;
  const double* Rb = R + (long)b * s * s * 2;
  for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < nv * s; e += gridDim.x * blockDim.x) {
    const int v = e / s, j = e % s;
    const double* x = V + ((long)b * nv + v) * s * 2;
    double tr = 0.0, ti = 0.0, sr = 0.0, si = 0.0;
    for (int k = 0; k < s; ++k) {
      const double xr = x[2 * k], xi = x[2 * k + 1];
      const double ar = Rb[((long)k * s + j) * 2], ai = Rb[((long)k * s + j) * 2 + 1];
      const double rr = Rb[((long)j * s + k) * 2], ri = Rb[((long)j * s + k) * 2 + 1];
      tr += ar * xr - ai * xi;
      ti += ar * xi + ai * xr;
      sr += rr * xr - ri * xi;
      si += rr * xi + ri * xr;
    }
    double* y = Y + (((long)b * 2 * nv + 2 * v) * s + j) * 2;
    y[0] = tr;
    y[1] = ti;
    y[(long)s * 2] = sr;
    y[(long)s * 2 + 1] = si;
  }
}

// q[b][p][t] = 1/2 conj(FFT(y1))[t] FFT(y2)[t]
__global__ void k_oqe_q(const double* __restrict__ Y, double* __restrict__ q, const int s,
                        const int npair) {
  const int b = blockIdx.y;
  for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < npair * s; e += gridDim.x * blockDim.x) {
    const int pidx = e / s, t = e % s;
    const double* y1 = Y + ((long)b * 2 * npair + 2 * pidx) * s * 2;
    const double* y2 = y1 + (long)s * 2;
    double ar = 0.0, ai = 0.0, br = 0.0, bi = 0.0;
    for (int j = 0; j < s; ++j) {
      double c, sn;
      twid(t, j, s, c, sn);
      ar += c * y1[2 * j] - sn * y1[2 * j + 1];
      ai += c * y1[2 * j + 1] + sn * y1[2 * j];
      br += c * y2[2 * j] - sn * y2[2 * j + 1];
      bi += c * y2[2 * j + 1] + sn * y2[2 * j];
    }
    q[((long)b * npair * s + e) * 2] = 0.5 * (ar * br + ai * bi);
    q[((long)b * npair * s + e) * 2 + 1] = 0.5 * (ar * bi - ai * br);
  }
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

extern "C" int hpx_dpss_project(int nb, int N, int nm, const double* d, const double* tw,
                                const double* modes, const double* icov, double* amps,
                                void* stream) {
  HPX_REQUIRE(nb > 0 && N > 0 && nm > 0 && nm < HPX_DPSS_MAXM && d && tw && modes && icov && amps,
              "hpx_dpss_project: bad argument (need 0 < nmodes < 32)");
  hipStream_t st = (hipStream_t)stream;
  const int NP = ceil16(N), ncol = ceil16(nm + 1);
  const size_t wsz = (size_t)NP * NP, bsz = (size_t)nb * NP * ncol;
  hpx_devbuf wbuf, dbuf;
  HPX_TRY(wbuf.alloc(2 * wsz));
  HPX_TRY(dbuf.alloc(4 * bsz));
  double *wre = wbuf.p, *wim = wre + wsz, *buf = dbuf.p;
  double *ire = buf, *iim = buf + bsz, *ore = buf + 2 * bsz, *oim = buf + 3 * bsz;
  hipLaunchKernelGGL(k_herm_planar, dim3(256), dim3(256), 0, st, icov, wre, wim, N, NP);
  hipLaunchKernelGGL(k_dpss_in, dim3(32, nb), dim3(256), 0, st, d, tw, modes, ire, iim, N, nm, NP,
                     ncol);
  // out = Ah * in: the stored planar matrix is Ah^T = conj(Ah), hence conjW = 1
  int rc = hpx_launch_dft(nb, NP, ncol, wre, wim, 1, ire, iim, (long)NP * ncol, ncol, nullptr, 0,
                          ore, oim, (long)NP * ncol, ncol, 1.0, st, 0);
  if (rc == HPX_OK) {
    hipLaunchKernelGGL(k_dpss_solve, dim3(nb), dim3(256), 0, st, tw, modes, ore, oim, amps, N, nm,
                       NP, ncol);
    if (hipGetLastError() != hipSuccess) rc = HPX_EHIP;
  }
  hipError_t e = hipStreamSynchronize(st);
  if (e != hipSuccess) { hpx_set_error("hpx_dpss_project: %s", hipGetErrorString(e)); return HPX_EHIP; }
  return rc;
}

extern "C" int hpx_oqe_fisher(int nb, int s, const double* R, double* F_out, int variant,
                              void* stream) {
  HPX_REQUIRE(nb > 0 && s > 0 && R && F_out && (variant == 0 || variant == 1),
              "hpx_oqe_fisher: bad argument");
  hipStream_t st = (hipStream_t)stream;
  const size_t sz = (size_t)nb * s * s * 2;
  hpx_devbuf dbuf;
  HPX_TRY(dbuf.alloc(3 * sz));
  double *T1 = dbuf.p, *X = T1 + sz, *Wm = T1 + 2 * sz;
  dim3 grid((s * s + 255) / 256, nb);
  // X = M R M^H
  hipLaunchKernelGGL(k_oqe_left, grid, dim3(256), 0, st, R, T1, s, 0);
  hipLaunchKernelGGL(k_oqe_right, grid, dim3(256), 0, st, T1, X, s, 1);
  if (variant == 0) {   // Wm = conj(M) R M^T
    hipLaunchKernelGGL(k_oqe_left, grid, dim3(256), 0, st, R, T1, s, 1);
    hipLaunchKernelGGL(k_oqe_right, grid, dim3(256), 0, st, T1, Wm, s, 0);
  }
  hipLaunchKernelGGL(k_oqe_combine, grid, dim3(256), 0, st, X, Wm, F_out, s, variant);
  hipError_t e = hipGetLastError();
  if (e == hipSuccess) e = hipStreamSynchronize(st);
  if (e != hipSuccess) { hpx_set_error("hpx_oqe_fisher: %s", hipGetErrorString(e)); return HPX_EHIP; }
  return HPX_OK;
}

extern "C" int hpx_oqe_qh(int nb, int npair, int s, const double* R, const double* V,
                          double* q_out, void* stream) {
  HPX_REQUIRE(nb > 0 && npair > 0 && s > 0 && R && V && q_out, "hpx_oqe_qh: bad argument");
  hipStream_t st = (hipStream_t)stream;
  hpx_devbuf ybuf;
  HPX_TRY(ybuf.alloc((size_t)nb * 2 * npair * s * 2));
  double* Y = ybuf.p;
  hipLaunchKernelGGL(k_oqe_rx, dim3((2 * npair * s + 255) / 256, nb), dim3(256), 0, st, R, V, Y, s,
                     2 * npair);
  hipLaunchKernelGGL(k_oqe_q, dim3((npair * s + 255) / 256, nb), dim3(256), 0, st, Y, q_out, s,
                     npair);
  hipError_t e = hipGetLastError();
  if (e == hipSuccess) e = hipStreamSynchronize(st);
  if (e != hipSuccess) { hpx_set_error("hpx_oqe_qh: %s", hipGetErrorString(e)); return HPX_EHIP; }
  return HPX_OK;
}

extern "C" int hpx_oqe_qauto(int nb, int nvis, int s, const double* R, const double* V,
                             double* q_out, void* stream) {
  HPX_REQUIRE(nb > 0 && nvis > 0 && s > 0 && R && V && q_out, "hpx_oqe_qauto: bad argument");
  hipStream_t st = (hipStream_t)stream;
  hpx_devbuf ybuf;
  HPX_TRY(ybuf.alloc((size_t)nb * 2 * nvis * s * 2));
  hipLaunchKernelGGL(k_oqe_rx2, dim3((nvis * s + 255) / 256, nb), dim3(256), 0, st, R, V, ybuf.p, s, nvis);
  hipLaunchKernelGGL(k_oqe_q, dim3((nvis * s + 255) / 256, nb), dim3(256), 0, st, ybuf.p, q_out, s, nvis);
  hipError_t e = hipGetLastError();
  if (e == hipSuccess) e = hipStreamSynchronize(st);
  if (e != hipSuccess) { hpx_set_error("hpx_oqe_qauto: %s", hipGetErrorString(e)); return HPX_EHIP; }
  return HPX_OK;
}
