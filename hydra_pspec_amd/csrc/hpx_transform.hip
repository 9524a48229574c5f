// Batched centred DFT along the channel axis as a dense contraction with the
// (symmetric) Fourier operator on FP64 MFMA.  General N (zero padded to a
// multiple of 16).  out[b][x][c] = scale * sum_k W[x][k] * rs[b][k] * in[b][k][c].
#include "hpx_internal.h"

namespace {

// One wave: one 16-row x-tile, CTN 16-column tiles.  Grid (x-tile groups of 4, nbl).
template <int CTN>
__device__ __forceinline__ void dft_tile(const double* __restrict__ Wre,
                                         const double* __restrict__ Wim, const double wsign,
                                         const double* __restrict__ inre,
                                         const double* __restrict__ inim, const int in_ld,
                                         const double* __restrict__ rs, const int rs_n,
                                         double* __restrict__ outre, double* __restrict__ outim,
                                         const int out_ld, const int NP, const int x0,
                                         const int c0, const double scale, const int lane) {
  const int li = lane & 15, g = lane >> 4;
  d4 ar[CTN], ai[CTN];
#pragma unroll
  for (int c = 0; c < CTN; ++c) {
    ar[c] = (d4){0., 0., 0., 0.};
    ai[c] = (d4){0., 0., 0., 0.};
  }
  const int nks = NP >> 2;
#pragma unroll 2
  for (int ks = 0; ks < nks; ++ks) {
    const int k = 4 * ks + g;
    const double wr = Wre[(long)k * NP + x0 + li];
    const double wi = wsign * Wim[(long)k * NP + x0 + li];
    const double sc = rs ? (k < rs_n ? rs[k] : 0.0) : 1.0;
#pragma unroll
    for (int c = 0; c < CTN; ++c) {
      const long o = (long)k * in_ld + c0 + 16 * c + li;
      const double br = inre[o] * sc, bi = inim[o] * sc;
      ar[c] = mfma64(wr, br, ar[c]);
      ar[c] = mfma64(-wi, bi, ar[c]);
      ai[c] = mfma64(wr, bi, ai[c]);
      ai[c] = mfma64(wi, br, ai[c]);
    }
  }
#pragma unroll
  for (int c = 0; c < CTN; ++c)
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      const long o = (long)(x0 + HPX_ACC_ROW(g, v)) * out_ld + c0 + 16 * c + li;
      outre[o] = ar[c][v] * scale;
      outim[o] = ai[c][v] * scale;
    }
}

__global__ __launch_bounds__(256, 2) void k_dft(const double* __restrict__ Wre,
                                             const double* __restrict__ Wim, const int conjW,
                                             const double* __restrict__ inre,
                                             const double* __restrict__ inim,
                                             const long in_bstride, const int in_ld,
                                             const double* __restrict__ rs, const int rs_n,
                                             double* __restrict__ outre,
                                             double* __restrict__ outim, const long out_bstride,
                                             const int out_ld, const int NP, const int ncol,
                                             const double scale) {
  const int b = blockIdx.y;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int xt = blockIdx.x * 4 + wave;
  if (xt * 16 >= NP) return;
  const double wsign = conjW ? -1.0 : 1.0;
  const double* ir = inre + (long)b * in_bstride;
  const double* ii = inim + (long)b * in_bstride;
  double* orr = outre + (long)b * out_bstride;
  double* oi = outim + (long)b * out_bstride;
  const double* rsb = rs ? rs + (long)b * rs_n : nullptr;
  int c0 = 0;
  for (; c0 + 32 <= ncol; c0 += 32)
    dft_tile<2>(Wre, Wim, wsign, ir, ii, in_ld, rsb, rs_n, orr, oi, out_ld, NP, xt * 16, c0, scale,
                lane);
  if (c0 < ncol)
    dft_tile<1>(Wre, Wim, wsign, ir, ii, in_ld, rsb, rs_n, orr, oi, out_ld, NP, xt * 16, c0, scale,
                lane);
}

// interleaved (nb,T,N) c128  <->  planar [b][NP][TP] (channel major)
__global__ void k_tn_to_planar(const double* __restrict__ in, double* __restrict__ ore,
                               double* __restrict__ oim, const int T, const int N, const int NP,
                               const int TP) {
  const int b = blockIdx.y;
  const long tot = (long)NP * TP;
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < tot;
       e += (long)gridDim.x * blockDim.x) {
    const int x = (int)(e / TP), t = (int)(e % TP);
    double vr = 0.0, vi = 0.0;
    if (x < N && t < T) {
      const long o = (((long)b * T + t) * N + x) * 2;
      vr = in[o];
      vi = in[o + 1];
    }
    ore[(long)b * tot + e] = vr;
    oim[(long)b * tot + e] = vi;
  }
}

__global__ void k_planar_to_tn(const double* __restrict__ ire, const double* __restrict__ iim,
                               double* __restrict__ out, const int T, const int N, const int NP,
                               const int TP) {
  const int b = blockIdx.y;
  const long tot = (long)T * N;
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < tot;
       e += (long)gridDim.x * blockDim.x) {
    const int t = (int)(e / N), x = (int)(e % N);
    const long o = (long)b * NP * TP + (long)x * TP + t;
    out[((long)b * tot + e) * 2] = ire[o];
    out[((long)b * tot + e) * 2 + 1] = iim[o];
  }
}

// fop (N,N) c128 interleaved -> planar zero-padded [NP][NP]
__global__ void k_fop_planar(const double* __restrict__ fop, double* __restrict__ re,
                             double* __restrict__ im, const int N, const int NP) {
  const long tot = (long)NP * NP;
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < tot;
       e += (long)gridDim.x * blockDim.x) {
    const int k = (int)(e / NP), x = (int)(e % NP);
    double vr = 0.0, vi = 0.0;
    if (k < N && x < N) {
      vr = fop[((long)k * N + x) * 2];
      vi = fop[((long)k * N + x) * 2 + 1];
    }
    re[e] = vr;
    im[e] = vi;
  }
}

}  // namespace

int hpx_launch_dft(int nbl, int NP, int ncol, const double* Wre, const double* Wim, int conjW,
                   const double* inre, const double* inim, long in_bstride, int in_ld,
                   const double* rs, int rs_n, double* outre, double* outim, long out_bstride,
                   int out_ld, double scale, hipStream_t st) {
  if ((NP & 15) || (ncol & 15)) {
    hpx_set_error("hpx_launch_dft: NP and ncol must be multiples of 16");
    return HPX_EINVAL;
  }
  dim3 grid((NP / 16 + 3) / 4, nbl);
  hipLaunchKernelGGL(k_dft, grid, dim3(256), 0, st, Wre, Wim, conjW, inre, inim, in_bstride,
                     in_ld, rs, rs_n, outre, outim, out_bstride, out_ld, NP, ncol, scale);
  HPX_HIP(hipGetLastError());
  return HPX_OK;
}

int hpx_fop_to_planar(const double* fop, double* re, double* im, int N, int NP, hipStream_t st) {
  hipLaunchKernelGGL(k_fop_planar, dim3(256), dim3(256), 0, st, fop, re, im, N, NP);
  HPX_HIP(hipGetLastError());
  return HPX_OK;
}

extern "C" int hpx_dft_batched(int nb, int T, int N, const double* fop, const double* in,
                               double* out, int inverse, void* stream) {
  HPX_REQUIRE(nb > 0 && T > 0 && N > 0 && fop && in && out, "hpx_dft_batched: bad argument");
  hipStream_t st = (hipStream_t)stream;
  const int NP = ceil16(N), TP = ceil16(T);
  double *wre = nullptr, *wim = nullptr, *buf = nullptr;
  const size_t wb = (size_t)NP * NP * sizeof(double), pb = (size_t)nb * NP * TP * sizeof(double);
  HPX_HIP(hipMalloc(&wre, wb));
  HPX_HIP(hipMalloc(&wim, wb));
  HPX_HIP(hipMalloc(&buf, 4 * pb));
  double *ire = buf, *iim = buf + (size_t)nb * NP * TP, *ore = iim + (size_t)nb * NP * TP,
         *oim = ore + (size_t)nb * NP * TP;
  int rc = hpx_fop_to_planar(fop, wre, wim, N, NP, st);
  if (rc == HPX_OK) {
    hipLaunchKernelGGL(k_tn_to_planar, dim3(64, nb), dim3(256), 0, st, in, ire, iim, T, N, NP, TP);
    rc = hpx_launch_dft(nb, NP, TP, wre, wim, inverse ? 1 : 0, ire, iim, (long)NP * TP, TP, nullptr,
                        0, ore, oim, (long)NP * TP, TP, inverse ? 1.0 / N : 1.0, st);
  }
  if (rc == HPX_OK) {
    hipLaunchKernelGGL(k_planar_to_tn, dim3(64, nb), dim3(256), 0, st, ore, oim, out, T, N, NP, TP);
    if (hipGetLastError() != hipSuccess) rc = HPX_EHIP;
  }
  hipError_t e = hipStreamSynchronize(st);
  (void)hipFree(wre); (void)hipFree(wim); (void)hipFree(buf);
  if (e != hipSuccess) { hpx_set_error("hpx_dft_batched: %s", hipGetErrorString(e)); return HPX_EHIP; }
  return rc;
}
