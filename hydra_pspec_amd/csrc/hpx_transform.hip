// Batched centred DFT along the channel axis as a dense contraction with the
// (symmetric) Fourier operator on FP64 MFMA.  General N (zero padded to a
// multiple of 16).  out[b][x][c] = scale * sum_k W[x][k] * rs[b][k] * in[b][k][c].
#include "hpx_internal.h"
#include "hpx_fft.h"

namespace {

// One wave: one 16-row x-tile, CTN 16-column tiles.  Grid (x-tile groups of 4, nbl).
template <int CTN, bool RS>
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
  // k-steps in pairs with two operand register sets: the loads of the next step are in flight
  // while the current one is multiplied (nks = NP / 4 is a multiple of 4)
  const int nks = NP >> 2;
  double w0r, w0i, w1r, w1i, s0, s1, b0r[CTN], b0i[CTN], b1r[CTN], b1i[CTN];
#define HPX_DFT_LOAD(wr_, wi_, sc_, br_, bi_, ks_)                                  \
  {                                                                                 \
    const int k_ = 4 * (ks_) + g;                                                   \
    wr_ = Wre[(long)k_ * NP + x0 + li];                                             \
    wi_ = wsign * Wim[(long)k_ * NP + x0 + li];                                     \
    sc_ = RS ? rs[min(k_, rs_n - 1)] * (k_ < rs_n ? 1.0 : 0.0) : 1.0;               \
    _Pragma("unroll") for (int c = 0; c < CTN; ++c) {                               \
      const long o_ = (long)k_ * in_ld + c0 + 16 * c + li;                          \
      br_[c] = inre[o_];                                                            \
      bi_[c] = inim[o_];                                                            \
    }                                                                               \
  }
#define HPX_DFT_MMA(wr_, wi_, sc_, br_, bi_)                                        \
  _Pragma("unroll") for (int c = 0; c < CTN; ++c) {                                 \
    const double br = br_[c] * sc_, bi = bi_[c] * sc_;                              \
    ar[c] = mfma64(wr_, br, ar[c]);                                                 \
    ar[c] = mfma64(-wi_, bi, ar[c]);                                                \
    ai[c] = mfma64(wr_, bi, ai[c]);                                                 \
    ai[c] = mfma64(wi_, br, ai[c]);                                                 \
  }
  HPX_DFT_LOAD(w0r, w0i, s0, b0r, b0i, 0)
  for (int ks = 0; ks < nks; ks += 2) {
    HPX_DFT_LOAD(w1r, w1i, s1, b1r, b1i, ks + 1)
    __builtin_amdgcn_sched_barrier(0);
    HPX_DFT_MMA(w0r, w0i, s0, b0r, b0i)
    __builtin_amdgcn_sched_barrier(0);
    HPX_DFT_LOAD(w0r, w0i, s0, b0r, b0i, min(ks + 2, nks - 1))     // branch-free: re-read at the end
    __builtin_amdgcn_sched_barrier(0);
    HPX_DFT_MMA(w1r, w1i, s1, b1r, b1i)
    __builtin_amdgcn_sched_barrier(0);
  }
#undef HPX_DFT_LOAD
#undef HPX_DFT_MMA
#pragma unroll
  for (int c = 0; c < CTN; ++c)
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      const long o = (long)(x0 + HPX_ACC_ROW(g, v)) * out_ld + c0 + 16 * c + li;
      outre[o] = ar[c][v] * scale;
      outim[o] = ai[c][v] * scale;
    }
}

// The same product for NB consecutive batch members that share ONE operator and have a single 16-column
// tile each (the DPSS group stage: inv(cov) times every group's weighted modes): the wave keeps one
// accumulator tile per member, so an operator fragment feeds 4 NB MFMAs instead of 4 -- with one member
// per wave the kernel is bound by its loads (one operand load per MFMA pair), not by the matrix pipe.
template <int NB>
__global__ __launch_bounds__(256, 2) void k_dft_shared(const double* __restrict__ Wre, const double* __restrict__ Wim,
                                                       const int conjW, const double* __restrict__ inre,
                                                       const double* __restrict__ inim, const long in_bstride,
                                                       const int in_ld, double* __restrict__ outre,
                                                       double* __restrict__ outim, const long out_bstride,
                                                       const int out_ld, const int NP, const double scale,
                                                       const int nbl) {
  const int b0 = blockIdx.y * NB;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int xt = blockIdx.x * 4 + wave;
  if (xt * 16 >= NP) return;
  const int li = lane & 15, g = lane >> 4, x0 = xt * 16;
  const double wsign = conjW ? -1.0 : 1.0;
  long boff[NB];
#pragma unroll
  for (int c = 0; c < NB; ++c) boff[c] = (long)min(b0 + c, nbl - 1) * in_bstride;     // (the tail repeats the last member)
  d4 ar[NB], ai[NB];
#pragma unroll
  for (int c = 0; c < NB; ++c) {
    ar[c] = (d4){0., 0., 0., 0.};
    ai[c] = (d4){0., 0., 0., 0.};
  }
  const int nks = NP >> 2;
  double w0r, w0i, w1r, w1i, b0r[NB], b0i[NB], b1r[NB], b1i[NB];
#define HPX_DFS_LOAD(wr_, wi_, br_, bi_, ks_)                                       \
  {                                                                                 \
    const int k_ = 4 * (ks_) + g;                                                   \
    wr_ = Wre[(long)k_ * NP + x0 + li];                                             \
    wi_ = wsign * Wim[(long)k_ * NP + x0 + li];                                     \
    _Pragma("unroll") for (int c = 0; c < NB; ++c) {                                \
      const long o_ = boff[c] + (long)k_ * in_ld + li;                              \
      br_[c] = inre[o_];                                                            \
      bi_[c] = inim[o_];                                                            \
    }                                                                               \
  }
#define HPX_DFS_MMA(wr_, wi_, br_, bi_)                                             \
  _Pragma("unroll") for (int c = 0; c < NB; ++c) {                                  \
    ar[c] = mfma64(wr_, br_[c], ar[c]);                                             \
    ar[c] = mfma64(-wi_, bi_[c], ar[c]);                                            \
    ai[c] = mfma64(wr_, bi_[c], ai[c]);                                             \
    ai[c] = mfma64(wi_, br_[c], ai[c]);                                             \
  }
  HPX_DFS_LOAD(w0r, w0i, b0r, b0i, 0)
  for (int ks = 0; ks < nks; ks += 2) {
    HPX_DFS_LOAD(w1r, w1i, b1r, b1i, ks + 1)
    __builtin_amdgcn_sched_barrier(0);
    HPX_DFS_MMA(w0r, w0i, b0r, b0i)
    __builtin_amdgcn_sched_barrier(0);
    HPX_DFS_LOAD(w0r, w0i, b0r, b0i, min(ks + 2, nks - 1))
    __builtin_amdgcn_sched_barrier(0);
    HPX_DFS_MMA(w1r, w1i, b1r, b1i)
    __builtin_amdgcn_sched_barrier(0);
  }
#undef HPX_DFS_LOAD
#undef HPX_DFS_MMA
#pragma unroll
  for (int c = 0; c < NB; ++c)
    if (b0 + c < nbl)
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        const long o = (long)(b0 + c) * out_bstride + (long)(x0 + HPX_ACC_ROW(g, v)) * out_ld + li;
        outre[o] = ar[c][v] * scale;
        outim[o] = ai[c][v] * scale;
      }
}

__global__ __launch_bounds__(256, 2) void k_dft(const double* Wre, const double* Wim,
                                                const int conjW,
                                             const double* __restrict__ inre,
                                             const double* __restrict__ inim,
                                             const long in_bstride, const int in_ld,
                                             const double* __restrict__ rs, const int rs_n,
                                             double* __restrict__ outre,
                                             double* __restrict__ outim, const long out_bstride,
                                             const int out_ld, const int NP, const int ncol,
                                             const double scale, const long W_bstride) {
  const int b = blockIdx.y;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int xt = blockIdx.x * 4 + wave;
  if (xt * 16 >= NP) return;
  Wre += (long)b * W_bstride;
  Wim += (long)b * W_bstride;
  const double wsign = conjW ? -1.0 : 1.0;
  const double* ir = inre + (long)b * in_bstride;
  const double* ii = inim + (long)b * in_bstride;
  double* orr = outre + (long)b * out_bstride;
  double* oi = outim + (long)b * out_bstride;
  const double* rsb = rs ? rs + (long)b * rs_n : nullptr;
  int c0 = 0;
  if (rsb) {                                   // optional row scaling: its own instantiation
    for (; c0 + 32 <= ncol; c0 += 32)
      dft_tile<2, true>(Wre, Wim, wsign, ir, ii, in_ld, rsb, rs_n, orr, oi, out_ld, NP, xt * 16, c0, scale, lane);
    if (c0 < ncol)
      dft_tile<1, true>(Wre, Wim, wsign, ir, ii, in_ld, rsb, rs_n, orr, oi, out_ld, NP, xt * 16, c0, scale, lane);
  } else {
    for (; c0 + 32 <= ncol; c0 += 32)
      dft_tile<2, false>(Wre, Wim, wsign, ir, ii, in_ld, rsb, rs_n, orr, oi, out_ld, NP, xt * 16, c0, scale, lane);
    if (c0 < ncol)
      dft_tile<1, false>(Wre, Wim, wsign, ir, ii, in_ld, rsb, rs_n, orr, oi, out_ld, NP, xt * 16, c0, scale, lane);
  }
}

// Power-of-two N: in-place radix-2 decimation-in-frequency FFT in LDS, TC columns per
// workgroup (columns = time samples, contiguous in memory).  The centred transform is the
// plain one with sign flips on both sides (N % 4 == 0):
//   (F v)[k] = (-1)^k sum_x (-1)^x v[x] e^{-2 pi i k x / N},   F^H likewise with e^{+...}.
// Twiddles e^{-2 pi i j / N} are row N/2+1 of the operator itself (numpy's exp on the host).
#ifndef HPX_FFT_DIAG
#define HPX_FFT_DIAG 0     // timing-only ablations (wrong results): 1 no passes, 2 no stores, 4 no loads
#endif
template <int SIGN>
__global__ __launch_bounds__(256) void k_fft(const double* __restrict__ Wre,
                                             const double* __restrict__ Wim,
                                             const double* __restrict__ inre,
                                             const double* __restrict__ inim,
                                             const long in_bstride, const int in_ld,
                                             const double* __restrict__ rs, const int rs_n,
                                             double* __restrict__ outre,
                                             double* __restrict__ outim, const long out_bstride,
                                             const int out_ld, const int N, const int logN,
                                             const int ncol, const int tcs, const double scale,
                                             const int nbl, const int ncg) {
  extern __shared__ double fl[];
  const int TC = 1 << tcs;
  double* fre = fl;
  double* fim = fl + ((long)N << tcs);
  double* tw = fim + ((long)N << tcs);         // cos(2 pi j / N), then -sin(2 pi j / N), j < N/2
  // The column groups of one baseline read and write interleaved pieces of the same cache lines
  // (TC consecutive doubles of every row): they go to ONE XCD, whose L2 then merges them.
  // (Workgroup ids are dealt round-robin to the 8 XCDs.)
  const int b = ((int)(blockIdx.x >> 3) / ncg) * 8 + (int)(blockIdx.x & 7);
  if (b >= nbl) return;
  const int c0 = ((int)(blockIdx.x >> 3) % ncg) * TC, tid = threadIdx.x;
  const double* ir = inre + (long)b * in_bstride;
  const double* ii = inim + (long)b * in_bstride;
  const double* rsb = rs ? rs + (long)b * rs_n : nullptr;
  const int h = N >> 1;
  for (int j = tid; j < h; j += 256) {
    tw[j] = Wre[(long)(h + 1) * N + h + j];
    tw[h + j] = Wim[(long)(h + 1) * N + h + j];
  }
  // Loads in batches of 16 per thread, all issued before the first LDS write: one element at a
  // time the loop waits out a full memory round trip per iteration (16 of them at N = 1024),
  // which was two thirds of the kernel's time.
#ifndef HPX_FFT_UB
#define HPX_FFT_UB 16
#endif
  constexpr int UB = HPX_FFT_UB;
  const int total = N << tcs;
  for (int e0 = tid; e0 < total; e0 += 256 * UB) {
    double vr[UB], vi[UB], sc[UB];
#pragma unroll
    for (int u = 0; u < UB; ++u) {
      const int e = min(e0 + 256 * u, total - 1);      // clamped: in-bounds, unused beyond the end
      const int k = e >> tcs, tc = min(e & (TC - 1), ncol - 1 - c0);
      vr[u] = (HPX_FFT_DIAG & 4) ? 0.0 : ir[(long)k * in_ld + c0 + tc];
      vi[u] = (HPX_FFT_DIAG & 4) ? 0.0 : ii[(long)k * in_ld + c0 + tc];
      sc[u] = (k & 1) ? -1.0 : 1.0;
    }
#pragma unroll
    for (int u = 0; u < UB; ++u) {
      const int e = e0 + 256 * u;
      if (e < total) {
        const bool live = c0 + (e & (TC - 1)) < ncol;
        fre[e] = live ? vr[u] * sc[u] : 0.0;
        fim[e] = live ? vi[u] * sc[u] : 0.0;
      }
    }
  }
  if (rsb) {                         // optional row scaling (no caller on the hot path uses it)
    for (int e = tid; e < total; e += 256) {
      const int k = e >> tcs;
      const double r = (k < rs_n) ? rsb[k] : 0.0;
      fre[e] *= r;
      fim[e] *= r;
    }
  }
  int s = (HPX_FFT_DIAG & 1) ? logN : 0;
  for (; s + 3 <= logN; s += 3) {
    __syncthreads();
    fft_pass<3, SIGN>(fre, fim, tw, N, h, logN, s, tcs, tid);
  }
  if (logN - s == 2) {
    __syncthreads();
    fft_pass<2, SIGN>(fre, fim, tw, N, h, logN, s, tcs, tid);
  } else if (logN - s == 1) {
    __syncthreads();
    fft_pass<1, SIGN>(fre, fim, tw, N, h, logN, s, tcs, tid);
  }
  __syncthreads();
  double* orr = outre + (long)b * out_bstride;
  double* oi = outim + (long)b * out_bstride;
  for (int e = tid; e < (N << tcs); e += 256) {
    const int pidx = e >> tcs, tc = e & (TC - 1);
    if (c0 + tc >= ncol) continue;
    if ((HPX_FFT_DIAG & 2) && fre[e] != 12345.678) continue;
    const int x = (int)(__brev((unsigned)pidx) >> (32 - logN));
    const double sc = (x & 1) ? -scale : scale;
    orr[(long)x * out_ld + c0 + tc] = fre[e] * sc;
    oi[(long)x * out_ld + c0 + tc] = fim[e] * sc;
  }
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

// The FFT path needs W to be the centred Fourier operator of order NP exactly (no padding):
// callers that pass another matrix (DPSS inverse covariance) set fft_ok = 0.
int hpx_dft_use_fft = 1;

int hpx_launch_dft(int nbl, int NP, int ncol, const double* Wre, const double* Wim, int conjW,
                   const double* inre, const double* inim, long in_bstride, int in_ld,
                   const double* rs, int rs_n, double* outre, double* outim, long out_bstride,
                   int out_ld, double scale, hipStream_t st, int fft_ok, long W_bstride) {
  if ((NP & 15) || (ncol & 15)) {
    hpx_set_error("hpx_launch_dft: NP and ncol must be multiples of 16");
    return HPX_EINVAL;
  }
  if (hpx_dft_use_fft && NP >= 16 && NP <= 4096 && (NP & (NP - 1)) == 0 && fft_ok && W_bstride == 0) {
    int logN = 0;
    while ((1 << logN) < NP) ++logN;
    int TC = 4096 / NP;                 // 64 KiB of LDS per workgroup (+ the twiddle table)
    if (TC > 16) TC = 16;
    if (TC < 1) TC = 1;
    int tcs = 0;
    while ((1 << tcs) < TC) ++tcs;
    const size_t lds = ((size_t)NP * TC * 2 + NP) * sizeof(double);
    static hpx_lds_limit lim_fwd, lim_inv;
    HPX_TRY(lim_fwd.ensure(reinterpret_cast<const void*>(&k_fft<1>), lds));
    HPX_TRY(lim_inv.ensure(reinterpret_cast<const void*>(&k_fft<-1>), lds));
    const int ncg = (ncol + TC - 1) / TC;
    dim3 grid(((nbl + 7) / 8) * 8 * ncg);
    if (conjW)
      hipLaunchKernelGGL(k_fft<1>, grid, dim3(256), lds, st, Wre, Wim, inre, inim, in_bstride, in_ld,
                         rs, rs_n, outre, outim, out_bstride, out_ld, NP, logN, ncol, tcs, scale, nbl, ncg);
    else
      hipLaunchKernelGGL(k_fft<-1>, grid, dim3(256), lds, st, Wre, Wim, inre, inim, in_bstride, in_ld,
                         rs, rs_n, outre, outim, out_bstride, out_ld, NP, logN, ncol, tcs, scale, nbl, ncg);
    HPX_HIP(hipGetLastError());
    return HPX_OK;
  }
  if (W_bstride == 0 && ncol == 16 && nbl >= 4 && !rs) {     // one operator, one column tile per member
    dim3 grid4((NP / 16 + 3) / 4, (nbl + 3) / 4);
    hipLaunchKernelGGL(k_dft_shared<4>, grid4, dim3(256), 0, st, Wre, Wim, conjW, inre, inim, in_bstride, in_ld,
                       outre, outim, out_bstride, out_ld, NP, scale, nbl);
    HPX_HIP(hipGetLastError());
    return HPX_OK;
  }
  dim3 grid((NP / 16 + 3) / 4, nbl);
  hipLaunchKernelGGL(k_dft, grid, dim3(256), 0, st, Wre, Wim, conjW, inre, inim, in_bstride,
                     in_ld, rs, rs_n, outre, outim, out_bstride, out_ld, NP, ncol, scale, W_bstride);
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
  const size_t wn = (size_t)NP * NP, pn = (size_t)nb * NP * TP;
  hpx_devbuf wbuf, dbuf;
  HPX_TRY(wbuf.alloc(2 * wn));
  HPX_TRY(dbuf.alloc(4 * pn));
  double *wre = wbuf.p, *wim = wre + wn;
  double *ire = dbuf.p, *iim = ire + pn, *ore = iim + pn, *oim = ore + pn;
  int rc = hpx_fop_to_planar(fop, wre, wim, N, NP, st);
  if (rc == HPX_OK) {
    hipLaunchKernelGGL(k_tn_to_planar, dim3(64, nb), dim3(256), 0, st, in, ire, iim, T, N, NP, TP);
    rc = hpx_launch_dft(nb, NP, TP, wre, wim, inverse ? 1 : 0, ire, iim, (long)NP * TP, TP, nullptr,
                        0, ore, oim, (long)NP * TP, TP, inverse ? 1.0 / N : 1.0, st, N == NP);
  }
  if (rc == HPX_OK) {
    hipLaunchKernelGGL(k_planar_to_tn, dim3(64, nb), dim3(256), 0, st, ore, oim, out, T, N, NP, TP);
    if (hipGetLastError() != hipSuccess) rc = HPX_EHIP;
  }
  hipError_t e = hipStreamSynchronize(st);
  if (e != hipSuccess) { hpx_set_error("hpx_dft_batched: %s", hipGetErrorString(e)); return HPX_EHIP; }
  return rc;
}
