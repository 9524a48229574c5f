// In-LDS radix-2 DIF FFT passes shared by k_fft (hpx_transform.hip) and the fused
// transform + residual kernel (hpx_chain.hip).
#pragma once
#include "hpx_internal.h"

namespace {

// R consecutive radix-2 DIF stages (starting at stage s) fused in registers: the 2^R elements
// base + m (N >> (s + R)), m = 0 .. 2^R - 1, only interact with each other over these stages,
// so one LDS round trip and one barrier serve R stages (N = 512: three passes instead of nine).
// Same butterflies, twiddles and (bit-reversed) output order as the unfused stages.
template <int R, int SIGN, int NTH = 256>
__device__ __forceinline__ void fft_pass(double* __restrict__ fre, double* __restrict__ fim,
                                         const double* __restrict__ tw, const int N, const int h,
                                         const int logN, const int s, const int tcs,
                                         const int tid) {
  constexpr int E = 1 << R;
  const int lq = logN - s - R;                 // log2 of the spacing between the E elements
  const int Hq = 1 << lq;
  const int ngroups = (N >> R) << tcs;
  for (int gidx = tid; gidx < ngroups; gidx += NTH) {      // NTH: threads of the workgroup
    const int tc = gidx & ((1 << tcs) - 1), gi = gidx >> tcs;
    const int j = gi & (Hq - 1), blk = gi >> lq;
    const int base = (((blk << R) << lq) + j) << tcs;
    double xr[E], xi[E];
#pragma unroll
    for (int m = 0; m < E; ++m) {
      xr[m] = fre[base + ((m << lq) << tcs) + tc];
      xi[m] = fim[base + ((m << lq) << tcs) + tc];
    }
#pragma unroll
    for (int r = 0; r < R; ++r) {
      constexpr int dummy = 0;
      (void)dummy;
      const int dist = E >> (r + 1);
#pragma unroll
      for (int m = 0; m < E; ++m) {
        if ((m / dist) & 1) continue;          // m is the upper element of its pair
        const int pidx = (((m & (dist - 1)) << lq) + j) << (s + r);     // twiddle exponent
        const double wr = tw[pidx], wi = (SIGN > 0 ? -1.0 : 1.0) * tw[h + pidx];
        const double ar = xr[m], ai = xi[m], br = xr[m + dist], bi = xi[m + dist];
        const double dr = ar - br, di = ai - bi;
        xr[m] = ar + br;
        xi[m] = ai + bi;
        xr[m + dist] = dr * wr - di * wi;
        xi[m + dist] = dr * wi + di * wr;
      }
    }
#pragma unroll
    for (int m = 0; m < E; ++m) {
      fre[base + ((m << lq) << tcs) + tc] = xr[m];
      fim[base + ((m << lq) << tcs) + tc] = xi[m];
    }
  }
}

}  // namespace
