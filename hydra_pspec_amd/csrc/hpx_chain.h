// Shared by the translation units of the chain (hpx_plan / hpx_setup / hpx_chain / hpx_post / hpx_woodbury .hip):
// deterministic block reductions, the plan's allocator, the per-iteration output descriptor, and the few host
// functions one unit calls in another.  (Round 5: hpx_chain.hip was one 2 900-line file holding five input modes.)
#pragma once
#include <math.h>
#include <stdarg.h>
#include <string.h>
#include "hpx_internal.h"
#include <string>
#include <stdlib.h>

namespace {


constexpr double SQRT2 = 1.4142135623730951;   // 2**0.5 (pspec.py:217)

// deterministic block reductions (256 threads): fixed shuffle tree + fixed wave order
__device__ __forceinline__ double block_sum(double v, double* red) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  return (red[0] + red[1]) + (red[2] + red[3]);
}
__device__ __forceinline__ int block_sum_int(int v, int* red) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  return red[0] + red[1] + red[2] + red[3];
}
__device__ __forceinline__ double block_min(double v, double* red) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmin(v, __shfl_down(v, o));
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  return fmin(fmin(red[0], red[1]), fmin(red[2], red[3]));
}
__device__ __forceinline__ double block_max(double v, double* red) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_down(v, o));
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  return fmax(fmax(red[0], red[1]), fmax(red[2], red[3]));
}

// ia = 1/a = sqrt(N / ps); bandpowers below HPX_PS_FLOOR (incl. zero) are treated as the floor:
// the channel's signal is then pinned to ~0, which is what a -> 0 means in the unscaled system.
#define HPX_PS_FLOOR 1e-280
__device__ __forceinline__ double inv_a(const double ps, const double dN) {
  // (ps < floor) is false for a NaN, which therefore propagates into the pivots and is reported
  return sqrt(dN / ((ps < HPX_PS_FLOOR) ? HPX_PS_FLOOR : ps));
}

// Plan-owned device buffer.  A pointer that is already set is released first, so that the
// setters (set_static / set_rng / set_solver) can be called again on the same plan without
// the plan growing.
template <typename Tp>
int dev_alloc(hpx_plan* p, Tp** ptr, size_t count) {
  if (*ptr) {
    for (size_t i = 0; i < p->allocs.size(); ++i)
      if (p->allocs[i].first == (void*)*ptr) {
        (void)hipFree(*ptr);
        p->bytes -= (int64_t)p->allocs[i].second;
        p->allocs.erase(p->allocs.begin() + i);
        break;
      }
    *ptr = nullptr;
  }
  void* q = nullptr;
  const size_t bytes = count * sizeof(Tp);
  hipError_t e = hipMalloc(&q, bytes ? bytes : 8);
  if (e != hipSuccess) {
    hpx_set_error("hipMalloc(%zu bytes) failed: %s", bytes, hipGetErrorString(e));
    return HPX_EHIP;
  }
  p->allocs.push_back(std::make_pair(q, bytes));
  p->bytes += (int64_t)bytes;
  *ptr = (Tp*)q;
  return HPX_OK;
}

// fg[u] = fg[u / T]  ((nbl,N,M) c128 -> (nbl*T,N,M))
__global__ void k_pt_expand_fg(const double* __restrict__ src, double* __restrict__ dst, const int T, const long per) {
  const int u = blockIdx.y;
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < per; e += (long)gridDim.x * blockDim.x)
    dst[(long)u * per + e] = src[(long)(u / T) * per + e];
}

// sum of the ln posterior's channel term from its sums over groups of 16 channels (k_draw, k_lnpost_combine): four
// groups make a block of 64, ((g0 + g1) + (g2 + g3)); the blocks are added in order
template <typename P>
__device__ __forceinline__ double lnpost_blocks(const P* g, const int nsub) {
  double s = 0.0;
  for (int q = 0; q < nsub; q += 4) {
    const double g0 = g[q], g1 = (q + 1 < nsub) ? g[q + 1] : 0.0, g2 = (q + 2 < nsub) ? g[q + 2] : 0.0,
                 g3 = (q + 3 < nsub) ? g[q + 3] : 0.0;
    s += (g0 + g1) + (g2 + g3);
  }
  return s;
}

}  // namespace


// ln-posterior term / beta, masked transform (flags), bandpower draw.  `rs` = row scaling of
// y' in the back transform (a = sqrt(ps/N), or NULL when X already holds s' = Sh' y').
struct IterOut {
  const double* ps_forced;   // already offset to this iteration, or NULL
  double *ps_out, *lnpost_out, *cr_out, *fg_out, *chisq_out;   // ps/lnpost offset to this iteration
  long ps_bstride, forced_bstride, lnpost_pitch;
  long cr_bstride, fg_bstride, chisq_bstride;
};

// ---- host functions shared between the units
hpx_gen_batch hpx_gen_of(const hpx_plan* p);              // hpx_plan.hip: the augmented-system generator of a plan ...
hpx_gen_batch hpx_gen_of_child(const hpx_plan* p);        // ... and of its per-time units
int hpx_plan_create_impl(hpx_plan** out, int nbl, int T, int N, int M, int extra_rhs);   // hpx_plan.hip (the per-time units are a plan of their own)
int hpx_mark(hpx_plan* p, hipStream_t st);                // hpx_plan.hip: next stage boundary (profiling event)
// hpx_post.hip: everything after the solve (back transform, residual, chi^2, ln posterior, bandpower draw)
int hpx_post_solve(hpx_plan* p, int it_abs, const IterOut& O, hipStream_t st);
// hpx_woodbury.hip: the rank-f correction of a dense-noise solve with flagged channels, on plan q's solution block
// (T data columns); per-time units: hpx_child_woodbury (transform of the unit solutions + the correction, T = 1)
int hpx_woodbury_correct(hpx_plan* q, int T, int iter_tag, hipStream_t st);
int hpx_child_woodbury(hpx_plan* c, int iter_tag, hipStream_t st);
