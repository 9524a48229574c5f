// Backward substitution L^H X = Z for orders beyond the register-resident form (hpx_backsolve.hip), third form:
// eight waves, one 16-column tile column each, the solution rows they all need through LDS once.
//
// k_backsolve (hpx_factor.hip) has each of a workgroup's four waves run down its own 32-wide block column over all
// the rows below the super-block of 128 columns, loading for every chunk of 16 rows its two tiles of L (8 KB) AND the
// chunk's 16 x 32 block of X (8 KB) -- the same block in all four waves.  Two operand sets of 64 registers next to 96
// of accumulators leave a wave one chunk in flight while it multiplies the other (50 registers spill as it is), and
// at C3 the kernel runs at 46 % MFMA-busy and 3 TB/s of factor bytes: neither pipe is the limit, the chunks are waited
// for.  Here
//   * a workgroup has EIGHT waves, wave w owns tile column 8 J + w of super-block J: 48 registers of accumulators,
//     16 per set of L operands -- 128 registers a lane are enough, so a CU holds sixteen waves (two workgroups),
//     each with its next chunk's tile in flight;
//   * a chunk's block of X is copied into a ring of eight LDS slots once per workgroup (LDS-DMA: 4 KB per wave and
//     group of four chunks, a group ahead of its use; one workgroup barrier per group) and read from there by all
//     eight waves: an eighth of the L2 requests for X, no registers for it beyond the k-step in hand;
//   * inside the super-block the tile columns are finished from the last to the first, 16 rows a step: the owner
//     multiplies by the inverse of its diagonal tile, and the finished rows go to the waves on its left through LDS
//     (no store, barrier, load through L2 in that chain); their tile of L is requested before the owner starts.
// The vector-memory traffic of the chunk loop is issued from asm statements and waited for with hand-counted vmcnt
// (the compiler cannot count through LDS-DMA and ordinary loads in flight together, see hpx_factor_tiles.h).
// Three-product complex arithmetic as in k_backsolve: A1 = Zr/2 - S1, A2 = Zr/2 - S2, A3 = Zi + S3 with S1 = lr xr,
// S2 = lm xi, S3 = (lr + lm)(xr - xi); X = conj(inv(L_jj))^T (A1 + A2, A1 - A2 + A3).
// Shapes: TP = 32 right-hand-side columns (Ntimes 17 .. 32: every BASELINE configuration); others stay with k_backsolve.
#include "hpx_internal.h"

#define HPX_INL __forceinline__

namespace {

typedef __attribute__((address_space(3))) double lds_f64;
typedef double bx_d2 __attribute__((ext_vector_type(2)));

constexpr int BX_NW = 8;                // waves: tile columns of a super-block
constexpr int BX_RING = 8;              // chunks of X in LDS: two groups of four
constexpr int BX_SLOT = 1024;           // doubles per slot: 16 rows x 32 columns, re then im

#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"
template <int OFF>
__device__ HPX_INL bx_d2 bx_ld16(const double* ubase, const unsigned lane_bytes) {
  bx_d2 r;
  asm volatile("global_load_dwordx4 %0, %1, %2 offset:%3 nt" : "=v"(r) : "v"(lane_bytes), "s"(ubase), "n"(OFF) : "memory");
  return r;
}
template <int OFF>
__device__ HPX_INL double bx_ld8(const double* ubase, const unsigned lane_bytes) {
  double r;
  asm volatile("global_load_dwordx2 %0, %1, %2 offset:%3" : "=v"(r) : "v"(lane_bytes), "s"(ubase), "n"(OFF) : "memory");
  return r;
}
// 4 KB (uniform base) + 16 lane -> LDS byte address `lds` + 16 lane, four 1 KB pieces
__device__ HPX_INL void bx_glds4k(const double* ubase, const unsigned lane16, const unsigned lds) {
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\t"
               "global_load_lds_dwordx4 %0, %1\n\t"
               "global_load_lds_dwordx4 %0, %1 offset:1024\n\t"
               "global_load_lds_dwordx4 %0, %1 offset:2048\n\t"
               "global_load_lds_dwordx4 %0, %1 offset:3072"
               :: "v"(lane16), "s"(ubase), "s"(lds) : "memory", "m0");
}
#pragma clang diagnostic pop
template <int N>
__device__ HPX_INL void bx_wait_vm() {
  asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N) : "memory");
}
__device__ HPX_INL void bx_lds_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}
__device__ HPX_INL unsigned bx_lds_addr(const double* p) {
  return (unsigned)(unsigned long)(const __attribute__((address_space(3))) double*)p;
}

// Step trace (-DHPX_BX_TRACE, tools/experiments/trace/bx_trace.py): [id << 32 | low word of s_memtime] per stamp and
// wave of workgroups 0 .. 3, written with SCALAR stores (lgkmcnt: the hand-counted vmcnt waits see nothing of them)
#ifdef HPX_BX_TRACE
#define HPX_BX_REC 1024
__device__ unsigned long long hpx_bx_trace[4 * BX_NW * HPX_BX_REC];
#define BX_TR(id_)                                                                               \
  if (tp) {                                                                                      \
    unsigned long long t_;                                                                       \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory");                \
    const unsigned long long rec_ = (t_ & 0xffffffffull) | ((unsigned long long)(id_) << 32);    \
    asm volatile("s_store_dwordx2 %0, %1, 0x0" :: "s"(rec_), "s"(tp) : "memory");                \
    tp += 1;                                                                                     \
  }
#else
#define BX_TR(id_)
#endif

// one L operand set: rows 4 g .. 4 g + 3 (k index (g, s)) of column li of a tile
struct BxL {
  bx_d2 re[2], im[2];
};
__device__ HPX_INL void bx_issue_l(BxL& S, const double* tile, const unsigned ll) {
  S.re[0] = bx_ld16<0>(tile, ll);
  S.re[1] = bx_ld16<16>(tile, ll);
  S.im[0] = bx_ld16<128>(tile, ll);
  S.im[1] = bx_ld16<144>(tile, ll);
}

// acc -= L[chunk rows, tile columns]^H X[chunk rows, :] with X from an LDS slot (row-major 16 x 32, re | im)
__device__ HPX_INL void bx_mma(d4 (&a1)[2], d4 (&a2)[2], d4 (&a3)[2], BxL& S, const lds_f64* xs, const int li,
                               const int g) {
#pragma unroll
  for (int h = 0; h < 2; ++h) asm volatile("" : "+v"(S.re[h]), "+v"(S.im[h]));   // (written by asm loads)
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    const double lr = S.re[s >> 1][s & 1], lm = S.im[s >> 1][s & 1];
    const double nlr = -lr, nlm = -lm, lsm = lr + lm;
#pragma unroll
    for (int tt = 0; tt < 2; ++tt) {
      const double xr = xs[(4 * g + s) * 32 + 16 * tt + li];
      const double xi = xs[512 + (4 * g + s) * 32 + 16 * tt + li];
      a1[tt] = mfma64(nlr, xr, a1[tt]);
      a2[tt] = mfma64(nlm, xi, a2[tt]);
      a3[tt] = mfma64(lsm, xr - xi, a3[tt]);
    }
  }
}

__global__ __launch_bounds__(64 * BX_NW, 4) void k_backsolve_x(const double* __restrict__ L_all,
                                                               const double* __restrict__ Wre_all,
                                                               const double* __restrict__ Wim_all,
                                                               double* __restrict__ Xre_all,
                                                               double* __restrict__ Xim_all, const int npad,
                                                               const int ld) {
  __shared__ double ring[BX_RING * BX_SLOT];         // 64 KB: two workgroups per CU
  const int b = blockIdx.x;
  const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  const int li = lane & 15, g = lane >> 4;
  const double* Lre = L_all + (long)b * npad * ld * 2;
  const int nct = npad >> 4, nsb = (nct + BX_NW - 1) / BX_NW;
  const int nblk = (npad + HPX_NB - 1) / HPX_NB;
  const double* Wgre = Wre_all + (long)b * nblk * 1024;
  const double* Wgim = Wim_all + (long)b * nblk * 1024;
  double* Xre = Xre_all + (long)b * npad * 32;
  double* Xim = Xim_all + (long)b * npad * 32;
  const unsigned ll = 8u * (li * 32 + 4 * g);        // L operand: column li of a tile, rows 4 g ..
  const unsigned l16 = 16u * lane;
  const lds_f64* xs = (const lds_f64*)ring;
#ifdef HPX_BX_TRACE
  unsigned long long* tp = (b < 4) ? hpx_bx_trace + ((long)b * BX_NW + wave) * HPX_BX_REC : nullptr;
#endif

  for (int J = nsb - 1; J >= 0; --J) {
    const int jt = BX_NW * J + wave;                 // this wave's tile column
    const bool have = jt < nct;
    const int jc = min(jt, nct - 1);                 // (a wave without one requests a valid tile and drops it)
    const int rbeg = min(npad, 16 * BX_NW * (J + 1));
    const int nch = (npad - rbeg) >> 4;              // chunks of phase A (uniform over the workgroup)
    BxL S0, S1;
    // the inverse of this wave's diagonal tile (the diagonal 16 x 16 sub-block of the 32 x 32 inverse block both factor
    // kernels write; A[m = li][k = g + 4 s] = W[k][m]), requested while phase A drains: on the chain of the 16-row
    // steps below it was a round trip to L2 per step
    double wr[4], wi[4];
    auto issue_w = [&] {
      const double* wr_ = Wgre + (long)(jc >> 1) * 1024 + (16 * (jc & 1)) * 33;
      const double* wi_ = Wgim + (long)(jc >> 1) * 1024 + (16 * (jc & 1)) * 33;
      const unsigned lw = 8u * (g * 32 + li);
      wr[0] = bx_ld8<0>(wr_, lw); wr[1] = bx_ld8<1024>(wr_, lw); wr[2] = bx_ld8<2048>(wr_, lw); wr[3] = bx_ld8<3072>(wr_, lw);
      wi[0] = bx_ld8<0>(wi_, lw); wi[1] = bx_ld8<1024>(wi_, lw); wi[2] = bx_ld8<2048>(wi_, lw); wi[3] = bx_ld8<3072>(wi_, lw);
    };
    // phase A (the rows below the super-block): chunk `ch` lives in ring slot ch mod 8; a group of four chunks is staged
    // by the eight waves (4 KB each: wave w the re (even w) or im (odd w) half of chunk 4 grp + w / 2); chunks past the
    // end are staged from the last one (never read).
    auto stage = [&](const int grp) {
      const int q = 4 * grp + (wave >> 1);
      const int ch = min(q, nch - 1);
      const double* src = ((wave & 1) ? Xim : Xre) + (long)(rbeg + 16 * ch) * 32;
      bx_glds4k(src, l16, bx_lds_addr(ring + (q & (BX_RING - 1)) * BX_SLOT + 512 * (wave & 1)));
    };
    auto tile_of = [&](const int ch) {
      return Lre + (((long)((rbeg >> 4) + min(ch, nch - 1)) * npad + 16 * jc) << 5);
    };
    // ---- Z[c][t] = conj(Laug[npad + t][c]) into the three-product accumulators
    d4 a1[2], a2[2], a3[2];
#pragma unroll
    for (int tt = 0; tt < 2; ++tt)
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        const double* zb = Lre + ((((long)((npad >> 4) + tt) * npad + 16 * jc + HPX_ACC_ROW(g, v)) << 5) + li);
        const double zr = zb[0], zi = zb[16];
        a1[tt][v] = 0.5 * zr;
        a2[tt][v] = 0.5 * zr;
        a3[tt][v] = -zi;
      }
    BX_TR(0x100 | J)
    __builtin_amdgcn_s_waitcnt(0x0f70);              // vmcnt(0): the compiler's loads are in (and it knows)
    BX_TR(0x200 | J)
    if (nch == 0) issue_w();
    if (nch > 0) {
      stage(0);
      bx_issue_l(S0, tile_of(0), ll);
      bx_issue_l(S1, tile_of(1), ll);
      bx_wait_vm<8>();                               // this wave's part of group 0 is in LDS
      asm volatile("s_barrier" ::: "memory");
      BX_TR(0x300 | J)
      // four chunks per trip (a group: staging, barrier) x two rounds of the two operand sets.  Before chunk ch is
      // multiplied, everything but the requests issued after its own is complete: the tile of chunk ch + 1 (4 loads)
      // and, in the first two chunks of a group, this wave's staging of the next group (4 loads).
#define BX_CHUNK(I_, SET_)                                                                      \
  {                                                                                             \
    const int ch_ = ch0 + (I_);                                                                 \
    if ((I_) == 0) stage((ch_ >> 2) + 1);            /* the next group, into the other half */ \
    bx_wait_vm<((I_) < 2) ? 8 : 4>();                                                           \
    if (have && ch_ < nch) bx_mma(a1, a2, a3, SET_, xs + (ch_ & (BX_RING - 1)) * BX_SLOT, li, g); \
    __builtin_amdgcn_sched_barrier(0);                                                          \
    bx_issue_l(SET_, tile_of(ch_ + 2), ll);                                                     \
    __builtin_amdgcn_sched_barrier(0);                                                          \
    if ((I_) == 3) bx_lds_barrier();                 /* the group is read, the next one staged */ \
  }
      for (int ch0 = 0; ch0 < nch; ch0 += 4) {
        BX_CHUNK(0, S0) BX_CHUNK(1, S1) BX_CHUNK(2, S0) BX_CHUNK(3, S1)
        BX_TR(0x400 | (ch0 >> 2))
      }
#undef BX_CHUNK
      issue_w();
      bx_wait_vm<0>();                               // the clamped requests past the end (and the inverse tile)
      asm volatile("s_barrier" ::: "memory");        // (no DMA of the loop lands in a slot of the hand-off below)
    }
    BX_TR(0x500 | J)
    // ---- phase B: the super-block's own tile columns, last to first; step w uses ring slot w
    for (int w = BX_NW - 1; w >= 0; --w) {
      const int jw = BX_NW * J + w;
      if (jw >= nct) continue;                       // uniform over the workgroup
      double* slot = ring + w * BX_SLOT;
      const bool left = wave < w;                    // (the waves to the left of an existing tile column have one)
      BX_TR(0x600 | w)
      if (left) bx_issue_l(S0, Lre + (((long)jw * npad + 16 * jt) << 5), ll);
      if (wave == w) {
        // X = conj(inv(L_jj))^T y
        if (nch == 0) bx_wait_vm<0>();               // (the top pass: the inverse tile was requested just now)
#pragma unroll
        for (int s = 0; s < 4; ++s) asm volatile("" : "+v"(wr[s]), "+v"(wi[s]));       // (written by asm loads)
        d4 fxr[2], fxi[2];
#pragma unroll
        for (int tt = 0; tt < 2; ++tt) {
          const d4 yr = a1[tt] + a2[tt], yi = a1[tt] - a2[tt] + a3[tt];
          d4 xr = {0., 0., 0., 0.}, xi = {0., 0., 0., 0.};
#pragma unroll
          for (int s = 0; s < 4; ++s) {
            xr = mfma64(wr[s], yr[s], xr);
            xr = mfma64(wi[s], yi[s], xr);
            xi = mfma64(wr[s], yi[s], xi);
            xi = mfma64(-wi[s], yr[s], xi);
          }
#pragma unroll
          for (int v = 0; v < 4; ++v) {
            const int r = HPX_ACC_ROW(g, v);
            slot[r * 32 + 16 * tt + li] = xr[v];
            slot[512 + r * 32 + 16 * tt + li] = xi[v];
          }
          fxr[tt] = xr;
          fxi[tt] = xi;
        }
        bx_lds_barrier();                            // the tile column's rows of X are in their slot ...
#pragma unroll
        for (int tt = 0; tt < 2; ++tt)               // ... and go to memory behind the barrier, off the chain of the steps
#pragma unroll
          for (int v = 0; v < 4; ++v) {
            const long xo = (long)(16 * jw + HPX_ACC_ROW(g, v)) * 32 + 16 * tt + li;
            Xre[xo] = fxr[tt][v];
            Xim[xo] = fxi[tt][v];
          }
      } else {
        bx_lds_barrier();
      }
      BX_TR(0x700 | w)
      if (left) {
        bx_wait_vm<0>();
        bx_mma(a1, a2, a3, S0, (const lds_f64*)slot, li, g);
      }
    }
    // the rows this pass stored are staged from global memory by the next one
    BX_TR(0x800 | J)
    bx_wait_vm<0>();
    bx_lds_barrier();
    BX_TR(0x900 | J)
  }
}

}  // namespace

#ifdef HPX_BX_TRACE
extern "C" int hpx_debug_bx_trace(unsigned long long* host) {
  HPX_HIP(hipMemcpyFromSymbol(host, HIP_SYMBOL(hpx_bx_trace), sizeof(unsigned long long) * 4 * BX_NW * HPX_BX_REC));
  return HPX_OK;
}
#endif

#ifndef HPX_BS_X
#define HPX_BS_X 1
#endif
// 1 where this form applies: 32 right-hand-side columns (orders of the register form are taken there first)
int hpx_backsolve_x_ok(int npad, int TP) { return HPX_BS_X && TP == 32 && npad >= 144; }

int hpx_launch_backsolve_x(int nbl, int npad, int ld, const double* L, const double* Wre, const double* Wim, double* Xre,
                           double* Xim, hipStream_t st) {
  hipLaunchKernelGGL(k_backsolve_x, dim3(nbl), dim3(64 * BX_NW), 0, st, L, Wre, Wim, Xre, Xim, npad, ld);
  HPX_HIP(hipGetLastError());
  return HPX_OK;
}
