// Shared pieces of the register-tile factorisations (hpx_factor_wide.hip, hpx_factor_split.hip): the staging
// buffers, the hand-placed vector-memory helpers, a tile's initial values and the fused 16 x 16
// Cholesky + inverse.  Included once per translation unit (everything sits in an anonymous namespace; the
// LDS arrays are per kernel).
#pragma once
#include "hpx_internal.h"

#define HPX_INL __forceinline__

namespace {

typedef __attribute__((address_space(3))) double lds_f64;
typedef double cplx __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) cplx lds_cplx;

constexpr int KC = 16;             // columns per chunk
constexpr int PP = KC / 4;         // 1 KB pieces (4 columns) per tile and chunk = k-steps per chunk
constexpr int TILE_D = KC * 32;    // doubles of one tile's chunk
constexpr int BUF_D = 8 * TILE_D;  // eight tiles per buffer

#ifndef HPX_TILES_STAGE0_D
#define HPX_TILES_STAGE0_D BUF_D
#endif
__shared__ double hpx_stage0[HPX_TILES_STAGE0_D];
__shared__ double hpx_stage1[BUF_D];
template <int PAR> __device__ HPX_INL double* stage_buf() { return PAR ? hpx_stage1 : hpx_stage0; }
// F's scratch over the staging buffers: Xs[8 tiles][512] = buffer 0 (slot 0: the diagonal tile handed to the
// elimination; the K-split partial sums of narrow_column in slots 1..4); in buffer 1 one inverse tile and the
// elimination's matrices
constexpr int FV_OFF = 0, FD_OFF = 512, F_END = FD_OFF + 2 * 2 * 16 * 17 + 16;
static_assert(F_END <= BUF_D, "F scratch must fit a staging buffer");

// ---- vector-memory traffic of the staged loops: ALL of it through inline asm, waits included ----------------
// The compiler's own s_waitcnt insertion cannot be used here: once an LDS-DMA and an ordinary global load are
// both in flight it waits with vmcnt(0) at the first use of the loaded register (measured: the next chunk's
// DMA, just issued, then had to land before the step could compute), and it adds a vmcnt(0) in front of any
// ds_read that might alias a pending DMA.  Operations issued from asm statements are invisible to it; the
// counted waits below are placed by hand, on the facts that the counter retires in issue order and that every
// wave issues the same number of operations per step.  Whatever else the compiler issues in between (the loads
// of a strip's initial values, a spill) only makes a counted wait stricter, never wrong.
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"
// one tile's chunk (16 columns = four 1 KB pieces) from (uniform base) + (lane offset, bytes) to the LDS byte
// address `lds` + 16 lane: the instruction offset moves the global and the LDS address together
__device__ HPX_INL void glds_tile(const double* ubase, const unsigned lane_bytes, const unsigned lds) {
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\t"
               "global_load_lds_dwordx4 %0, %1\n\t"
               "global_load_lds_dwordx4 %0, %1 offset:1024\n\t"
               "global_load_lds_dwordx4 %0, %1 offset:2048\n\t"
               "global_load_lds_dwordx4 %0, %1 offset:3072"
               :: "v"(lane_bytes), "s"(ubase), "s"(lds) : "memory", "m0");
}
#pragma clang diagnostic pop
// non-temporal 8-byte load from (uniform base) + (lane offset, bytes) + OFF; the value may be used only behind a
// wait_vm_keep that names it
template <int OFF>
__device__ HPX_INL double ld_nt(const double* ubase, const unsigned lane_bytes) {
  double r;
  asm volatile("global_load_dwordx2 %0, %1, %2 offset:%3 nt" : "=v"(r) : "v"(lane_bytes), "s"(ubase), "n"(OFF) : "memory");
  return r;
}
template <int OFF>
__device__ HPX_INL void st_g(double* ubase, const unsigned lane_bytes, const double v) {
  asm volatile("global_store_dwordx2 %0, %1, %2 offset:%3" :: "v"(lane_bytes), "v"(v), "s"(ubase), "n"(OFF) : "memory");
}
#ifdef HPX_DBG_WAIT0
#define HPX_WAITN(n_) 0
#else
#define HPX_WAITN(n_) (n_)
#endif
template <int N>
__device__ HPX_INL void wait_vm() {
  asm volatile("s_waitcnt vmcnt(%0)" :: "n"(HPX_WAITN(N)) : "memory");
}
// The row-operand registers were written by ld_nt asm statements the compiler knows nothing about: it must
// not touch them before the wait that covers their loads.  The wait itself has no operands (tied operands
// let the register allocator put copies of the not-yet-loaded registers IN FRONT of it: observed, as
// timing-dependent garbage); this anchor follows it, ties the pair, and everything that uses it depends on it.
__device__ HPX_INL void anchor_pair(double& a, double& b) {
  asm volatile("" : "+v"(a), "+v"(b));
}
__device__ HPX_INL void wg_barrier() {
  asm volatile("s_barrier" ::: "memory");
}
// The compiler's own loads (a tile's initial values from the edge tiles or the factor buffer) are complete, and
// the compiler knows it: otherwise it waits for them at their first use, a static s_waitcnt vmcnt(0) INSIDE the
// step loops that drains the hand-placed traffic there on every trip.
__device__ HPX_INL void wait_compiler_loads() {
  __builtin_amdgcn_s_waitcnt(0x0f70);
}
__device__ HPX_INL unsigned lds_addr(const double* p) {
  return (unsigned)(unsigned long)(const __attribute__((address_space(3))) double*)p;
}
// Workgroup barrier for data handed over through LDS only: the wave's LDS operations are complete, its global
// stores need not be (__syncthreads() also drains vmcnt, i.e. waits ~1-2 us for every store issued just before).
__device__ HPX_INL void lds_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}
__device__ HPX_INL double rsqrt_nr(const double d) {
  double q = __builtin_amdgcn_rsq(d);
  q = fma(q * 0.5, fma(-d * q, q, 1.0), q);
  q = fma(q * 0.5, fma(-d * q, q, 1.0), q);
  return q;
}
#define HPX_NTLD(p_) __builtin_nontemporal_load(p_)
// The lane / thread index through an empty asm: what is derived from it afterwards (LDS addresses, masks)
// cannot be hoisted out of the phase it is used in.  Otherwise dozens of such per-thread invariants are
// computed once at kernel entry, live across the register-bound S and P loops, are spilled there and reloaded
// inside F -- each reload behind an s_waitcnt vmcnt(0) that drains the stores of the step before (~2 us a time).
__device__ HPX_INL int opaque(int v) {
  asm volatile("" : "+v"(v));
  return v;
}
// the same for a wave-uniform value (stays in a scalar register): what is derived from the result inside a branch
// is computed inside that branch, not hoisted in front of it
__device__ HPX_INL int opaque_s(int v) {
  asm volatile("" : "+s"(v));
  return v;
}

// 1 / a and the circulant's first column with their address space in the type (LDS copies when GLDS): through
// the generic pointers of hpx_gen they become FLAT loads, which count on vmcnt as well and make the compiler
// wait for everything in flight at their first use
typedef __attribute__((address_space(1))) double glb_f64;
template <bool GLDS> struct GenVec;
template <> struct GenVec<true> { const lds_f64 *ia, *cre, *cim; };
template <> struct GenVec<false> { const glb_f64 *ia, *cre, *cim; };

struct WideCtx {
  double* Lb;          // this baseline's factor
  double* Vt;          // this baseline's inverse diagonal tiles
  double* Wgre;        // this baseline's 32 x 32 inverse blocks
  double* Wgim;
  long ptile;          // doubles per 16-row panel
  int npad, nct, nrt, wave, lane, tid;
#ifdef HPX_WIDE_TRACE
  mutable unsigned long long* tp;     // this wave's next trace record (a debugging build)
#endif
};
// Step trace of the wide form (-DHPX_WIDE_TRACE, tools/experiments/trace/wide_trace.py): one 8-byte record per stamp
// and wave, [id << 32 | low word of s_memtime], written with SCALAR stores -- they count on lgkmcnt, so the
// hand-counted vmcnt waits of the staged loops see nothing of them.
#ifdef HPX_WIDE_TRACE
#define HPX_WTRACE_REC 1024
__device__ HPX_INL void wtrace(const WideCtx& X, const unsigned id) {
  unsigned long long t;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory");
  const unsigned long long rec = (t & 0xffffffffull) | ((unsigned long long)id << 32);
  asm volatile("s_store_dwordx2 %0, %1, 0x0" :: "s"(rec), "s"(X.tp) : "memory");
  X.tp += 1;
}
#define HPX_TR(X_, ph_, sb_, a_, b_) wtrace(X_, ((unsigned)(ph_) << 24) | ((unsigned)(sb_) << 16) | ((unsigned)(a_) << 8) | (unsigned)(b_))
#else
#define HPX_TR(X_, ph_, sb_, a_, b_)
#endif

// entry (r, c), r >= c, of the augmented matrix before the factorisation: closed form, edge tiles or
// the factor buffer (hpx_internal.h)
template <bool GEN, bool GLDS>
__device__ HPX_INL void entry_init(const hpx_gen& G, const GenVec<GLDS>& V, const double* __restrict__ Lb, const int r,
                                   const int c, const int npad, double& vr, double& vi) {
  if (GEN && c < G.rmin) {
    if (r < G.rmin) {
      if (r > c) { vr = V.cre[r - c]; vi = V.cim[r - c]; }
      else if (r == c) { const double ic = V.ia[c]; vr = fma(ic, ic, V.cre[0]); vi = 0.0; }
      else { vr = 0.0; vi = 0.0; }
    } else {
      hpx_edge_init<GEN>(G, Lb, Lb + 16, r, c, npad, G.ere != nullptr, vr, vi);
    }
  } else {
    const long off = HPX_LIDX(r, c, npad);
    vr = Lb[off];
    vi = Lb[off + 16];
  }
}
// the closed-form part of tile_init alone (signal x signal tile, r0 >= c0, both below rmin): no vector-memory
// instruction on this path when the vectors are in LDS
template <bool GLDS>
__device__ HPX_INL void tile_init_closed(const GenVec<GLDS>& V, const int r0, const int c0, const int li, const int g,
                                         d4& vr, d4& vi) {
  if (r0 > c0) {
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      const int d = r0 - c0 + li - HPX_ACC_ROW(g, v);
      vr[v] = V.cre[d];
      vi[v] = V.cim[d];
    }
  } else {
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      const int d = li - HPX_ACC_ROW(g, v);
      const int dd = d > 0 ? d : 0;
      const double ic = V.ia[c0 + li];
      const double cr = V.cre[dd], cm = V.cim[dd];
      vr[v] = d > 0 ? cr : (d == 0 ? fma(ic, ic, cr) : 0.0);
      vi[v] = d > 0 ? cm : 0.0;
    }
  }
}
// the 16 x 16 tile at (r0, c0), r0 >= c0, as acc^T: lane li <-> row r0 + li, register v <-> column c0 + g + 4 v.
// rmin is a multiple of 32, so a tile lies on one side of it and the source is chosen per tile.
// TAILGEN (with GEN): the tiles of the last columns (c0 >= rmin: foreground x foreground block, identity padding, their
// right-hand sides) are generated here too, entry by entry (hpx_gen_entry -- what k_assemble_tail lays out for the
// other forms): the split form then needs no assembly launch in front of it.
template <bool GEN, bool GLDS, bool TAILGEN = false>
__device__ HPX_INL void tile_init(const hpx_gen& G, const GenVec<GLDS>& V, const double* __restrict__ Lb, const int r0,
                                  const int c0, const int npad, const int li, const int g, d4& vr, d4& vi) {
  if (GEN && c0 < G.rmin) {
    if (r0 < G.rmin) {
      if (r0 > c0) {
#pragma unroll
        for (int v = 0; v < 4; ++v) {
          const int d = r0 - c0 + li - HPX_ACC_ROW(g, v);
          vr[v] = V.cre[d];
          vi[v] = V.cim[d];
        }
      } else {
#pragma unroll
        for (int v = 0; v < 4; ++v) {
          const int d = li - HPX_ACC_ROW(g, v);
          const int dd = d > 0 ? d : 0;
          const double ic = V.ia[c0 + li];
          const double cr = V.cre[dd], cm = V.cim[dd];
          vr[v] = d > 0 ? cr : (d == 0 ? fma(ic, ic, cr) : 0.0);
          vi[v] = d > 0 ? cm : 0.0;
        }
      }
    } else if (G.ere != nullptr) {
      // an edge tile (foreground rows, padding, right-hand sides): hpx_edge_init's arithmetic, operation for
      // operation, but with the tile's four column loads issued together through global-address-space pointers and
      // the row test taken once per tile (r0 and npad are multiples of 16: a tile lies on one side).  Element by
      // element through the generic pointers of hpx_gen every value was a FLAT load behind a per-lane branch and a
      // full wait -- some 130 dependent memory round trips per strip of right-hand-side rows (80 - 100 us per
      // super-block at C3: tools/experiments/trace/wide_trace.py).
      const glb_f64* e = (const glb_f64*)G.ere + HPX_EIDX(r0, c0 + g, G.rmin) + li;
      double er[4], ei[4];
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        er[v] = e[(4 * v) * 32];
        ei[v] = e[(4 * v) * 32 + 16];
      }
      if (r0 >= npad) {
        if (G.has_omega) {
          const int t0 = r0 - npad;
          const glb_f64* pr = (const glb_f64*)G.p2tre + ((((long)(t0 >> 4) * G.NP + c0 + g) << 4) + li);
          const glb_f64* pi = (const glb_f64*)G.p2tim + ((((long)(t0 >> 4) * G.NP + c0 + g) << 4) + li);
          double qr[4], qi[4], ic[4];
#pragma unroll
          for (int v = 0; v < 4; ++v) {
            qr[v] = pr[(4 * v) * 16];
            qi[v] = pi[(4 * v) * 16];
            ic[v] = V.ia[c0 + g + 4 * v];
          }
#pragma unroll
          for (int v = 0; v < 4; ++v) {
            er[v] = fma(ic[v], qr[v], er[v]);
            ei[v] = fma(ic[v], qi[v], ei[v]);
          }
        }
#pragma unroll
        for (int v = 0; v < 4; ++v) ei[v] = -ei[v];
      }
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        vr[v] = er[v];
        vi[v] = ei[v];
      }
    } else {
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        double a, b;
        hpx_edge_init<GEN>(G, Lb, Lb + 16, r0 + li, c0 + HPX_ACC_ROW(g, v), npad, false, a, b);
        vr[v] = a;
        vi[v] = b;
      }
    }
  } else if (GEN && TAILGEN) {
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      double a, b;
      hpx_gen_entry(G, r0 + li, c0 + HPX_ACC_ROW(g, v), npad, a, b);
      vr[v] = a;
      vi[v] = b;
    }
  } else {
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      const long off = HPX_LIDX(r0 + li, c0 + HPX_ACC_ROW(g, v), npad);
      vr[v] = Lb[off];
      vi[v] = Lb[off + 16];
    }
  }
}

// ---- 16 x 16 fused Cholesky + inverse (all 256 threads; hpx_factor.hip diag_panel, steps A / D) -----
// in: Ein (re | im, row-major [r][c], leading dimension 16) lower part incl. diagonal, written by THIS
//     thread or in front of a barrier.
// out: L tile -> global; inv(L) tile -> Vt (global, tile layout, zero above the diagonal), LDS copy `Vs`
//      (tile layout with the odd-column swizzle), and the 16 x 16 sub-block of the 32 x 32 inverse block.
__device__ HPX_INL bool elim16(const WideCtx& X, const int tcol /* global tile column */, const bool last_tile) {
  lds_f64* const Ein = (lds_f64*)hpx_stage0;                  // slot 0 of Xs: re 256 | im 256
  lds_f64* const Vs = (lds_f64*)(hpx_stage1 + FV_OFF);
  lds_cplx* const Dm = (lds_cplx*)(hpx_stage1 + FD_OFF);
  lds_cplx* const Ym = Dm + 16 * 17;
  lds_f64* const dg = (lds_f64*)(Ym + 16 * 17);
  const int tid = opaque(X.tid), q = tid & 15, ib = tid >> 4;
  const bool dia = (q == ib), low = (q < ib);
  double dr = Ein[ib * 16 + q], di = Ein[256 + ib * 16 + q];
  if (!low) Dm[ib * 17 + q] = (cplx){0.0, 0.0};               // diagonal and above stay zero in LDS
  double yr = dia ? 1.0 : 0.0, yi = 0.0;
  __builtin_amdgcn_s_setprio(2);
#ifdef HPX_DBG_F_NOELIM
  if (dia) dg[ib] = dr;
  for (int k = 0; k < 0; ++k) {
#else
#pragma unroll
  for (int k = 0; k < 16; ++k) {
#endif
    if (dia) dg[ib] = dr; else if (low) Dm[ib * 17 + q] = (cplx){dr, di};
    Ym[ib * 17 + q] = (cplx){yr, yi};
    lds_barrier();
    const double dkk = dg[k];
    const cplx c = Dm[ib * 17 + k], cq = Dm[q * 17 + k], sy = Ym[k * 17 + q];
    const double r0 = __builtin_amdgcn_rcp(dkk);
    const double rinv = fma(r0, fma(-dkk, r0, 1.0), r0);
    const double lr = c.x * rinv, lm = c.y * rinv;
    yr = fma(-lr, sy.x, yr);
    yr = fma(lm, sy.y, yr);
    yi = fma(-lr, sy.y, yi);
    yi = fma(-lm, sy.x, yi);
    dr = fma(-lr, cq.x, dr);
    dr = fma(-lm, cq.y, dr);
    di = fma(-lm, cq.x, di);
    di = fma(lr, cq.y, di);
  }
  lds_barrier();
  __builtin_amdgcn_s_setprio(0);
  bool bad = false;
  double wr = 0.0, wi = 0.0;
  if (q <= ib) {
    const double pq = dg[q], pib = dg[ib];
    if (!(pq > 0.0) || !(pib > 0.0)) bad = true;
    const double sq = rsqrt_nr(pq);
    const long off = HPX_LIDX(tcol * 16 + ib, tcol * 16 + q, X.npad);
    X.Lb[off] = dr * sq;
    X.Lb[off + 16] = dia ? 0.0 : di * sq;
    const double sv = rsqrt_nr(pib);
    wr = yr * sv;
    wi = yi * sv;
  }
  // inv(L)[ib][q]: tile layout = column q, row ib
  double* vt = X.Vt + (long)tcol * 512 + q * 32 + ib;
  vt[0] = wr;
  vt[16] = wi;
  Vs[q * 32 + ib + 16 * (q & 1)] = wr;
  Vs[q * 32 + ib + 16 * (1 - (q & 1))] = wi;
  const int o = 16 * (tcol & 1);
  double* wgr = X.Wgre + (long)(tcol >> 1) * 1024;
  double* wgi = X.Wgim + (long)(tcol >> 1) * 1024;
  wgr[(o + ib) * 32 + o + q] = wr;
  wgi[(o + ib) * 32 + o + q] = wi;
  if (o == 0) {
    wgr[ib * 32 + 16 + q] = 0.0;                               // upper-right block of the inverse is zero
    wgi[ib * 32 + 16 + q] = 0.0;
    if (last_tile) {                                           // 16-wide last block: nothing below either
      wgr[(16 + ib) * 32 + q] = 0.0; wgi[(16 + ib) * 32 + q] = 0.0;
      wgr[(16 + ib) * 32 + 16 + q] = 0.0; wgi[(16 + ib) * 32 + 16 + q] = 0.0;
    }
  }
  lds_barrier();
  return bad;
}
// lane K (of each group of 16 lanes) to all 16 lanes of its group: DPP row_newbcast (gfx90a+), two 32-bit moves
template <int K>
__device__ HPX_INL double bcast16(const double x) {
  const long long raw = __double_as_longlong(x);
  // (every lane has a source lane: no "old" value to initialise)
  const int lo = __builtin_amdgcn_mov_dpp((int)(raw & 0xffffffffll), 0x150 + K, 0xf, 0xf, true);
  const int hi = __builtin_amdgcn_mov_dpp((int)(raw >> 32), 0x150 + K, 0xf, 0xf, true);
  return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}
// one lane's value as a wave-uniform one (v_readlane_b32 on the two halves; `lane` a constant after unrolling)
__device__ HPX_INL double lane_value(const double x, const int lane) {
  const long long raw = __double_as_longlong(x);
  const int lo = __builtin_amdgcn_readlane((int)(raw & 0xffffffffll), lane);
  const int hi = __builtin_amdgcn_readlane((int)(raw >> 32), lane);
  return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}
// (k is a constant after the unrolling of the loops that call this: the switch folds to one case)
__device__ HPX_INL double bcast16_of(const double x, const int k) {
  switch (k) {
    case 0: return bcast16<0>(x);   case 1: return bcast16<1>(x);   case 2: return bcast16<2>(x);   case 3: return bcast16<3>(x);
    case 4: return bcast16<4>(x);   case 5: return bcast16<5>(x);   case 6: return bcast16<6>(x);   case 7: return bcast16<7>(x);
    case 8: return bcast16<8>(x);   case 9: return bcast16<9>(x);   case 10: return bcast16<10>(x); case 11: return bcast16<11>(x);
    case 12: return bcast16<12>(x); case 13: return bcast16<13>(x); case 14: return bcast16<14>(x); default: return bcast16<15>(x);
  }
}
// ---- the same elimination on ONE wave with the tile in registers: lane (li, g) holds row li, columns g + 4 v -- the
// accumulator layout of a D^T tile, so the wave that owns the diagonal tile eliminates it where it lies.  No
// workgroup barrier inside (elim16 has eighteen, and spends most of its time in them): per step the pivot column
// and row k of the inverse go through LDS -- a wave's LDS operations execute in order, the reads behind the writes
// see them -- and everything else stays in the lane.  Same operations in the same order per element as elim16
// (the pivots' reciprocals included: see the loop).
// out: as elim16; the CALLER puts a workgroup barrier between this and the first use of Vs by another wave.
// `flag` (split form; else null): raised -- behind `publish()` -- as soon as the inverse tile is in Vt, the one thing
// another workgroup waits for; the factor's own tile and the W blocks follow.
// `Vs`: where the LDS copy of the inverse goes (the wide form's look-ahead keeps two: the next column's inverse is
// written while the other waves still read the current one).
template <class Publish>
__device__ HPX_INL bool elim16w(const WideCtx& X, const int tcol, const bool last_tile, const d4 re, const d4 im,
                             Publish publish, lds_f64* const Vs) {
  lds_cplx* const col = (lds_cplx*)(hpx_stage1 + FD_OFF);      // [4 lane groups][16 rows]: column k below the pivot
  lds_cplx* const yrw = col + 64;                               // [4 lane groups][4]: row k of the inverse
  lds_f64* const raw = (lds_f64*)(yrw + 16);                    // [4 lane groups][16]: the column's real parts, unmasked
  lds_f64* const dgs = raw + 64;                                // [16]: the pivots
  const int lane = opaque(X.lane), li = lane & 15, g = lane >> 4;
  double dr[4], di[4], yr[4], yi[4];
#pragma unroll
  for (int v = 0; v < 4; ++v) {
    dr[v] = re[v];
    di[v] = im[v];
    yr[v] = (li == g + 4 * v) ? 1.0 : 0.0;
    yi[v] = 0.0;
  }
  // Software-pipelined: the recurrence of the factorisation is  column k+1 after step k -> LDS -> everyone's operands
  // of step k+1, so per step ONLY  l = c / d_kk,  the 16 updates of D  and the LDS round trip are on the critical
  // path.  Off it, in the shadow of that round trip: (a) the next pivot's reciprocal -- every lane forms
  // d_{k+1,k+1} - l_{k+1,k} conj(D_{k+1,k}) itself, with the very operations its owner applies (same bits), from one
  // broadcast read of the column and of the diagonal entry before the step; (b) the inverse's updates, one step behind
  // (row k of Y is final after the step k-1 update, broadcast to all lanes and used one step later).
  // Round 6: only the COLUMN goes through LDS now.  Row k of the inverse sits in lane li = k of every group of 16 lanes,
  // register v <-> column g + 4 v -- a DPP row broadcast (row_newbcast:k), no masked LDS write / read-back; the two
  // single entries the next pivot needs come out of their lanes with v_readlane.  Stand-alone 3.69 -> 3.14 us
  // (tools/experiments/elim/elim16w_probe.hip), k_factor_split at C2 0.142 -> 0.136 ms.  Same operations per element.
  lds_f64* const rawn = raw;       // [4 lane groups][16]: real parts of the columns 4 (k+1 >> 2) + g BEFORE step k
#define HPX_E16_FENCE()                                   \
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");  \
  __builtin_amdgcn_wave_barrier()
  __builtin_amdgcn_s_setprio(2);
  col[g * 16 + li] = (cplx){li > 0 ? dr[0] : 0.0, li > 0 ? di[0] : 0.0};
  rawn[g * 16 + li] = dr[0];
  HPX_E16_FENCE();
  double dkk = rawn[0];                                 // D[0][0]
  cplx c = col[li], cq[4], cn = col[1];
  double dn = rawn[16 + 1];                             // D[1][1] before step 0
#pragma unroll
  for (int v = 0; v < 4; ++v) cq[v] = col[g + 4 * v];
  dgs[0] = dkk;
  double rinv;
  {
    const double r0 = __builtin_amdgcn_rcp(dkk);
    rinv = fma(r0, fma(-dkk, r0, 1.0), r0);
  }
  double plr = 0.0, plm = 0.0;                          // the multipliers of the step before (for the lagging Y update)
  cplx sy[4];
#pragma unroll
  for (int v = 0; v < 4; ++v) sy[v] = (cplx){0.0, 0.0};
#pragma unroll
  for (int k = 0; k < 16; ++k) {
    // ---- A (critical): multipliers, D update, the next column out, the next operands requested
    const double lr = c.x * rinv, lm = c.y * rinv;
    // (register v holds the columns g + 4 v: those with 4 v + 3 <= k are finished -- their operand cq is the masked
    // zero -- and are skipped at compile time; likewise below, row k of the inverse is zero right of column k)
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      if (4 * v + 3 <= k) continue;
      dr[v] = fma(-lr, cq[v].x, dr[v]);
      dr[v] = fma(-lm, cq[v].y, dr[v]);
      di[v] = fma(-lm, cq[v].x, di[v]);
      di[v] = fma(lr, cq[v].y, di[v]);
      // (computed HERE: left alone, the compiler sinks the update chains of the elements that are only stored at the
      // end into that final block and keeps every step's operands alive for it -- in scratch)
      asm volatile("" : "+v"(dr[v]), "+v"(di[v]));
    }
    const cplx cn_k = cn;
    const double dn_k = dn, rinv_k = rinv;
    if (k < 15) {
      const int k1 = k + 1, kv1 = k1 >> 2, kg1 = k1 & 3;
      const bool below = li > k1;
      col[g * 16 + li] = (cplx){below ? dr[kv1] : 0.0, below ? di[kv1] : 0.0};
      HPX_E16_FENCE();
      __builtin_amdgcn_sched_barrier(0);
      c = col[kg1 * 16 + li];
#pragma unroll
      for (int v = 0; v < 4; ++v)
        if (4 * v + 3 > k1) cq[v] = col[kg1 * 16 + g + 4 * v];
      if (k1 < 15) {
        // the two values the NEXT pivot needs are single entries: straight out of their lane's registers (v_readlane,
        // wave-uniform), not through LDS -- D[k1+1][k1] (lane (k1+1, kg1), register kv1) and D[k1+1][k1+1] as it is now
        cn.x = lane_value(dr[kv1], kg1 * 16 + k1 + 1);
        cn.y = lane_value(di[kv1], kg1 * 16 + k1 + 1);
        dn = lane_value(dr[(k1 + 1) >> 2], ((k1 + 1) & 3) * 16 + k1 + 1);
      }
      __builtin_amdgcn_sched_barrier(0);
      // ---- in the shadow of that round trip: the next pivot (the owner's own operations on D[k+1][k+1]) ...
      const double lrn = cn_k.x * rinv_k, lmn = cn_k.y * rinv_k;
      double dkn = fma(-lrn, cn_k.x, dn_k);
      dkn = fma(-lmn, cn_k.y, dkn);
      dgs[k1] = dkn;
      const double r0 = __builtin_amdgcn_rcp(dkn);
      rinv = fma(r0, fma(-dkn, r0, 1.0), r0);
    }
    // ---- ... the inverse's update of the step before ...
    if (k > 0) {
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        if (4 * v > k - 1) continue;                   // row k - 1 of the inverse is zero right of column k - 1
        yr[v] = fma(-plr, sy[v].x, yr[v]);
        yr[v] = fma(plm, sy[v].y, yr[v]);
        yi[v] = fma(-plr, sy[v].y, yi[v]);
        yi[v] = fma(-plm, sy[v].x, yi[v]);
        asm volatile("" : "+v"(yr[v]), "+v"(yi[v]));
      }
    }
    // ---- ... and row k of the inverse (final since that update) to every lane: the row's entries for the columns
    // g + 4 v sit in lane li = k of the lane's own group of 16, register v -- a DPP row broadcast (row_newbcast:k), no LDS
    // round trip and no masked write
#pragma unroll
    for (int v = 0; v < 4; ++v)
      if (4 * v <= k) {
        sy[v].x = bcast16_of(yr[v], k);
        sy[v].y = bcast16_of(yi[v], k);
      }
    plr = lr;
    plm = lm;
    __builtin_amdgcn_sched_barrier(0);
  }
#pragma unroll
  for (int v = 0; v < 4; ++v) {                          // the last step's update of the inverse (all zero multipliers:
    yr[v] = fma(-plr, sy[v].x, yr[v]);                   // no row lies below row 15 -- kept for the form)
    yr[v] = fma(plm, sy[v].y, yr[v]);
    yi[v] = fma(-plr, sy[v].y, yi[v]);
    yi[v] = fma(-plm, sy[v].x, yi[v]);
  }
  HPX_E16_FENCE();
#undef HPX_E16_FENCE
  __builtin_amdgcn_s_setprio(0);
  bool bad = false;
  const double pib = dgs[li];
  const double sv = rsqrt_nr(pib);
  double wr[4], wi[4], sq[4];
#pragma unroll
  for (int v = 0; v < 4; ++v) {
    const int q = g + 4 * v;                                   // column; row li
    wr[v] = 0.0;
    wi[v] = 0.0;
    sq[v] = 0.0;
    if (q <= li) {
      const double pq = dgs[q];
      if (!(pq > 0.0) || !(pib > 0.0)) bad = true;
      sq[v] = rsqrt_nr(pq);
      wr[v] = yr[v] * sv;
      wi[v] = yi[v] * sv;
    }
    double* vt = X.Vt + (long)tcol * 512 + q * 32 + li;        // inv(L)[li][q]: tile layout = column q, row li
    vt[0] = wr[v];
    vt[16] = wi[v];
    Vs[q * 32 + li + 16 * (q & 1)] = wr[v];
    Vs[q * 32 + li + 16 * (1 - (q & 1))] = wi[v];
  }
  publish();
  const int o = 16 * (tcol & 1);
  double* wgr = X.Wgre + (long)(tcol >> 1) * 1024;
  double* wgi = X.Wgim + (long)(tcol >> 1) * 1024;
#pragma unroll
  for (int v = 0; v < 4; ++v) {
    const int q = g + 4 * v;
    if (q <= li) {
      const long off = HPX_LIDX(tcol * 16 + li, tcol * 16 + q, X.npad);
      X.Lb[off] = dr[v] * sq[v];
      X.Lb[off + 16] = (q == li) ? 0.0 : di[v] * sq[v];
    }
    wgr[(o + li) * 32 + o + q] = wr[v];
    wgi[(o + li) * 32 + o + q] = wi[v];
    if (o == 0) {
      wgr[li * 32 + 16 + q] = 0.0;                             // upper-right block of the inverse is zero
      wgi[li * 32 + 16 + q] = 0.0;
      if (last_tile) {                                         // 16-wide last block: nothing below either
        wgr[(16 + li) * 32 + q] = 0.0; wgi[(16 + li) * 32 + q] = 0.0;
        wgr[(16 + li) * 32 + 16 + q] = 0.0; wgi[(16 + li) * 32 + 16 + q] = 0.0;
      }
    }
  }
  return bad;
}
template <class Publish>
__device__ HPX_INL bool elim16w(const WideCtx& X, const int tcol, const bool last_tile, const d4 re, const d4 im,
                             Publish publish) {
  return elim16w(X, tcol, last_tile, re, im, publish, (lds_f64*)(hpx_stage1 + FV_OFF));
}
__device__ HPX_INL bool elim16w(const WideCtx& X, const int tcol, const bool last_tile, const d4 re, const d4 im) {
  return elim16w(X, tcol, last_tile, re, im, [] {});
}
}  // namespace
