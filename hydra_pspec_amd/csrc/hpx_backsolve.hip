// Backward substitution L^H X = Z with the whole solution block in registers (orders up to 640).
//
// k_backsolve of hpx_factor.hip reads every tile of L once but re-reads the solution X once per 128-column
// super-block and runs at two 4-wave workgroups per CU; at C3 it moved 3.9 GB per launch for 2.8 GB of
// algorithmic traffic and took 0.85 ms (docs/HISTORY.md sections 9.4, 10.9).  Here ONE workgroup of eight waves owns a
// baseline and keeps all of X -- (n/16) row tiles x the t-tiles of a pass, as f64 MFMA accumulators -- in its
// registers, right-looking from the last tile row to the first:
//
//   step J:  X_J = inv(L_JJ)^H acc_J   (the wave that owns row tile J; published through LDS and stored)
//            acc_I -= L[J, I]^H X_J    for every row tile I < J, on the wave that owns I
//
// so that L is read exactly once (tile (J, I) by the owner of I, 32 bytes per lane and load), Z once and X is
// only written.  Row tiles are dealt to the waves boustrophedon (tile 8 q + p to wave p for even q, 7 - p for odd
// q): every wave gets the same number of tile updates.  The finalisation of X_{J-1} is taken off the critical
// path: its owner updates that tile first, finalises and publishes it, then does its other updates, so the next
// step's operand is in LDS (two slots) when the barrier opens.  Each wave re-loads its L operand registers for
// step J - 1 as soon as step J has used them (a whole step of latency cover).
//
// inv(L_JJ) is read from the diagonal 16 x 16 sub-blocks of W (the 32 x 32 inverse blocks both factor kernels
// write).  Complex products: up to two row tiles per wave (17 tile rows, N = 256) from THREE real MFMAs on three
// accumulators per tile (24 instead of 16 registers, 12 instead of 16 MFMAs per tile and step: 0.308 against 0.333 ms
// for 1024 baselines at N = 256); with more row tiles per wave the third accumulator no longer fits 256 registers
// (54 .. 62 spilled) and the four-product form on (re, im) accumulators stays.  The whole register file of a CU
// (512 KB) cannot hold 33 row tiles x 2 t-tiles x 3 accumulators (396 KB) next to one set of L operands (132 KB).
#include "hpx_internal.h"
#include <type_traits>

#define HPX_INL __forceinline__

// (The register-resident form beyond 17 tile rows -- one t-tile per workgroup, three to five slots per wave -- measured
// 1.19 ms against 0.84 ms for k_backsolve at C3 and is not part of the product: tools/experiments/backsolve_reg33.patch.)

namespace {

typedef __attribute__((address_space(3))) double lds_f64;

__device__ HPX_INL void lds_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}
// (uniform base) + (32-bit lane offset in bytes): the scalar-base addressing form, one offset register for all
// accesses instead of a 64-bit address pair each (which, hoisted out of the step loop, filled the register file)
template <typename T>
__device__ HPX_INL const T* lane_ptr(const double* ubase, const unsigned lane_bytes) {
  return reinterpret_cast<const T*>(reinterpret_cast<const char*>(ubase) + lane_bytes);
}
__device__ HPX_INL double* lane_ptr_w(double* ubase, const unsigned lane_bytes) {
  return reinterpret_cast<double*>(reinterpret_cast<char*>(ubase) + lane_bytes);
}
// row tile of block q owned by `wave`
__device__ HPX_INL int tile_of(const int q, const int wave) { return 8 * q + ((q & 1) ? 7 - wave : wave); }
__device__ HPX_INL int owner_of(const int I) { return ((I >> 3) & 1) ? 7 - (I & 7) : (I & 7); }

// NS: row tiles (accumulator + L operand slots) per wave, ceil((n/16 - 1) / 8): the last tile row has nothing
// below it and is finalised straight from Z when it is alone in its block of eight (n/16 = 8 m + 1, every
// BASELINE shape), without a slot; NT: t-tiles of this pass
// DEPTH: steps the L operands are requested ahead (a ring of DEPTH register sets, the loop over the tile rows unrolled
// by DEPTH so that every set is a compile-time register range).  One step ahead leaves a step's worth of time --
// under a microsecond at small orders -- to cover a memory round trip of two to three: the chain of dependent steps
// then runs at the memory latency (config 2: 52 us for 16 steps).  Four steps ahead where the registers allow it.
// M3: complex products from THREE real ones (see bs_tile_pass_ul, whose results this form then reproduces bit for
// bit -- a baseline gives the same chain alone and inside a large batch): ar = Re z - sum lr xr, ai = sum lm xi,
// a3 = (Re z + Im z) - sum (lr - lm)(xr + xi); the accumulator is (ar - ai) + i (a3 - ar - ai).
template <int NS, int NT, int DEPTH, bool M3>
__device__ HPX_INL void bs_reg_pass(const double* __restrict__ Lre, const double* __restrict__ Wgre,
                                    const double* __restrict__ Wgim, double* __restrict__ Xre,
                                    double* __restrict__ Xim, double* xs, const int npad, const int TP, const int t0,
                                    const int wave, const int lane) {
  const int nct = npad >> 4;
  const int li = lane & 15, g = lane >> 4;
  constexpr int NL = NS;
  const unsigned lz = 8u * (g * 32 + li);          // lane offsets (bytes): Z / X tiles, L operand, inverse tile
  const unsigned ll = 8u * (li * 32 + 4 * g);
  const unsigned lw = 8u * (g * 32 + li);
  d4 ar[NS > 0 ? NS : 1][NT], ai[NS > 0 ? NS : 1][NT], a3[NS > 0 ? NS : 1][NT];
  // ---- Z[c][t] = conj(Laug[npad + t][c]) as acc[m = c][n = t]
#pragma unroll
  for (int q = 0; q < NS; ++q) {
    const int I = tile_of(q, wave);
#pragma unroll
    for (int tt = 0; tt < NT; ++tt)
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        double zr = 0.0, zi = 0.0;
        if (I < nct) {      // row npad + t0 + 16 tt + li (a tile row of its own), column 16 I + g + 4 v
          const double* zb = Lre + (((long)((npad + t0) >> 4) + tt) * npad + 16 * I + 4 * v) * 32;
          zr = *lane_ptr<double>(zb, lz);
          zi = -*lane_ptr<double>(zb + 16, lz);
        }
        ar[q][tt][v] = zr;
        ai[q][tt][v] = M3 ? 0.0 : zi;
        a3[q][tt][v] = zr + zi;
      }
  }
  // ---- L operand of a step: rows 4 g .. 4 g + 3 (k index (g, s)) of column li of tile (J, I)
  double lr[DEPTH][NL > 0 ? NL : 1][4], lm[DEPTH][NL > 0 ? NL : 1][4];
#define HPX_BS_LOADL(S_, q_, J_)                                                         \
  {                                                                                      \
    const int I_ = tile_of(q_, wave);                                                    \
    if (I_ < (J_)) {                                                                     \
      const double* lb_ = Lre + ((long)(J_) * npad + 16 * I_) * 32;                      \
      const double2 a0_ = *lane_ptr<double2>(lb_, ll);                                   \
      const double2 a1_ = *lane_ptr<double2>(lb_ + 2, ll);                               \
      const double2 b0_ = *lane_ptr<double2>(lb_ + 16, ll);                              \
      const double2 b1_ = *lane_ptr<double2>(lb_ + 18, ll);                              \
      lr[S_][q_][0] = a0_.x; lr[S_][q_][1] = a0_.y; lr[S_][q_][2] = a1_.x; lr[S_][q_][3] = a1_.y;   \
      lm[S_][q_][0] = b0_.x; lm[S_][q_][1] = b0_.y; lm[S_][q_][2] = b1_.x; lm[S_][q_][3] = b1_.y;   \
    }                                                                                    \
  }
  // acc_I -= L[J, I]^H X_J with X_J from LDS slot `sl`:  conj(l) x = (lr xr + lm xi) + i (lr xi - lm xr)
#define HPX_BS_UPDATE(S_, q_, sl_)                                                       \
  {                                                                                      \
    const lds_f64* xb_ = (const lds_f64*)(xs + (sl_) * (NT * 512));                      \
    _Pragma("unroll") for (int s = 0; s < 4; ++s) {                                      \
      __builtin_amdgcn_sched_barrier(0);                                                 \
      const double nlr_ = -lr[S_][q_][s], nlm_ = -lm[S_][q_][s], plm_ = lm[S_][q_][s];   \
      const double dl_ = lm[S_][q_][s] - lr[S_][q_][s];                                  \
      _Pragma("unroll") for (int tt = 0; tt < NT; ++tt) {                                \
        const double xr_ = xb_[tt * 512 + (4 * g + s) * 16 + li];                        \
        const double xi_ = xb_[tt * 512 + 256 + (4 * g + s) * 16 + li];                  \
        if (M3) {                                                                        \
          ar[q_][tt] = mfma64(nlr_, xr_, ar[q_][tt]);                                    \
          ai[q_][tt] = mfma64(plm_, xi_, ai[q_][tt]);                                    \
          a3[q_][tt] = mfma64(dl_, xr_ + xi_, a3[q_][tt]);                               \
        } else {                                                                         \
          ar[q_][tt] = mfma64(nlr_, xr_, ar[q_][tt]);                                    \
          ar[q_][tt] = mfma64(nlm_, xi_, ar[q_][tt]);                                    \
          ai[q_][tt] = mfma64(nlr_, xi_, ai[q_][tt]);                                    \
          ai[q_][tt] = mfma64(plm_, xr_, ai[q_][tt]);                                    \
        }                                                                                \
      }                                                                                  \
    }                                                                                    \
  }
  // X_J = inv(L_JJ)^H acc_J: stored, and published in LDS slot J & 1 as [t-tile][re 16 x 16 | im 16 x 16], row-major
  // inv(L_JJ) for the tile this wave finalises NEXT, fetched a whole round of steps ahead (its latency would
  // otherwise sit on the critical path of every step): inv(L)[c' = 4 s + g][c = li]
  double w_r[4], w_i[4];
#define HPX_BS_LOADW(J_)                                                                 \
  if ((J_) >= 0) {                                                                       \
    const double* wr_ = Wgre + (long)((J_) >> 1) * 1024 + (16 * ((J_) & 1)) * 33;        \
    const double* wi_ = Wgim + (long)((J_) >> 1) * 1024 + (16 * ((J_) & 1)) * 33;        \
    _Pragma("unroll") for (int s = 0; s < 4; ++s) {                                      \
      w_r[s] = *lane_ptr<double>(wr_ + (4 * s) * 32, lw);                                \
      w_i[s] = *lane_ptr<double>(wi_ + (4 * s) * 32, lw);                                \
    }                                                                                    \
  }
#define HPX_BS_FINAL_OF(J_)                                                              \
  {                                                                                      \
    double* xb_ = xs + ((J_) & 1) * (NT * 512);                                          \
    _Pragma("unroll") for (int tt = 0; tt < NT; ++tt) {                                  \
      d4 xr_ = {0., 0., 0., 0.}, xi_ = {0., 0., 0., 0.};                                 \
      const d4 cr_ = HPX_BS_ACC_R(tt), ci_ = HPX_BS_ACC_I(tt);                           \
      if (M3) {                                                                          \
        const d4 cs_ = cr_ + ci_;                                                        \
        d4 p3_ = {0., 0., 0., 0.};                                                       \
        _Pragma("unroll") for (int s = 0; s < 4; ++s) {                                  \
          xr_ = mfma64(w_r[s], cr_[s], xr_);                                             \
          xi_ = mfma64(w_i[s], ci_[s], xi_);                                             \
          p3_ = mfma64(w_r[s] - w_i[s], cs_[s], p3_);                                    \
        }                                                                                \
        const d4 p1_ = xr_;                                                              \
        xr_ = p1_ + xi_;                                                                 \
        xi_ = (p3_ - p1_) + xi_;                                                         \
      } else {                                                                           \
        _Pragma("unroll") for (int s = 0; s < 4; ++s) {                                  \
          xr_ = mfma64(w_r[s], cr_[s], xr_);                                             \
          xr_ = mfma64(w_i[s], ci_[s], xr_);                                             \
          xi_ = mfma64(w_r[s], ci_[s], xi_);                                             \
          xi_ = mfma64(-w_i[s], cr_[s], xi_);                                            \
        }                                                                                \
      }                                                                                  \
      _Pragma("unroll") for (int v = 0; v < 4; ++v) {                                    \
        const int c_ = HPX_ACC_ROW(g, v);                                                \
        xb_[tt * 512 + c_ * 16 + li] = xr_[v];                                           \
        xb_[tt * 512 + 256 + c_ * 16 + li] = xi_[v];                                     \
        const long xo_ = (long)(16 * (J_) + 4 * v) * TP + t0 + 16 * tt;                  \
        *lane_ptr_w(Xre + xo_, 8u * (g * TP + li)) = xr_[v];                             \
        *lane_ptr_w(Xim + xo_, 8u * (g * TP + li)) = xi_[v];                             \
      }                                                                                  \
    }                                                                                    \
  }
  const int Jlast = nct - 1;
  // the highest tile this wave owns (it finalises its tiles from there downwards, one per block of eight)
  int mynext = tile_of((Jlast >> 3), wave);
  if (mynext > Jlast) mynext = tile_of((Jlast >> 3) - 1, wave);       // (-1 .. : tile_of of a negative block is < 0)
  HPX_BS_LOADW(mynext)
#pragma unroll
  for (int d = 0; d < DEPTH; ++d)
#pragma unroll
    for (int q = 0; q < NL; ++q) HPX_BS_LOADL(d, q, Jlast - d)        // (rows below 1 are never used: the guard I < J)
  if (wave == owner_of(Jlast)) {
    if ((Jlast >> 3) >= NS) {                 // alone in its block: no slot, Z straight from memory
      d4 zr[NT], zi[NT];
#pragma unroll
      for (int tt = 0; tt < NT; ++tt)
#pragma unroll
        for (int v = 0; v < 4; ++v) {
          const double* zb = Lre + (((long)((npad + t0) >> 4) + tt) * npad + 16 * Jlast + 4 * v) * 32;
          zr[tt][v] = *lane_ptr<double>(zb, lz);
          zi[tt][v] = -*lane_ptr<double>(zb + 16, lz);
        }
#define HPX_BS_ACC_R(tt_) zr[tt_]
#define HPX_BS_ACC_I(tt_) zi[tt_]
      HPX_BS_FINAL_OF(Jlast)
#undef HPX_BS_ACC_R
#undef HPX_BS_ACC_I
    } else {
#pragma unroll
      for (int q = 0; q < NS; ++q)
        if (q == (Jlast >> 3)) {
#define HPX_BS_ACC_R(tt_) (M3 ? ar[q][tt_] - ai[q][tt_] : ar[q][tt_])
#define HPX_BS_ACC_I(tt_) (M3 ? (a3[q][tt_] - ar[q][tt_]) - ai[q][tt_] : ai[q][tt_])
          HPX_BS_FINAL_OF(Jlast)
#undef HPX_BS_ACC_R
#undef HPX_BS_ACC_I
        }
    }
  }
  if (wave == owner_of(Jlast)) {               // its next tile: one block further down
    mynext = tile_of((Jlast >> 3) - 1, wave);
    HPX_BS_LOADW(mynext)
  }
  // one step: X_J is in its slot; operand set S holds L[J, .]; it is refilled with L[J - DEPTH, .] after use
#define HPX_BS_STEP(S_, J_)                                                                            \
  if ((J_) >= 1) {                                                                                     \
    const int J = (J_);                                                                                \
    lds_barrier();                              /* X_J is in its slot; the other slot is free again */ \
    const int sl = J & 1;                                                                              \
    const int qn = (J - 1) >> 3;                /* the next tile to finalise lives in this slot of its owner */ \
    const bool next_mine = (wave == owner_of(J - 1));                                                  \
    if (next_mine) {                                                                                   \
      _Pragma("unroll") for (int q = 0; q < NS; ++q)                                                   \
        if (q == qn) {                                                                                 \
          HPX_BS_UPDATE(S_, q, sl)                                                                     \
          HPX_BS_FINAL_OF(J - 1)                                                                       \
          mynext = tile_of(q - 1, wave);                                                               \
          HPX_BS_LOADW(mynext)                                                                         \
          HPX_BS_LOADL(S_, q, J - DEPTH)                                                               \
        }                                                                                              \
    }                                                                                                  \
    /* the other tiles, k-step by k-step with the X operand of one k-step in registers at a time */   \
    {                                                                                                  \
      const lds_f64* xb = (const lds_f64*)(xs + sl * (NT * 512));                                      \
      _Pragma("unroll") for (int s = 0; s < 4; ++s) {                                                  \
        __builtin_amdgcn_sched_barrier(0);                                                             \
        double xr[NT], xi[NT], xq[NT];                                                                 \
        _Pragma("unroll") for (int tt = 0; tt < NT; ++tt) {                                            \
          xr[tt] = xb[tt * 512 + (4 * g + s) * 16 + li];                                               \
          xi[tt] = xb[tt * 512 + 256 + (4 * g + s) * 16 + li];                                         \
          xq[tt] = xr[tt] + xi[tt];                                                                    \
        }                                                                                              \
        _Pragma("unroll") for (int q = 0; q < NL; ++q)                                                 \
          if (!(next_mine && q == qn) && tile_of(q, wave) < J) {                                       \
            const double nlr = -lr[S_][q][s], nlm = -lm[S_][q][s], plm = lm[S_][q][s];                 \
            const double dl = lm[S_][q][s] - lr[S_][q][s];                                             \
            _Pragma("unroll") for (int tt = 0; tt < NT; ++tt) {                                        \
              if (M3) {                                                                                \
                ar[q][tt] = mfma64(nlr, xr[tt], ar[q][tt]);                                            \
                ai[q][tt] = mfma64(plm, xi[tt], ai[q][tt]);                                            \
                a3[q][tt] = mfma64(dl, xq[tt], a3[q][tt]);                                             \
              } else {                                                                                 \
                ar[q][tt] = mfma64(nlr, xr[tt], ar[q][tt]);                                            \
                ar[q][tt] = mfma64(nlm, xi[tt], ar[q][tt]);                                            \
                ai[q][tt] = mfma64(nlr, xi[tt], ai[q][tt]);                                            \
                ai[q][tt] = mfma64(plm, xr[tt], ai[q][tt]);                                            \
              }                                                                                        \
            }                                                                                          \
          }                                                                                            \
      }                                                                                                \
      __builtin_amdgcn_sched_barrier(0);                                                               \
      _Pragma("unroll") for (int q = 0; q < NL; ++q)                                                   \
        if (!(next_mine && q == qn) && tile_of(q, wave) < J) HPX_BS_LOADL(S_, q, J - DEPTH)            \
    }                                                                                                  \
  }
#define HPX_BS_ACC_R(tt_) (M3 ? ar[q][tt_] - ai[q][tt_] : ar[q][tt_])
#define HPX_BS_ACC_I(tt_) (M3 ? (a3[q][tt_] - ar[q][tt_]) - ai[q][tt_] : ai[q][tt_])
  for (int J0 = Jlast; J0 >= 1; J0 -= DEPTH) {
    HPX_BS_STEP(0, J0)
    if (DEPTH > 1) HPX_BS_STEP(1 % DEPTH, J0 - 1)
    if (DEPTH > 2) HPX_BS_STEP(2 % DEPTH, J0 - 2)
    if (DEPTH > 3) HPX_BS_STEP(3 % DEPTH, J0 - 3)
  }
#undef HPX_BS_ACC_R
#undef HPX_BS_ACC_I
#undef HPX_BS_STEP
#undef HPX_BS_LOADL
#undef HPX_BS_UPDATE
#undef HPX_BS_FINAL_OF
#undef HPX_BS_LOADW
}

// ---- one t-tile per workgroup with HAND-COUNTED operand waits (the small-batch form) --------------------------------
// bs_reg_pass's operand loads sit under data-dependent conditions (tile below the current row?  this wave's turn to
// finalise?), so the compiler cannot count what is in flight: after every such branch it waits with vmcnt(0) -- for the
// operands it has just requested for later steps too.  Every step then costs a memory round trip, whatever the
// prefetch distance (config 2: 3.4 us per step for 1 us of dependent work).  Here every wave issues the SAME loads
// in every step (clamped addresses where a tile does not exist or is not needed: tiles above the diagonal are read
// and dropped -- up to twice the factor's bytes, which is why only the small batches whose chain of steps is what
// takes the time come here), through asm statements the compiler does not see, and the waits are counted by hand:
// before step J uses operand set S, all but the (DEPTH - 1) NS * 4 youngest loads are complete.  The stores of a
// finalised tile (compiler-issued, on the owner's path only) count on the same counter: they only make a wait
// stricter.  The inverse diagonal tiles of all the wave's row tiles are fetched once, up front.
#ifndef HPX_BS_M3_MAXNS
#define HPX_BS_M3_MAXNS 2          // slots per wave up to which the three-product form is used
#endif
#ifndef HPX_BS_DEPTH2W
#define HPX_BS_DEPTH2W 1            // operand sets ahead, two slots x two t-tiles with three accumulators each (two spill)
#endif
#define HPX_BS_M3(NS_) ((NS_) <= HPX_BS_M3_MAXNS)
typedef double bs_d2 __attribute__((ext_vector_type(2)));
__device__ HPX_INL bs_d2 bs_ld16(const double* ubase, const unsigned lane_bytes) {
  bs_d2 r;
  asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(r) : "v"(lane_bytes), "s"(ubase) : "memory");
  return r;
}
template <int N>
__device__ HPX_INL void bs_wait_vm() {
  asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N < 63 ? N : 63) : "memory");
}
template <int NS, int DEPTH>
__device__ HPX_INL void bs_tile_pass_ul(const double* __restrict__ Lre, const double* __restrict__ Wgre,
                                        const double* __restrict__ Wgim, double* __restrict__ Xre,
                                        double* __restrict__ Xim, double* xs, const int npad, const int TP, const int t0,
                                        const int wave, const int lane) {
  static_assert(NS >= 1 && DEPTH >= 1 && (DEPTH - 1) * NS * 4 <= 60, "operand ring out of range");
  const int nct = npad >> 4, Jlast = nct - 1;
  const int li = lane & 15, g = lane >> 4;
  const unsigned lz = 8u * (g * 32 + li);          // lane offsets (bytes): Z / X tiles, L operand, inverse tile
  const unsigned ll = 8u * (li * 32 + 4 * g);
  const unsigned lw = 8u * (g * 32 + li);
  int tl[NS];                                      // the wave's row tiles (slot q), clamped for addressing
#pragma unroll
  for (int q = 0; q < NS; ++q) tl[q] = tile_of(q, wave);
  // ---- everything the compiler loads: Z, the inverse tiles (complete before the first hand-placed load is issued)
  // Complex products with THREE real ones (as in the factor's trailing updates): for acc -= conj(l) x, per row tile
  //   a1 = Re z - sum lr xr,   a2 = sum lm xi,   a3 = (Re z + Im z) - sum (lr - lm)(xr + xi)
  // and acc = (a1 - a2) + i (a3 - a1 - a2), formed once, when the tile is finalised.
  d4 a1[NS], a2[NS], a3[NS];
#pragma unroll
  for (int q = 0; q < NS; ++q) {
    const int Ic = min(tl[q], nct - 1);
#pragma unroll
    for (int v = 0; v < 4; ++v) {                  // Z[c][t] = conj(Laug[npad + t][c]) as acc[m = c][n = t]
      const double* zb = Lre + ((long)((npad + t0) >> 4) * npad + 16 * Ic + 4 * v) * 32;
      const double a_ = *lane_ptr<double>(zb, lz), b_ = *lane_ptr<double>(zb + 16, lz);
      a1[q][v] = (tl[q] < nct) ? a_ : 0.0;
      a2[q][v] = 0.0;
      a3[q][v] = (tl[q] < nct) ? a_ - b_ : 0.0;
    }
  }
  double wa_r[NS + 1][4], wa_i[NS + 1][4];          // inv(L_II) of slot q's tile; [NS]: of the last tile row
#pragma unroll
  for (int q = 0; q <= NS; ++q) {
    const int Ic = (q < NS) ? min(tl[q < NS ? q : 0], nct - 1) : nct - 1;
    const double* wr_ = Wgre + (long)(Ic >> 1) * 1024 + (16 * (Ic & 1)) * 33;
    const double* wi_ = Wgim + (long)(Ic >> 1) * 1024 + (16 * (Ic & 1)) * 33;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      wa_r[q][s] = *lane_ptr<double>(wr_ + (4 * s) * 32, lw);
      wa_i[q][s] = *lane_ptr<double>(wi_ + (4 * s) * 32, lw);
    }
  }
  d4 zl_r = {0., 0., 0., 0.}, zl_i = zl_r;          // Z of the last tile row when it has no slot (alone in its block)
  const bool last_alone = (Jlast >> 3) >= NS;
  if (last_alone) {
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      const double* zb = Lre + ((long)((npad + t0) >> 4) * npad + 16 * Jlast + 4 * v) * 32;
      zl_r[v] = *lane_ptr<double>(zb, lz);
      zl_i[v] = -*lane_ptr<double>(zb + 16, lz);
    }
  }
  __builtin_amdgcn_s_waitcnt(0x0f70);               // vmcnt(0): the compiler's loads are in (and it knows)
  // ---- the operand ring: set S holds rows 4 g .. 4 g + 3 (k index (g, s)) of column li of tiles (J, I_q)
  bs_d2 la[DEPTH][NS][2], lb[DEPTH][NS][2];
  auto issue = [&](auto sc, const int J) {
    constexpr int S = decltype(sc)::value;
    const int Jc = max(J, 0);
#pragma unroll
    for (int q = 0; q < NS; ++q) {
      const double* lb_ = Lre + ((long)Jc * npad + 16 * min(tl[q], nct - 1)) * 32;
      la[S][q][0] = bs_ld16(lb_, ll);
      la[S][q][1] = bs_ld16(lb_ + 2, ll);
      lb[S][q][0] = bs_ld16(lb_ + 16, ll);
      lb[S][q][1] = bs_ld16(lb_ + 18, ll);
    }
  };
  // X_J = inv(L_JJ)^H acc: stored, and published in LDS slot J & 1 as [re 16 x 16 | im 16 x 16], row-major.
  // conj(w) acc = (wr accr + wi acci) + i (wr acci - wi accr), again from three products:
  //   p1 = wr accr,  p2 = wi acci,  p3 = (wr - wi)(accr + acci):  re = p1 + p2,  im = p3 - p1 + p2
  auto finalise = [&](const int J, const d4& accr, const d4& acci, const double (&wr)[4], const double (&wi)[4]) {
    double* xb_ = xs + (J & 1) * 512;
    const d4 accs = accr + acci;
    d4 p1 = {0., 0., 0., 0.}, p2 = p1, p3 = p1;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      p1 = mfma64(wr[s], accr[s], p1);
      p2 = mfma64(wi[s], acci[s], p2);
      p3 = mfma64(wr[s] - wi[s], accs[s], p3);
    }
    const d4 xr_ = p1 + p2, xi_ = (p3 - p1) + p2;
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      const int c_ = HPX_ACC_ROW(g, v);
      xb_[c_ * 16 + li] = xr_[v];
      xb_[256 + c_ * 16 + li] = xi_[v];
      const long xo_ = (long)(16 * J + 4 * v) * TP + t0;
      *lane_ptr_w(Xre + xo_, 8u * (g * TP + li)) = xr_[v];
      *lane_ptr_w(Xim + xo_, 8u * (g * TP + li)) = xi_[v];
    }
  };
  auto finalise3 = [&](const int J, const d4& b1, const d4& b2, const d4& b3, const double (&wr)[4], const double (&wi)[4]) {
    finalise(J, b1 - b2, (b3 - b1) - b2, wr, wi);
  };
  if (wave == owner_of(Jlast)) {
    if (last_alone) finalise(Jlast, zl_r, zl_i, wa_r[NS], wa_i[NS]);
    else {
#pragma unroll
      for (int q = 0; q < NS; ++q)
        if (q == (Jlast >> 3)) finalise3(Jlast, a1[q], a2[q], a3[q], wa_r[q], wa_i[q]);
    }
  }
  // the ring's first DEPTH rows (behind the owner's stores above: they only make the first waits stricter)
  issue(std::integral_constant<int, 0>{}, Jlast);
  if (DEPTH > 1) issue(std::integral_constant<int, (DEPTH > 1 ? 1 : 0)>{}, Jlast - 1);
  if (DEPTH > 2) issue(std::integral_constant<int, (DEPTH > 2 ? 2 : 0)>{}, Jlast - 2);
  if (DEPTH > 3) issue(std::integral_constant<int, (DEPTH > 3 ? 3 : 0)>{}, Jlast - 3);
  // acc_I -= L[J, I]^H X_J with X_J from LDS slot `sl`:  conj(l) x = (lr xr + lm xi) + i (lr xi - lm xr)
  auto step = [&](auto sc, const int J) {
    constexpr int S = decltype(sc)::value;
    lds_barrier();                                  // X_J is in its slot; the other slot is free again
    bs_wait_vm<(DEPTH - 1) * NS * 4>();             // operand set S has landed (all but the later sets' loads)
#pragma unroll
    for (int q = 0; q < NS; ++q)
#pragma unroll
      for (int h = 0; h < 2; ++h) asm volatile("" : "+v"(la[S][q][h]), "+v"(lb[S][q][h]));
    const int sl = J & 1, qn = (J - 1) >> 3;
    const bool next_mine = (wave == owner_of(J - 1));
    const lds_f64* xb = (const lds_f64*)(xs + sl * 512);
    double xr_[4], xi_[4], xs_[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      xr_[s] = xb[(4 * g + s) * 16 + li];
      xi_[s] = xb[256 + (4 * g + s) * 16 + li];
      xs_[s] = xr_[s] + xi_[s];
    }
    auto update = [&](const int q) {
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const double lr_ = la[S][q][s >> 1][s & 1], lm_ = lb[S][q][s >> 1][s & 1];
        a1[q] = mfma64(-lr_, xr_[s], a1[q]);
        a2[q] = mfma64(lm_, xi_[s], a2[q]);
        a3[q] = mfma64(lm_ - lr_, xs_[s], a3[q]);
      }
    };
    if (next_mine) {                                // the next tile to finalise first: it is what the next step waits for
#pragma unroll
      for (int q = 0; q < NS; ++q)
        if (q == qn) {
          update(q);
          finalise3(J - 1, a1[q], a2[q], a3[q], wa_r[q], wa_i[q]);
        }
    }
#pragma unroll
    for (int q = 0; q < NS; ++q)
      if (!(next_mine && q == qn) && tl[q] < J) update(q);
    issue(sc, J - DEPTH);                           // (every wave, every step: the counts above rest on it)
  };
  for (int J0 = Jlast; J0 >= 1; J0 -= DEPTH) {
    step(std::integral_constant<int, 0>{}, J0);
    if (DEPTH > 1 && J0 - 1 >= 1) step(std::integral_constant<int, (DEPTH > 1 ? 1 : 0)>{}, J0 - 1);
    if (DEPTH > 2 && J0 - 2 >= 1) step(std::integral_constant<int, (DEPTH > 2 ? 2 : 0)>{}, J0 - 2);
    if (DEPTH > 3 && J0 - 3 >= 1) step(std::integral_constant<int, (DEPTH > 3 ? 3 : 0)>{}, J0 - 3);
  }
  bs_wait_vm<0>();                                  // nothing lands in a register after its last use
}

// TSPLIT: one workgroup per (baseline, t-tile) -- for batches that would leave most CUs without a baseline (config 2:
// 64 baselines on 256 CUs).  The right-hand-side columns are independent, so the t-tiles of a baseline run side by
// side; each workgroup then reads the baseline's factor for itself (the second read comes from L2 / the Infinity
// Cache: the whole batch's factors are a few tens of MB) and its chain of dependent steps is half as heavy.
template <int NS, bool TSPLIT>
__global__ __launch_bounds__(512, 2) void k_backsolve_reg(const double* __restrict__ L_all,
                                                          const double* __restrict__ Wre_all,
                                                          const double* __restrict__ Wim_all,
                                                          double* __restrict__ Xre_all, double* __restrict__ Xim_all,
                                                          const int npad, const int TP, const int ld, const int nbl) {
  __shared__ double xs[2 * 2 * 512];      // two slots x two t-tiles x (re | im) 16 x 16
  // TSPLIT: the t-tiles of a baseline run on ONE XCD right after each other (block ids are dealt round-robin over the
  // 8 XCDs: ids 8 (TT q + t) + x are baseline 8 q + x, t-tile t), in step with each other: what the first one pulls in
  // from HBM the others find in that XCD's L2
  const int TTs = TP >> 4;
  const int b = TSPLIT ? 8 * ((int)blockIdx.x / (8 * TTs)) + ((int)blockIdx.x & 7) : (int)blockIdx.x;
  const int tsel = TSPLIT ? ((int)blockIdx.x >> 3) % TTs : 0;
  if (TSPLIT && b >= nbl) return;
  const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  const double* Lre = L_all + (long)b * npad * ld * 2;
  const int nblk = (npad + HPX_NB - 1) / HPX_NB;
  const double* Wgre = Wre_all + (long)b * nblk * 1024;
  const double* Wgim = Wim_all + (long)b * nblk * 1024;
  double* Xre = Xre_all + (long)b * npad * TP;
  double* Xim = Xim_all + (long)b * npad * TP;
  const int TT = TP >> 4;
  static_assert(NS <= 2, "the register-resident form is instantiated for at most 17 tile rows");
  constexpr int DEPTH = TSPLIT ? 3 : (NS == 2 && HPX_BS_M3(NS) ? HPX_BS_DEPTH2W : 2);   // operand sets the registers hold without spills
  if (TSPLIT) {
    if constexpr (NS >= 1 && NS <= 2) bs_tile_pass_ul<NS, 3>(Lre, Wgre, Wgim, Xre, Xim, xs, npad, TP, tsel << 4, wave, lane);
    else bs_reg_pass<NS, 1, DEPTH, HPX_BS_M3(NS)>(Lre, Wgre, Wgim, Xre, Xim, xs, npad, TP, tsel << 4, wave, lane);
    return;
  }
  constexpr int NTMAX = 2;                      // accumulators + L operands within the register file
  for (int tp = 0; tp < TT; tp += NTMAX) {
    if (NTMAX == 2 && tp + 1 < TT) bs_reg_pass<NS, 2, DEPTH, HPX_BS_M3(NS)>(Lre, Wgre, Wgim, Xre, Xim, xs, npad, TP, tp << 4, wave, lane);
    else bs_reg_pass<NS, 1, DEPTH, HPX_BS_M3(NS)>(Lre, Wgre, Wgim, Xre, Xim, xs, npad, TP, tp << 4, wave, lane);
    __syncthreads();                       // the slots are free for the next pass
  }
}

int device_cus() {
  static int cus[32] = {};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 32) return 0;
  if (!cus[dev]) {
    int n = 0;
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return 0;
    cus[dev] = n;
  }
  return cus[dev];
}

template <int NS>
int launch_reg(int nbl, int npad, int TP, int ld, const double* L, const double* Wre, const double* Wim, double* Xre,
               double* Xim, hipStream_t st) {
  const int TT = TP >> 4;
  // one workgroup per (baseline, t-tile) for small batches (most CUs would be idle otherwise)
  if (TT >= 2 && 2 * nbl <= device_cus())
    hipLaunchKernelGGL((k_backsolve_reg<NS, true>), dim3(8 * TT * ((nbl + 7) / 8)), dim3(512), 0, st, L, Wre, Wim, Xre,
                       Xim, npad, TP, ld, nbl);
  else
    hipLaunchKernelGGL((k_backsolve_reg<NS, false>), dim3(nbl), dim3(512), 0, st, L, Wre, Wim, Xre, Xim, npad, TP, ld,
                       nbl);
  HPX_HIP(hipGetLastError());
  return HPX_OK;
}

}  // namespace

// 1 where the register-resident form is the faster one.  It is correct for orders up to 656 (41 tiles) and any
// number of right-hand sides, but: from 18 tiles on (three or more accumulator slots per wave) the accumulators,
// the L operands of the next step and the temporaries no longer fit 256 registers without spills, and every spill
// reload waits for the operand prefetch in flight -- 1.57 ms against 0.84 ms at C3 (33 tiles) on MI355X; with more
// than two t-tiles the factor is read once per pair of them.  At 17 tiles (N = 256): 0.077 against 0.095 ms for 64
// baselines, 0.32 against 0.32 ms for 1024.
int hpx_backsolve_reg_ok(int npad, int TP) {
  const int nct = npad >> 4;
  return nct <= 17 && TP <= 32;
}

int hpx_launch_backsolve_reg(int nbl, int npad, int TP, int ld, const double* L, const double* Wre, const double* Wim,
                             double* Xre, double* Xim, hipStream_t st) {
  const int nct = npad >> 4;
  const int ns = (nct - 1 + 7) / 8;                   // tile rows with a tile row below them, per wave
  switch (ns) {
    case 0: return launch_reg<0>(nbl, npad, TP, ld, L, Wre, Wim, Xre, Xim, st);
    case 1: return launch_reg<1>(nbl, npad, TP, ld, L, Wre, Wim, Xre, Xim, st);
    case 2: return launch_reg<2>(nbl, npad, TP, ld, L, Wre, Wim, Xre, Xim, st);
    default: break;
  }
  hpx_set_error("hpx_launch_backsolve_reg: order %d not handled", npad);
  return HPX_EINVAL;
}
