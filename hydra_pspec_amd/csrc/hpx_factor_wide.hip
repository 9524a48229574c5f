// Batched complex-Hermitian Cholesky, wide form: left-looking by SUPER-BLOCKS of 128 columns with the
// panel operand of every k-loop staged ONCE per workgroup through LDS (LDS-DMA, global_load_lds).
//
// Why (docs/HISTORY.md sections 9.3, 10.9, 10.10, 11): the 32-wide kernel of hpx_factor.hip re-reads every
// element of L once per 32-wide block column to its right (12 MB per baseline at C3) and sits on the HBM
// roof at that traffic.  With 128-wide block columns the row-tile operand is read n/128 instead of n/32
// times, and the 128 x k panel operand, which every wave of the workgroup needs, is brought in once per
// group of four row strips instead of once per wave.
//
// One workgroup (4 waves) per baseline, two workgroups per CU.  Per super-block J (columns c0 .. c0+127):
//   S  the 36 lower 16 x 16 tiles of the diagonal block  D = K[J,J] - sum_{k<c0} L[J,k] L[J,k]^H,
//      nine per wave, both MFMA operands from the staged chunks of the block's own rows;
//   F  D = L_JJ L_JJ^H in registers, right-looking by 16-wide tile columns: 16 x 16 fused Cholesky +
//      inverse on the vector ALU (the elimination of hpx_factor.hip), tiles below by one MFMA product with
//      the inverse, trailing tiles updated through an LDS exchange of the new column;
//   P  the row strips below, four at a time (one 16 x 128 strip = 8 accumulator tiles per wave):
//      k-loop over the staged chunks of the block's rows (row operand straight from global memory,
//      non-temporal, one chunk ahead), then the triangular solve as a CONTINUATION of the same stream:
//      tail chunk cj holds inv(L_cjcj) and the tiles L[ci][cj] below it, X_cj = inv(L_cjcj) acc_cj is
//      stored and becomes the B operand (in registers) of the later column tiles' updates.
// The npad/16 mod 8 tile columns left over after the last full super-block (one at every BASELINE shape:
// npad = N + 16) go one at a time through a register-only path (narrow_column).
// Complex products in S and P are three real MFMAs (hpx_factor.hip, top); F uses the four-product form
// on (re, im) tiles.
//
// Storage: the factor's 16-row panel-major layout (HPX_LIDX); Vt[b][tile][16 cols][re16|im16] holds the
// inverses of the diagonal 16 x 16 tiles in the same tile layout (the tail chunks' first operand);
// W[b][block][32][32] the inverses of the diagonal 32 x 32 blocks for k_backsolve, as before.
//
// Staging: chunks of 16 columns x 8 tiles (32 KB) alternate between TWO SEPARATE LDS arrays, and the
// buffer a step reads / fills is a compile-time constant (every loop is unrolled by two; a super-block's chunk
// counts are even).  This matters: the compiler makes a ds_read wait for every pending LDS-DMA whose target it
// cannot prove distinct from the read's -- with one array and a run-time buffer index that is an
// s_waitcnt vmcnt(0) after every stage, i.e. no chunk in flight behind the computation at all.  For the same
// reason the counted waits are the s_waitcnt builtin (visible to that analysis), fenced against compiler
// motion by empty asm statements.
#include "hpx_factor_tiles.h"

namespace {

#ifdef HPX_WIDE_TRACE
__device__ unsigned long long hpx_wtrace[1024 * 4 * HPX_WTRACE_REC];
#endif

// ---- staging ---------------------------------------------------------------------------------------
// Every wave stages two tile slots (2 wave, 2 wave + 1) of every chunk, PP pieces each: a constant
// number (2 PP = 8) of LDS-DMA operations per wave and stage, so that the counted vmcnt waits hold for every wave.
// Lane's 16 bytes inside a 1 KB piece (4 columns x [re16 | im16]): odd columns are stored [im | re] so that the
// two halves of a wave read disjoint banks (src_lane / rd_re / rd_im below).
template <int PAR>
__device__ HPX_INL void stage_k(const WideCtx& X, const int ct0, const int chunk, const unsigned src_lane) {
  const unsigned bb = lds_addr(stage_buf<PAR>());
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    const int ci = 2 * X.wave + u;
    glds_tile(X.Lb + (long)(ct0 + ci) * X.ptile + (long)chunk * TILE_D, 8 * src_lane, bb + ci * TILE_D * 8);
  }
}
// tail step cj: slot cj <- inv(L_cjcj) (Vt), slots ci > cj <- L[ci][cj]; the slots above are not read and not staged
// (a wave then issues fewer than 8 operations for this stage: the counted waits of the tail steps only name what
// follows the stage -- the wave's 8 stores -- so they hold whatever the count)
template <int PAR>
__device__ HPX_INL void stage_t(const WideCtx& X, const int ct0, const int cj, const unsigned src_lane) {
  const unsigned bb = lds_addr(stage_buf<PAR>());
  const double* vsrc = X.Vt + (long)(ct0 + cj) * 512;
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    const int ci = 2 * X.wave + u;
    if (ci < cj) continue;
    const double* src = vsrc;
    if (ci > cj) src = X.Lb + (long)(ct0 + ci) * X.ptile + (long)((ct0 + cj) * 16) * 32;
    glds_tile(src, 8 * src_lane, bb + ci * TILE_D * 8);
  }
}
// step `st` of a pass (k-chunks 0 .. nk-1, then the eight tail steps) into buffer PAR
template <int PAR>
__device__ HPX_INL void stage_step(const WideCtx& X, const int ct0, const int st, const int nk,
                                   const unsigned src_lane) {
  if (st < nk) stage_k<PAR>(X, ct0, st, src_lane);
  else stage_t<PAR>(X, ct0, st - nk, src_lane);
}


// ---- S: the diagonal block's update.  Wave W owns tile rows 7 - W and W of the block's lower triangle
//      (8 - W and W + 1 tiles: nine each): slot s < 8 - W is tile (7 - W, s), the others (W, s - (8 - W)).
//      Per k-step the two row operands are read once and the column operands one tile ahead of their
//      MFMAs; compile-time tile indices, so every LDS offset is an immediate.
//      On return a1 = re, a2 = im of D^T[c][r] per tile.
template <int W> struct DiagDeal {
  static constexpr int row(const int s) { return s < 8 - W ? 7 - W : W; }
  static constexpr int col(const int s) { return s < 8 - W ? s : s - (8 - W); }
};
template <bool GEN, bool GLDS, int W>
__device__ HPX_INL void diag_update(const WideCtx& X, const hpx_gen& G, const GenVec<GLDS>& V, const int ct0,
                                    d4 (&a1)[9], d4 (&a2)[9]) {
  typedef DiagDeal<W> TD;
  constexpr int RA = 7 - W, RB = W;                 // the wave's two tile rows
  const int lane = opaque(X.lane), li = lane & 15, g = lane >> 4;
  const unsigned src_lane = g * 32 + 2 * ((li + 8 * (g & 1)) & 15);
  const int rd_re = g * 32 + li + 16 * (g & 1), rd_im = g * 32 + li + 16 * (1 - (g & 1));
  const int c0 = ct0 * 16;
  d4 a3[9];
  if (GEN && GLDS && c0 + 128 <= G.rmin) {       // closed form from LDS: no vector-memory instruction on this path
#pragma unroll
    for (int s = 0; s < 9; ++s) {
      d4 vr, vi;
      tile_init_closed<GLDS>(V, c0 + 16 * TD::row(s), c0 + 16 * TD::col(s), li, g, vr, vi);
      a1[s] = -0.5 * vr;
      a2[s] = -0.5 * vr;
      a3[s] = vi;
    }
  } else {
#pragma unroll
    for (int s = 0; s < 9; ++s) {
      d4 vr, vi;
      tile_init<GEN, GLDS>(G, V, X.Lb, c0 + 16 * TD::row(s), c0 + 16 * TD::col(s), X.npad, li, g, vr, vi);
      a1[s] = -0.5 * vr;
      a2[s] = -0.5 * vr;
      a3[s] = vi;
    }
    wait_compiler_loads();
  }
  const int nk = c0 / KC;                  // (a multiple of 8)
  HPX_TR(X, 1, ct0 >> 3, 0, 0);
  if (nk > 0) {
    stage_k<0>(X, ct0, 0, src_lane);
    // one chunk: wait for it (nothing but LDS-DMA is in flight), barrier, issue the next one into the other
    // buffer (after the last chunk: a harmless re-stage of it), compute
#define HPX_S_STEP(PAR_, ch_)                                                                        \
    {                                                                                                \
      wait_vm<0>();                                                                                  \
      wg_barrier();                                                                                  \
      stage_k<1 - (PAR_)>(X, ct0, min((ch_) + 1, nk - 1), src_lane);                                 \
      const lds_f64* B = (const lds_f64*)stage_buf<PAR_>();                                          \
      _Pragma("unroll") for (int p = 0; p < PP; ++p) {                                               \
        const double bra = B[RA * TILE_D + p * 128 + rd_re], bma = B[RA * TILE_D + p * 128 + rd_im]; \
        const double brb = B[RB * TILE_D + p * 128 + rd_re], bmb = B[RB * TILE_D + p * 128 + rd_im]; \
        const double bda = bra - bma, bdb = brb - bmb;                                               \
        double pr = B[p * 128 + rd_re], pi = B[p * 128 + rd_im];                                     \
        _Pragma("unroll") for (int c = 0; c <= RA; ++c) {                                            \
          /* column tile c: slot c (row RA) and, for c <= RB, slot 8 - W + c (row RB) */             \
          const double cr = pr, cm = pi, psm = pr + pi;                                              \
          if (c + 1 <= RA) {                                                                         \
            pr = B[(c + 1) * TILE_D + p * 128 + rd_re];                                              \
            pi = B[(c + 1) * TILE_D + p * 128 + rd_im];                                              \
          }                                                                                          \
          __builtin_amdgcn_sched_barrier(0);                                                         \
          a1[c] = mfma64(cr, bra, a1[c]);                                                            \
          a2[c] = mfma64(cm, bma, a2[c]);                                                            \
          a3[c] = mfma64(psm, bda, a3[c]);                                                           \
          if (c <= RB) {                                                                             \
            a1[8 - W + c] = mfma64(cr, brb, a1[8 - W + c]);                                          \
            a2[8 - W + c] = mfma64(cm, bmb, a2[8 - W + c]);                                          \
            a3[8 - W + c] = mfma64(psm, bdb, a3[8 - W + c]);                                         \
          }                                                                                          \
          __builtin_amdgcn_sched_barrier(0);                                                         \
        }                                                                                            \
      }                                                                                              \
    }
    for (int ch = 0; ch < nk; ch += 2) {
      HPX_S_STEP(0, ch)
      HPX_S_STEP(1, ch + 1)
    }
#undef HPX_S_STEP
    wait_vm<0>();                       // the re-staged chunk: nothing may land after F starts using the area
  }
  HPX_TR(X, 1, ct0 >> 3, 0, 1);
#pragma unroll
  for (int s = 0; s < 9; ++s) {
    const d4 re_ = -(a1[s] + a2[s]);
    const d4 im_ = a3[s] - a1[s] + a2[s];
    a1[s] = re_;
    a2[s] = im_;
  }
}

// ---- F: D = L_JJ L_JJ^H on the (re, im) tiles; D^T[c][r] per tile: register v <-> column g + 4 v, lane li <-> row
__device__ HPX_INL bool diag_factor(const WideCtx& X, const int ct0, d4 (&a1)[9], d4 (&a2)[9]) {
  int tr[9], tc[9];
#pragma unroll
  for (int s = 0; s < 9; ++s) {
    // (uniform: wave is a scalar) the deal of diag_update
    tr[s] = (s < 8 - X.wave) ? 7 - X.wave : X.wave;
    tc[s] = (s < 8 - X.wave) ? s : s - (8 - X.wave);
  }
  lds_f64* const Xs = (lds_f64*)hpx_stage0;
  lds_f64* const Vs = (lds_f64*)(hpx_stage1 + FV_OFF);
  bool bad = false;
  d4 t_re = {0., 0., 0., 0.}, t_im = {0., 0., 0., 0.};     // L10 inv(L00) of the current pair of tiles (one wave)
  lds_barrier();                                          // the staging area is free
  for (int i = 0; i < 8; ++i) {
    const int lane = opaque(X.lane), li = lane & 15, g = lane >> 4;
    const int rd_re = g * 32 + li + 16 * (g & 1), rd_im = g * 32 + li + 16 * (1 - (g & 1));
    // (1) the diagonal tile goes to the elimination: all 256 threads (elim16) -- the one-wave form that eliminates the
    // tile where it lies (elim16w) measured no faster here: 4.13 against 4.08 ms per launch at C3
    // (tools/experiments/elim/wide_elim_wave.patch)
    HPX_TR(X, 2, ct0 >> 3, i, 0);
    lds_barrier();                                        // Vs and the elimination's scratch are free
#pragma unroll
    for (int s = 0; s < 9; ++s)
      if (tr[s] == i && tc[s] == i) {
#pragma unroll
        for (int v = 0; v < 4; ++v) {
          Xs[li * 16 + HPX_ACC_ROW(g, v)] = a1[s][v];
          Xs[256 + li * 16 + HPX_ACC_ROW(g, v)] = a2[s][v];
        }
      }
    lds_barrier();
    bad |= elim16(X, ct0 + i, false);
    HPX_TR(X, 2, ct0 >> 3, i, 1);
    // (2b) odd tile of a pair: W10 = -inv(L11) (L10 inv(L00))
    if ((i & 1) && X.wave == ((i >> 1) & 3)) {
      d4 zr = {0., 0., 0., 0.}, zi = {0., 0., 0., 0.};
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        const double ar = Vs[v * 128 + rd_re], ai = Vs[v * 128 + rd_im];     // inv(L11)[i' = li][r = 4 v + g]
        zr = mfma64(-ar, t_re[v], zr);
        zr = mfma64(ai, t_im[v], zr);
        zi = mfma64(-ar, t_im[v], zi);
        zi = mfma64(-ai, t_re[v], zi);
      }
      double* wgr = X.Wgre + (long)((ct0 + i) >> 1) * 1024;
      double* wgi = X.Wgim + (long)((ct0 + i) >> 1) * 1024;
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        wgr[(16 + HPX_ACC_ROW(g, v)) * 32 + li] = zr[v];
        wgi[(16 + HPX_ACC_ROW(g, v)) * 32 + li] = zi[v];
      }
    }
    // (3) tiles below: X^T[c'][r'] = sum_c conj(inv(L)[c'][c]) D^T[c][r'], stored and handed to the others
#ifndef HPX_DBG_F_NOX
#pragma unroll
    for (int s = 0; s < 9; ++s)
      if (tc[s] == i && tr[s] > i) {
        d4 xr = {0., 0., 0., 0.}, xi = {0., 0., 0., 0.};
#pragma unroll
        for (int v = 0; v < 4; ++v) {
          const double pr = Vs[v * 128 + rd_re], pi = Vs[v * 128 + rd_im];
          xr = mfma64(pr, a1[s][v], xr);
          xr = mfma64(pi, a2[s][v], xr);
          xi = mfma64(pr, a2[s][v], xi);
          xi = mfma64(-pi, a1[s][v], xi);
        }
        double* o_ = X.Lb + HPX_LIDX((ct0 + tr[s]) * 16 + li, (ct0 + i) * 16 + g, X.npad);
        lds_f64* xs = Xs + tr[s] * 512;
#pragma unroll
        for (int v = 0; v < 4; ++v) {
          o_[(4 * v) * 32] = xr[v];
          o_[(4 * v) * 32 + 16] = xi[v];
          const int k = g + 4 * v;                         // column of the tile; row li
          xs[k * 32 + li + 16 * (k & 1)] = xr[v];
          xs[k * 32 + li + 16 * (1 - (k & 1))] = xi[v];
        }
      }
#endif
    HPX_TR(X, 2, ct0 >> 3, i, 2);
    lds_barrier();
    HPX_TR(X, 2, ct0 >> 3, i, 3);
    // (3b) even tile of a pair: T = L10 inv(L00), kept in registers until the pair's second inverse exists
    if (!(i & 1) && X.wave == ((i >> 1) & 3)) {
      t_re = (d4){0., 0., 0., 0.};
      t_im = t_re;
      const lds_f64* l10 = Xs + (i + 1) * 512;
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        const double ar = l10[v * 128 + rd_re], ai = l10[v * 128 + rd_im];      // L10[r = li][k = 4 v + g]
        const int k = 4 * v + g;                                                   // inv(L00)[k][c' = li]
        const double br = Vs[li * 32 + k + 16 * (li & 1)], bm = Vs[li * 32 + k + 16 * (1 - (li & 1))];
        t_re = mfma64(ar, br, t_re);
        t_re = mfma64(-ai, bm, t_re);
        t_im = mfma64(ar, bm, t_im);
        t_im = mfma64(ai, br, t_im);
      }
    }
    // (4) trailing tiles: D^T[c'][r'] -= sum_k conj(X(c,i)[c'][k]) X(r,i)[r'][k]
#ifndef HPX_DBG_F_NOUPD
#pragma unroll
    for (int s = 0; s < 9; ++s)
      if (tc[s] > i) {
        const lds_f64* pa = Xs + tc[s] * 512;
        const lds_f64* pb = Xs + tr[s] * 512;
#pragma unroll
        for (int v = 0; v < 4; ++v) {
          const double pr = pa[v * 128 + rd_re], pi = pa[v * 128 + rd_im];
          const double br = pb[v * 128 + rd_re], bm = pb[v * 128 + rd_im];
          a1[s] = mfma64(-pr, br, a1[s]);
          a1[s] = mfma64(-pi, bm, a1[s]);
          a2[s] = mfma64(-pr, bm, a2[s]);
          a2[s] = mfma64(pi, br, a2[s]);
        }
      }
#endif
    HPX_TR(X, 2, ct0 >> 3, i, 4);
  }
  __syncthreads();        // F's stores (L_JJ, Vt) are complete before the passes stage them
  HPX_TR(X, 2, ct0 >> 3, 8, 0);
  return bad;
}

// ---- P: the row strips below the full super-block at ct0, four at a time ----------------------------
// The chunks of all groups form ONE stream through the two staging buffers: every step waits for its own
// chunk, passes the barrier, issues the next chunk (the next group's first when this one is done) into the other
// buffer and computes; nothing is drained between the groups.  Counted waits: every wave issues 8 LDS-DMA
// operations per stage, and an active wave 8 loads (row operand of the next chunk) or 8 stores (its solved
// columns) per step AFTER them, so with the vmcnt counter in issue order "all but the 8 youngest" covers the
// chunk being waited for.  A super-block's steps per group (c0 / 16 + 8) are even: the buffer of a step is its
// parity within the group.
template <bool GEN, bool GLDS>
__device__ HPX_INL void strip_passes(const WideCtx& X, const hpx_gen& G, const GenVec<GLDS>& V, const int ct0) {
  const int lane = opaque(X.lane), li = lane & 15, g = lane >> 4;
  const unsigned src_lane = g * 32 + 2 * ((li + 8 * (g & 1)) & 15);
  const int rd_re = g * 32 + li + 16 * (g & 1), rd_im = g * 32 + li + 16 * (1 - (g & 1));
  const unsigned blane = g * 32 + li;
  const int c0 = ct0 * 16;
  const int first = ct0 + 8;
  if (first >= X.nrt) return;
  const int nk = c0 / KC, total = nk + 8;
  double rb[PP], rm[PP];
  // ---- prologue of the pass: the first stage, the first group's first row operand
  stage_step<0>(X, ct0, 0, nk, src_lane);
#define HPX_ROW_LOAD(base_)                                                      \
  {                                                                              \
    rb[0] = ld_nt<0>(base_, 8 * blane);    rm[0] = ld_nt<128>(base_, 8 * blane);   \
    rb[1] = ld_nt<1024>(base_, 8 * blane); rm[1] = ld_nt<1152>(base_, 8 * blane);  \
    rb[2] = ld_nt<2048>(base_, 8 * blane); rm[2] = ld_nt<2176>(base_, 8 * blane);  \
    rb[3] = ld_nt<3072>(base_, 8 * blane); rm[3] = ld_nt<3200>(base_, 8 * blane);  \
  }
  if (first + X.wave < X.nrt && nk > 0) HPX_ROW_LOAD(X.Lb + (long)(first + X.wave) * X.ptile)
  bool first_step = true;
  for (int rt0 = first; rt0 < X.nrt; rt0 += 4) {
    const int rt = rt0 + X.wave;
    const bool active = rt < X.nrt;
    const bool last_group = rt0 + 4 >= X.nrt;
    const bool next_active = rt + 4 < X.nrt;
    d4 a1[8], a2[8], a3[8];
    if (active) {
      if (GEN && GLDS && rt * 16 < G.rmin && c0 + 128 <= G.rmin) {     // no vector-memory instruction on this path
        // Rows BELOW the super-block: every tile is off the diagonal (r0 >= c0 + 128 > c0 + 16 ci), entry (r, c) is the
        // circulant's element r - c = d0 + 16 (7 - ci) + 4 (3 - v) with ONE per-lane index d0 >= 1 and compile-time
        // offsets.  Through tile_init_closed the compiler could not know that and laid out the diagonal form as well
        // (masked lanes, fma with 1 / a) with 19 hoisted per-lane addresses kept in scratch: every tile's reload sat
        // behind an s_waitcnt vmcnt(0) -- a drain of the last tail step's stores, the next row operand and the staged
        // chunk, then seven more scratch round trips: the 4.4 - 6 us "group prologue" of the step trace
        // (profiles/r05_trace_factor_wide_c3.txt).  The index hangs on an opaque copy of the lane: not hoisted.
        const int d0 = rt * 16 - c0 - 124 + opaque(li) - opaque(g);
        const auto qr = V.cre + d0;
        const auto qi = V.cim + d0;
#pragma unroll
        for (int ci = 0; ci < 8; ++ci) {
          d4 vr, vi;
#pragma unroll
          for (int v = 0; v < 4; ++v) {
            vr[v] = qr[16 * (7 - ci) + 4 * (3 - v)];
            vi[v] = qi[16 * (7 - ci) + 4 * (3 - v)];
          }
          a1[ci] = -0.5 * vr;
          a2[ci] = -0.5 * vr;
          a3[ci] = vi;
        }
      } else {
        // (the last strips of a super-block: edge tiles.  Their addresses hang on opaque copies of the indices:
        // computed at the top of every group's body instead -- where the compiler moved them -- they cost every
        // group some 500 instructions and 50 spills)
        const int rt16 = opaque_s(rt * 16), c0o = opaque_s(c0), lio = opaque(li), go = opaque(g);
        if (GEN && GLDS && G.ere != nullptr && rt16 >= G.rmin && c0o + 128 <= G.rmin) {
          // A strip of EDGE tiles (foreground rows / padding / right-hand sides) against signal columns: tile_init's
          // edge arithmetic, operation for operation (same bits), but laid out for the whole strip -- the three
          // conditions (edge tiles present, right-hand-side row, omega term) are uniform over the strip's eight tiles and
          // are taken ONCE, and every load lands in the accumulator register it ends up in: 128 loads in flight behind
          // one wait (plus 64 for the omega term) instead of six dependent round trips per tile, 8 tiles in turn
          // (12 - 14 us per super-block in the step trace, profiles/r06_trace_factor_wide_c3.txt).
          // (uniform base + 32-bit lane offset + immediate: one address register for the strip, not one per load)
          typedef const __attribute__((address_space(1))) char* gbytes;
          const gbytes eu = (gbytes)(G.ere + HPX_EIDX(rt16, c0o, G.rmin));
          const unsigned lb = 8u * (unsigned)(go * 32 + lio);
#pragma unroll
          for (int ci = 0; ci < 8; ++ci) {
            const gbytes tb = eu + ci * 4096;
#pragma unroll
            for (int v = 0; v < 4; ++v) {
              a1[ci][v] = *(const glb_f64*)(tb + lb + 1024 * v);
              a3[ci][v] = *(const glb_f64*)(tb + lb + 1024 * v + 128);
            }
          }
          if (rt16 >= X.npad) {
            if (G.has_omega) {
              const long pb = ((long)((rt16 - X.npad) >> 4) * G.NP + c0o) << 4;
              const gbytes pru = (gbytes)(G.p2tre + pb), piu = (gbytes)(G.p2tim + pb);
              const unsigned lp = 8u * (unsigned)(go * 16 + lio);
              const auto ia = V.ia + c0o + go;
#pragma unroll
              for (int ci = 0; ci < 8; ++ci)
#pragma unroll
                for (int v = 0; v < 4; ++v) a2[ci][v] = *(const glb_f64*)(pru + ci * 2048 + lp + 512 * v);
              // (1 / a read tile by tile, right in front of its products: all 32 values at once do not fit beside the
              // three accumulator sets, and what the compiler then spills are freshly LOADED values, each behind a full wait)
              __builtin_amdgcn_sched_barrier(0);
#pragma unroll
              for (int ci = 0; ci < 8; ++ci) {
#pragma unroll
                for (int v = 0; v < 4; ++v) a1[ci][v] = fma(ia[16 * ci + 4 * v], a2[ci][v], a1[ci][v]);
                __builtin_amdgcn_sched_barrier(0);
              }
#pragma unroll
              for (int ci = 0; ci < 8; ++ci)
#pragma unroll
                for (int v = 0; v < 4; ++v) a2[ci][v] = *(const glb_f64*)(piu + ci * 2048 + lp + 512 * v);
              __builtin_amdgcn_sched_barrier(0);
#pragma unroll
              for (int ci = 0; ci < 8; ++ci) {
#pragma unroll
                for (int v = 0; v < 4; ++v) a3[ci][v] = fma(ia[16 * ci + 4 * v], a2[ci][v], a3[ci][v]);
                __builtin_amdgcn_sched_barrier(0);
              }
            }
#pragma unroll
            for (int ci = 0; ci < 8; ++ci) a3[ci] = -a3[ci];
          }
#pragma unroll
          for (int ci = 0; ci < 8; ++ci) {
            a1[ci] = -0.5 * a1[ci];
            a2[ci] = a1[ci];
          }
        } else {
#pragma unroll
          for (int ci = 0; ci < 8; ++ci) {
            d4 vr, vi;
            tile_init<GEN, GLDS>(G, V, X.Lb, rt16, c0o + 16 * ci, X.npad, lio, go, vr, vi);
            a1[ci] = -0.5 * vr;
            a2[ci] = -0.5 * vr;
            a3[ci] = vi;
          }
        }
        wait_compiler_loads();
      }
    }
    const double* brow = X.Lb + (long)(active ? rt : first) * X.ptile;      // (uniform) + lane offset `blane`
    // step `st_` of this group (buffer PAR_): wait for its chunk, barrier, issue the next one
#define HPX_STEP_HEAD(PAR_, st_)                                                                     \
    HPX_TR(X, 3, ct0 >> 3, (rt0 - first) >> 2, st_);                                                 \
    if (first_step || !active) wait_vm<0>();            /* the chunk of this step has landed ... */  \
    else wait_vm<8>();              /* ... behind this wave's 8 row-operand loads or 8 stores */     \
    first_step = false;                                                                              \
    HPX_TR(X, 4, ct0 >> 3, (rt0 - first) >> 2, st_);                                                 \
    wg_barrier();                                                                                    \
    HPX_TR(X, 5, ct0 >> 3, (rt0 - first) >> 2, st_);                                                 \
    {                                                                                                \
      int sn_ = (st_) + 1;                                                                           \
      if (sn_ >= total) sn_ = last_group ? total - 1 : 0;        /* a harmless re-stage / the next group's first */ \
      stage_step<1 - (PAR_)>(X, ct0, sn_, nk, src_lane);                                             \
    }
    // ---- k-loop
#define HPX_K_STEP(PAR_, st_)                                                                        \
    {                                                                                                \
      HPX_STEP_HEAD(PAR_, st_)                                                                       \
      if (active) {                                                                                  \
        const lds_f64* B = (const lds_f64*)stage_buf<PAR_>();                                        \
        const double* bnext = brow + (long)min((st_) + 1, nk - 1) * TILE_D;   /* last: a harmless re-read */ \
        _Pragma("unroll") for (int p = 0; p < PP; ++p) {                                             \
          /* k-step p's pair was loaded right after k-step p of the step before: behind it are the  \
             pairs p+1 .. 3 of that step, the 8 DMA of this step's head and the pairs 0 .. p-1 of   \
             this step -- 14 operations whatever p is */                                            \
          wait_vm<14>();                                                                             \
          anchor_pair(rb[p], rm[p]);                                                                 \
          const double br = rb[p], bm = rm[p], bd = br - bm;                                         \
          double pr = B[p * 128 + rd_re], pi = B[p * 128 + rd_im];                                   \
          _Pragma("unroll") for (int ci = 0; ci < 8; ++ci) {                                         \
            const double cr = pr, cm = pi, psm = pr + pi;                                            \
            if (ci + 1 < 8) {                                                                        \
              pr = B[(ci + 1) * TILE_D + p * 128 + rd_re];                                           \
              pi = B[(ci + 1) * TILE_D + p * 128 + rd_im];                                           \
            }                                                                                        \
            __builtin_amdgcn_sched_barrier(0);                                                       \
            a1[ci] = mfma64(cr, br, a1[ci]);                                                         \
            a2[ci] = mfma64(cm, bm, a2[ci]);                                                         \
            a3[ci] = mfma64(psm, bd, a3[ci]);                                                        \
            __builtin_amdgcn_sched_barrier(0);                                                       \
          }                                                                                          \
          if (p == 0) { rb[0] = ld_nt<0>(bnext, 8 * blane);    rm[0] = ld_nt<128>(bnext, 8 * blane); }   \
          if (p == 1) { rb[1] = ld_nt<1024>(bnext, 8 * blane); rm[1] = ld_nt<1152>(bnext, 8 * blane); }  \
          if (p == 2) { rb[2] = ld_nt<2048>(bnext, 8 * blane); rm[2] = ld_nt<2176>(bnext, 8 * blane); }  \
          if (p == 3) { rb[3] = ld_nt<3072>(bnext, 8 * blane); rm[3] = ld_nt<3200>(bnext, 8 * blane); }  \
        }                                                                                            \
      }                                                                                              \
    }
    for (int st = 0; st < nk; st += 2) {
      HPX_K_STEP(0, st)
      HPX_K_STEP(1, st + 1)
    }
    // ---- tail: the triangular solve as further chunks of the same stream (step nk + CJ: buffer CJ & 1)
#define HPX_TAIL_STEP(CJ_)                                                                           \
  {                                                                                                  \
    d4 x1 = {0., 0., 0., 0.}, x2 = x1, x3 = x1;                                                      \
    HPX_STEP_HEAD((CJ_) & 1, nk + (CJ_))                                                             \
    if (active) {                                                                                    \
      const lds_f64* B = (const lds_f64*)stage_buf<(CJ_) & 1>();                                     \
      {                                                                                              \
        const d4 re_ = -(a1[CJ_] + a2[CJ_]);                                                         \
        const d4 im_ = a3[CJ_] - a1[CJ_] + a2[CJ_];                                                  \
        a1[CJ_] = re_; a2[CJ_] = im_; a3[CJ_] = re_ - im_;                                           \
      }                                                                                              \
      {                                                                                              \
        double pr = B[(CJ_) * TILE_D + rd_re], pi = B[(CJ_) * TILE_D + rd_im];                       \
        _Pragma("unroll") for (int p = 0; p < PP; ++p) {                                             \
          const double cr = pr, cm = pi, psm = pr + pi;                                              \
          if (p + 1 < PP) {                                                                          \
            pr = B[(CJ_) * TILE_D + (p + 1) * 128 + rd_re];                                          \
            pi = B[(CJ_) * TILE_D + (p + 1) * 128 + rd_im];                                          \
          }                                                                                          \
          __builtin_amdgcn_sched_barrier(0);                                                         \
          x1 = mfma64(cr, a1[CJ_][p], x1);                                                           \
          x2 = mfma64(cm, a2[CJ_][p], x2);                                                           \
          x3 = mfma64(psm, a3[CJ_][p], x3);                                                          \
          __builtin_amdgcn_sched_barrier(0);                                                         \
        }                                                                                            \
      }                                                                                              \
      if ((CJ_) == 7 && next_active && nk > 0) {                                                     \
        /* the next group's first row operand, in front of this step's stores (the youngest 8) */    \
        HPX_ROW_LOAD(X.Lb + (long)(rt + 4) * X.ptile)                                                \
      }                                                                                              \
      /* (uniform base) + lane offset: no 64-bit address registers per column tile */               \
      double* o_ = X.Lb + ((long)rt * X.npad + (ct0 + (CJ_)) * 16) * 32;                             \
      _Pragma("unroll") for (int v = 0; v < 4; ++v) {                                                \
        const double xr = x1[v] + x2[v], xi = x1[v] - x2[v] - x3[v];                                 \
        a1[CJ_][v] = xr; a2[CJ_][v] = xi; a3[CJ_][v] = xr - xi;                                      \
      }                                                                                              \
      st_g<0>(o_, 8 * blane, a1[CJ_][0]);    st_g<128>(o_, 8 * blane, a2[CJ_][0]);                   \
      st_g<1024>(o_, 8 * blane, a1[CJ_][1]); st_g<1152>(o_, 8 * blane, a2[CJ_][1]);                  \
      st_g<2048>(o_, 8 * blane, a1[CJ_][2]); st_g<2176>(o_, 8 * blane, a2[CJ_][2]);                  \
      st_g<3072>(o_, 8 * blane, a1[CJ_][3]); st_g<3200>(o_, 8 * blane, a2[CJ_][3]);                  \
      __builtin_amdgcn_sched_barrier(0);                                                             \
      if ((CJ_) < 7) {                                                                               \
        double pr = B[((CJ_) + 1) * TILE_D + rd_re], pi = B[((CJ_) + 1) * TILE_D + rd_im];           \
        _Pragma("unroll") for (int ci = (CJ_) + 1; ci < 8; ++ci) {                                   \
          _Pragma("unroll") for (int p = 0; p < PP; ++p) {                                           \
            const double cr = pr, cm = pi, psm = pr + pi;                                            \
            if (p + 1 < PP) {                                                                        \
              pr = B[ci * TILE_D + (p + 1) * 128 + rd_re];                                           \
              pi = B[ci * TILE_D + (p + 1) * 128 + rd_im];                                           \
            } else if (ci + 1 < 8) {                                                                 \
              pr = B[(ci + 1) * TILE_D + rd_re];                                                     \
              pi = B[(ci + 1) * TILE_D + rd_im];                                                     \
            }                                                                                        \
            __builtin_amdgcn_sched_barrier(0);                                                       \
            a1[ci] = mfma64(cr, a1[CJ_][p], a1[ci]);                                                 \
            a2[ci] = mfma64(cm, a2[CJ_][p], a2[ci]);                                                 \
            a3[ci] = mfma64(psm, a3[CJ_][p], a3[ci]);                                                \
            __builtin_amdgcn_sched_barrier(0);                                                       \
          }                                                                                          \
        }                                                                                            \
      }                                                                                              \
    }                                                                                                \
  }
    HPX_TAIL_STEP(0)
    HPX_TAIL_STEP(1)
    HPX_TAIL_STEP(2)
    HPX_TAIL_STEP(3)
    HPX_TAIL_STEP(4)
    HPX_TAIL_STEP(5)
    HPX_TAIL_STEP(6)
    HPX_TAIL_STEP(7)
#undef HPX_TAIL_STEP
#undef HPX_K_STEP
#undef HPX_STEP_HEAD
#undef HPX_ROW_LOAD
  }
  // the pass's stores and the re-staged chunk are done, every wave has finished reading the buffers
  HPX_TR(X, 6, ct0 >> 3, 0, 0);
  wait_vm<0>();
  wg_barrier();
  HPX_TR(X, 6, ct0 >> 3, 0, 1);
}

// ---- a single 16-wide tile column t (those after the last full super-block): register-only ----------
//   diagonal tile: K-split of the update over the four waves, partial sums through LDS, elimination;
//   tiles below: one per wave at a time, both operands straight from global memory (double buffered).
template <bool GEN, bool GLDS>
__device__ HPX_INL bool narrow_column(const WideCtx& X, const hpx_gen& G, const GenVec<GLDS>& V, const int t) {
  const int lane = opaque(X.lane), li = lane & 15, g = lane >> 4;
  const int rd_re = g * 32 + li + 16 * (g & 1), rd_im = g * 32 + li + 16 * (1 - (g & 1));
  lds_f64* const Xs = (lds_f64*)hpx_stage0;
  lds_f64* const Vs = (lds_f64*)(hpx_stage1 + FV_OFF);
  const int c0 = 16 * t;
  // ---- diagonal tile: partial sums over this wave's 16-column chunks (operand p = b = L[t rows][k])
  {
    d4 q1 = {0., 0., 0., 0.}, q2 = q1, q3 = q1;
    const double* prow = X.Lb + (long)t * X.ptile + g * 32 + li;
    for (int ch = X.wave; ch < t; ch += 4) {
      double pr[4], pi[4];
#pragma unroll
      for (int p = 0; p < 4; ++p) {
        pr[p] = prow[(long)ch * 512 + p * 128];
        pi[p] = prow[(long)ch * 512 + p * 128 + 16];
      }
#pragma unroll
      for (int p = 0; p < 4; ++p) {
        q1 = mfma64(pr[p], pr[p], q1);
        q2 = mfma64(pi[p], pi[p], q2);
        q3 = mfma64(pr[p] + pi[p], pr[p] - pi[p], q3);
      }
    }
    // sum_k conj(p) b as [row li][column g + 4 v] into this wave's slot (slots 1..4 of Xs)
    lds_f64* part = Xs + (1 + X.wave) * 512;
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      part[li * 16 + HPX_ACC_ROW(g, v)] = q1[v] + q2[v];
      part[256 + li * 16 + HPX_ACC_ROW(g, v)] = q1[v] - q2[v] - q3[v];
    }
  }
  lds_barrier();
  {
    const int tid = opaque(X.tid), q = tid & 15, ib = tid >> 4;
    double kr = 0.0, ki = 0.0;
    if (q <= ib) entry_init<GEN, GLDS>(G, V, X.Lb, c0 + ib, c0 + q, X.npad, kr, ki);
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      kr -= Xs[(1 + w) * 512 + ib * 16 + q];
      ki -= Xs[(1 + w) * 512 + 256 + ib * 16 + q];
    }
    Xs[ib * 16 + q] = kr;                   // Ein: read back by this thread
    Xs[256 + ib * 16 + q] = ki;
  }
  bool bad = elim16(X, t, (t + 1 == X.nct));
  // ---- odd tile of a pair: W10 = -inv(L11) L10 inv(L00), operands from global memory
  if ((t & 1) && X.wave == 0) {
    d4 tre = {0., 0., 0., 0.}, tim = tre;
    const double* l10 = X.Lb + HPX_LIDX(c0 + li, c0 - 16 + g, X.npad);        // L10[r = li][k = 4 v + g]
    const double* v00 = X.Vt + (long)(t - 1) * 512 + li * 32 + g;             // inv(L00)[k = 4 v + g][c' = li]
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      const double ar = l10[(4 * v) * 32], ai = l10[(4 * v) * 32 + 16];
      const double br = v00[4 * v], bm = v00[4 * v + 16];
      tre = mfma64(ar, br, tre);
      tre = mfma64(-ai, bm, tre);
      tim = mfma64(ar, bm, tim);
      tim = mfma64(ai, br, tim);
    }
    d4 zr = {0., 0., 0., 0.}, zi = zr;
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      const double ar = Vs[v * 128 + rd_re], ai = Vs[v * 128 + rd_im];
      zr = mfma64(-ar, tre[v], zr);
      zr = mfma64(ai, tim[v], zr);
      zi = mfma64(-ar, tim[v], zi);
      zi = mfma64(-ai, tre[v], zi);
    }
    double* wgr = X.Wgre + (long)(t >> 1) * 1024;
    double* wgi = X.Wgim + (long)(t >> 1) * 1024;
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      wgr[(16 + HPX_ACC_ROW(g, v)) * 32 + li] = zr[v];
      wgi[(16 + HPX_ACC_ROW(g, v)) * 32 + li] = zi[v];
    }
  }
  // ---- tiles below: X = (K[r, t] - sum_k L[r, k] L[t, k]^H) inv(L_tt)^H
  for (int rt = t + 1 + X.wave; rt < X.nrt; rt += 4) {
    d4 a1, a2, a3;
    {
      d4 vr, vi;
      tile_init<GEN, GLDS>(G, V, X.Lb, rt * 16, c0, X.npad, li, g, vr, vi);
      a1 = -0.5 * vr;
      a2 = -0.5 * vr;
      a3 = vi;
    }
    const double* prow = X.Lb + (long)t * X.ptile + g * 32 + li;
    const double* brow = X.Lb + (long)rt * X.ptile + g * 32 + li;
    double pr0[4], pi0[4], br0[4], bm0[4], pr1[4], pi1[4], br1[4], bm1[4];
#define HPX_NC_LOAD(pr_, pi_, br_, bm_, ch_)                                   \
  _Pragma("unroll") for (int p = 0; p < 4; ++p) {                              \
    pr_[p] = prow[(long)(ch_) * 512 + p * 128];                                \
    pi_[p] = prow[(long)(ch_) * 512 + p * 128 + 16];                           \
    br_[p] = HPX_NTLD(brow + (long)(ch_) * 512 + p * 128);                     \
    bm_[p] = HPX_NTLD(brow + (long)(ch_) * 512 + p * 128 + 16);                \
  }
#define HPX_NC_MMA(pr_, pi_, br_, bm_)                                         \
  _Pragma("unroll") for (int p = 0; p < 4; ++p) {                              \
    a1 = mfma64(pr_[p], br_[p], a1);                                           \
    a2 = mfma64(pi_[p], bm_[p], a2);                                           \
    a3 = mfma64(pr_[p] + pi_[p], br_[p] - bm_[p], a3);                         \
  }
    if (t > 0) {
      HPX_NC_LOAD(pr0, pi0, br0, bm0, 0)
      for (int ch = 0; ch < t; ch += 2) {
        const int c1 = min(ch + 1, t - 1);
        HPX_NC_LOAD(pr1, pi1, br1, bm1, c1)
        __builtin_amdgcn_sched_barrier(0);
        HPX_NC_MMA(pr0, pi0, br0, bm0)
        __builtin_amdgcn_sched_barrier(0);
        const int c2 = min(ch + 2, t - 1);
        HPX_NC_LOAD(pr0, pi0, br0, bm0, c2)
        __builtin_amdgcn_sched_barrier(0);
        if (ch + 1 < t) { HPX_NC_MMA(pr1, pi1, br1, bm1) }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
#undef HPX_NC_LOAD
#undef HPX_NC_MMA
    const d4 re_ = -(a1 + a2), im_ = a3 - a1 + a2, dd_ = re_ - im_;
    d4 x1 = {0., 0., 0., 0.}, x2 = x1, x3 = x1;
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      const double pr = Vs[v * 128 + rd_re], pi = Vs[v * 128 + rd_im];
      x1 = mfma64(pr, re_[v], x1);
      x2 = mfma64(pi, im_[v], x2);
      x3 = mfma64(pr + pi, dd_[v], x3);
    }
    double* o_ = X.Lb + HPX_LIDX(rt * 16 + li, c0 + g, X.npad);
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      o_[(4 * v) * 32] = x1[v] + x2[v];
      o_[(4 * v) * 32 + 16] = x1[v] - x2[v] - x3[v];
    }
  }
  __syncthreads();        // this column's stores are complete before the next one reads them
  return bad;
}

template <bool GEN, bool GLDS>
__global__ __launch_bounds__(256, 2) void k_factor_wide(double* __restrict__ L_all, double* __restrict__ Wre_all,
                                                        double* __restrict__ Wim_all, double* __restrict__ Vt_all,
                                                        int32_t* __restrict__ info, const int npad, const int ld,
                                                        const int iter_tag, const hpx_gen_batch GB) {
  extern __shared__ double lds_dyn[];      // 1 / a and the circulant (GLDS), behind the two static staging buffers
  const int b = blockIdx.x;
  const int tid = threadIdx.x;
  hpx_gen G = {};
  if (GEN) G = hpx_gen_for(GB, b);
  WideCtx X;
  X.tid = tid;
  X.wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  X.lane = tid & 63;
  X.npad = npad;
  X.nct = npad >> 4;
  X.nrt = ld >> 4;
  X.ptile = (long)npad * 32;
  X.Lb = L_all + (long)b * npad * ld * 2;
  X.Vt = Vt_all + (long)b * HPX_VT_STRIDE(npad);
  const int nblk = (npad + HPX_NB - 1) / HPX_NB;
  X.Wgre = Wre_all + (long)b * nblk * 1024;
  X.Wgim = Wim_all + (long)b * nblk * 1024;
#ifdef HPX_WIDE_TRACE
  X.tp = hpx_wtrace + ((long)b * 4 + X.wave) * HPX_WTRACE_REC;
  {
    unsigned long long rt_;
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(rt_) :: "memory");
    wtrace(X, (unsigned)rt_);                                                   // record 0: memtime | realtime (low words)
    wtrace(X, __builtin_amdgcn_s_getreg((31 << 11) | 4));                      // record 1: HW_REG_HW_ID
    wtrace(X, __builtin_amdgcn_s_getreg((3 << 11) | 20));                      // record 2: HW_REG_XCC_ID
  }
#endif
  GenVec<GLDS> V = {};
  if constexpr (GEN && GLDS) {
    double* ga = lds_dyn;
    double* gcr = ga + G.N;
    double* gci = gcr + G.N;
    for (int i = tid; i < G.N; i += 256) {
      ga[i] = G.ia[i];
      gcr[i] = G.cre[i];
      gci[i] = G.cim[i];
    }
    V.ia = (const lds_f64*)ga;
    V.cre = (const lds_f64*)gcr;
    V.cim = (const lds_f64*)gci;
    __syncthreads();
  } else if constexpr (GEN) {
    V.ia = (const glb_f64*)G.ia;
    V.cre = (const glb_f64*)G.cre;
    V.cim = (const glb_f64*)G.cim;
  }
  bool bad = false;
  int ct0 = 0;
  for (; ct0 + 8 <= X.nct; ct0 += 8) {
    d4 a1[9], a2[9];
#ifndef HPX_DBG_NO_S
    if (X.wave == 0) diag_update<GEN, GLDS, 0>(X, G, V, ct0, a1, a2);
    else if (X.wave == 1) diag_update<GEN, GLDS, 1>(X, G, V, ct0, a1, a2);
    else if (X.wave == 2) diag_update<GEN, GLDS, 2>(X, G, V, ct0, a1, a2);
    else diag_update<GEN, GLDS, 3>(X, G, V, ct0, a1, a2);
#else
    for (int s = 0; s < 9; ++s) { a1[s] = (d4){1.0 * tid, 0., 0., 0.}; a2[s] = a1[s]; }
#endif
#ifndef HPX_DBG_NO_F
    bad |= diag_factor(X, ct0, a1, a2);
#else
    for (int s = 0; s < 9; ++s) X.Lb[s * 64 + tid] = a1[s][0] + a2[s][1];
#endif
#ifndef HPX_DBG_NO_P
    strip_passes<GEN, GLDS>(X, G, V, ct0);
#endif
  }
  HPX_TR(X, 7, 0, 0, 0);
#ifndef HPX_DBG_NO_N
  for (; ct0 < X.nct; ++ct0) bad |= narrow_column<GEN, GLDS>(X, G, V, ct0);
#endif
  if (bad && info) atomicCAS(&info[b], 0, iter_tag);
#ifdef HPX_WIDE_TRACE
  {
    unsigned long long rt_;
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(rt_) :: "memory");
    wtrace(X, (unsigned)rt_);
    wtrace(X, 0xffffffffu);
    asm volatile("s_dcache_wb\n\ts_waitcnt lgkmcnt(0)" ::: "memory");
  }
#endif
}

template <bool GEN, bool GLDS>
int launch_wide_t(int nbl, size_t lds, int npad, int ld, double* L, double* Wre, double* Wim, double* Vt,
                  int32_t* info, int iter_tag, const hpx_gen_batch& gen, hipStream_t st) {
  static hpx_lds_limit limit;      // per instantiation
  if (lds > 0) HPX_TRY(limit.ensure(reinterpret_cast<const void*>(&k_factor_wide<GEN, GLDS>), lds));
  hipLaunchKernelGGL((k_factor_wide<GEN, GLDS>), dim3(nbl), dim3(256), lds, st, L, Wre, Wim, Vt, info, npad, ld,
                     iter_tag, gen);
  HPX_HIP(hipGetLastError());
  return HPX_OK;
}

}  // namespace

#ifdef HPX_WIDE_TRACE
extern "C" int hpx_debug_wide_trace(unsigned long long* host, int nblocks) {
  HPX_HIP(hipMemcpyFromSymbol(host, HIP_SYMBOL(hpx_wtrace), sizeof(unsigned long long) * (size_t)nblocks * 4 * HPX_WTRACE_REC));
  return HPX_OK;
}
#endif

int hpx_launch_factor_wide(int nbl, int npad, int ld, double* L, double* Wre, double* Wim, double* Vt,
                           int32_t* info, int iter_tag, const hpx_gen_batch* gen, hipStream_t st) {
  if (!gen) {
    hpx_gen_batch none = {};
    return launch_wide_t<false, false>(nbl, 0, npad, ld, L, Wre, Wim, Vt, info, iter_tag, none, st);
  }
  // 1 / a and the circulant in LDS while two workgroups still fit on a CU (64 KB of staging buffers each)
  const size_t staged = (size_t)3 * gen->N * sizeof(double);
  if (2 * BUF_D * sizeof(double) + staged <= (size_t)80 * 1024)
    return launch_wide_t<true, true>(nbl, staged, npad, ld, L, Wre, Wim, Vt, info, iter_tag, *gen, st);
  return launch_wide_t<true, false>(nbl, 0, npad, ld, L, Wre, Wim, Vt, info, iter_tag, *gen, st);
}
