// Tile pass of the split factorisation (gfx950): every 16-row tile BELOW a block column of CT*16
// columns,  X = (K[r, j] - sum_{k < c0} L[r, k] L[j, k]^H) Ljj^-H,  with BOTH MFMA operands staged
// through LDS by LDS-DMA (global_load_lds, no VGPR staging): three buffers, counted vmcnt + raw
// s_barrier, so that the row-tile operand is fetched exactly once per block column and the panel
// operand once per workgroup -- the left-looking 32-wide fused kernel re-read both through the CU's
// memory path.
//
//   workgroup = 4 waves; wave = RT row tiles x CT column tiles (accumulators in registers)
//   k-loop chunk = KC columns: CT panel tiles (shared by the waves) + 4*RTMAX row tiles per buffer
//   complex products are three real MFMAs (Gauss):  conj(p) b = (S1 + S2) + i (S1 - S2 - S3),
//     S1 = pr br, S2 = pi bi, S3 = (pr + pi)(br - bi); a tile keeps a1 = S1 - Kre/2, a2 = S2 - Kre/2,
//     a3 = S3 + Kim, so that  re = -(a1 + a2),  im = a3 - a1 + a2  (no operand negations)
//   after the k-loop, per 32-wide sub-block s of the block column (blocked triangular solve with the
//   32 x 32 inverse diagonal blocks W_ss = conj(Lss^-1), fragment images written by the diagonal-block
//   kernel):  X_s = W_ss a_s (stored), then the later column tiles take  a += conj(L[., s]) X_s  as
//   further k-steps with the B operand in registers.
// Storage is the factor's 16-row panel-major layout (HPX_LIDX).
#pragma once
#include "hpx_internal.h"

namespace hpx_pass {

typedef __attribute__((address_space(3))) double lds_f64;
typedef double v2d __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) v2d lds_v2d;

constexpr int W_FRAG = 1536;   // doubles of one 32 x 32 inverse block's fragment image (12 KB)

__device__ __forceinline__ void glds16(const double* src, double* dst) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                   (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
}
template <int N>
__device__ __forceinline__ void wait_vm() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

template <int CT, int RTMAX, int KC>
struct Cfg {
  static constexpr int PP = KC / 4;                 // 1 KB pieces (4 columns) per tile and chunk
  static constexpr int TILE_CH = KC * 32;           // doubles of one tile's chunk
  static constexpr int NPT = CT / 4;                // panel tiles each wave stages
  static constexpr int BUF_T = CT + 4 * RTMAX;      // tiles per buffer
  static constexpr int BUF_D = BUF_T * TILE_CH;
  static constexpr int NBUF = 3;
  // epilogue operands: CT = 4: W0 W1 L0(2 tiles);  CT = 8: W0 W1 L0(6 tiles) / L1(4) W2 W3 L2(2)
  static constexpr int EPI_D = (CT == 4) ? (2 * W_FRAG + 2 * 1024) : (2 * W_FRAG + 6 * 1024);
  static constexpr int LDS_D = (NBUF * BUF_D > EPI_D) ? NBUF * BUF_D : EPI_D;
  static_assert(CT == 4 || CT == 8, "block columns of 64 or 128");
  static_assert(KC == 4 || KC == 8, "chunks of 4 or 8 columns");
};

// the 32 x 32 block W (row-major planar in LDS, leading dimension ldw; W = conj(L^-1), zero above
// the diagonal) as the fragment image the pass reads: entry ((pair*4 + v)*64 + lane)*2 + {re, im},
// pair (ci, cj) = (0,0), (1,0), (1,1);  value W[16 ci + li][16 cj + g + 4 v]
template <typename P>
__device__ __forceinline__ void write_w_fragments(double* __restrict__ wf, const P* Yre, const P* Yim,
                                                  const int ldw, const int tid, const int nthreads) {
  for (int e = tid; e < 12 * 64; e += nthreads) {
    const int lane = e & 63, pv = e >> 6, pair = pv >> 2, v = pv & 3;
    const int ci = (pair == 0) ? 0 : 1, cj = (pair == 2) ? 1 : 0;
    const int li = lane & 15, g = lane >> 4;
    const int r = 16 * ci + li, c = 16 * cj + g + 4 * v;
    wf[2 * e] = Yre[r * ldw + c];
    wf[2 * e + 1] = Yim[r * ldw + c];
  }
}

// DG: timing-only ablations for tools/pass_probe.hip (wrong results; the library instantiates DG = 0):
// bit 0 no staging inside the k-loop, bit 1 no barriers there, bit 2 no k-loop MFMAs, bit 3 no triangular solve
template <int CT, int RTMAX, int RT, int KC, bool GEN, int DG = 0>
__device__ __forceinline__ void tile_pass(double* __restrict__ Lb, const double* __restrict__ Wf,
                                          double* lds, const int npad, const int c0, const int rt0,
                                          const int wave, const int lane, const bool active,
                                          const hpx_gen& G) {
  typedef Cfg<CT, RTMAX, KC> C;
  constexpr int TILE_CH = C::TILE_CH, PP = C::PP, NPT = C::NPT;
  const int li = lane & 15, g = lane >> 4;
  const long ptile = (long)npad * 32;
  // lane's 16 bytes inside a 1 KB piece (4 columns x [re16 | im16]); odd columns are stored
  // [im | re] so that the two halves of a wave read disjoint banks
  const int src_lane = g * 32 + 2 * ((li + 8 * (g & 1)) & 15);
  const int rd_re = g * 32 + li + 16 * (g & 1), rd_im = g * 32 + li + 16 * (1 - (g & 1));
  d4 a1[RT][CT], a2[RT][CT], a3[RT][CT];
  if (active) {
    const bool use_e = GEN && G.ere != nullptr;
#pragma unroll
    for (int t = 0; t < RT; ++t)
#pragma unroll
      for (int ci = 0; ci < CT; ++ci)
#pragma unroll
        for (int v = 0; v < 4; ++v) {
          const int r = (rt0 + t) * 16 + li, c = c0 + 16 * ci + HPX_ACC_ROW(g, v);
          double vr, vi;
          if (GEN && (rt0 + t) * 16 < G.rmin) hpx_gen_signal(G, r, c, vr, vi);
          else hpx_edge_init<GEN>(G, Lb, Lb + 16, r, c, npad, use_e, vr, vi);
          a1[t][ci][v] = -0.5 * vr;
          a2[t][ci][v] = -0.5 * vr;
          a3[t][ci][v] = vi;
        }
  }
  const double* pan = Lb + (long)((c0 >> 4) + wave * NPT) * ptile + src_lane;
  const double* row = Lb + (long)rt0 * ptile + src_lane;
  const int nch = c0 / KC;
#define HPX_PASS_STAGE(chunk_, bufi_)                                                            \
  {                                                                                              \
    const long ko_ = (long)(chunk_) * TILE_CH;                                                   \
    double* bb_ = lds + (bufi_) * C::BUF_D;                                                      \
    _Pragma("unroll") for (int pt = 0; pt < NPT; ++pt)                                           \
      _Pragma("unroll") for (int pp = 0; pp < PP; ++pp)                                          \
        glds16(pan + pt * ptile + ko_ + pp * 128, bb_ + (wave * NPT + pt) * TILE_CH + pp * 128); \
    _Pragma("unroll") for (int t = 0; t < RT; ++t)                                               \
      _Pragma("unroll") for (int pp = 0; pp < PP; ++pp)                                          \
        glds16(row + t * ptile + ko_ + pp * 128, bb_ + (CT + RTMAX * wave + t) * TILE_CH + pp * 128); \
  }
  if (nch > 0) {
    HPX_PASS_STAGE(0, 0)
    HPX_PASS_STAGE(nch > 1 ? 1 : 0, 1)
    int bi = 0;
    for (int ch = 0; ch < nch; ++ch) {
      if (!(DG & 1)) wait_vm<(NPT + RT) * PP>();   // chunk ch has landed (ch + 1 may be in flight)
      if (!(DG & 2)) __builtin_amdgcn_s_barrier();
      const int nx = min(ch + 2, nch - 1);     // branch-free tail: a harmless re-stage
      int bn = bi + 2;
      if (bn >= C::NBUF) bn -= C::NBUF;
      if (!(DG & 1)) HPX_PASS_STAGE(nx, bn)
      const lds_f64* B = (const lds_f64*)(lds + bi * C::BUF_D);
      if (active && !(DG & 4)) {
#pragma unroll
        for (int s = 0; s < PP; ++s) {
          double br[RT], bm[RT], bd[RT];
#pragma unroll
          for (int t = 0; t < RT; ++t) {
            br[t] = B[(CT + RTMAX * wave + t) * TILE_CH + s * 128 + rd_re];
            bm[t] = B[(CT + RTMAX * wave + t) * TILE_CH + s * 128 + rd_im];
            bd[t] = br[t] - bm[t];
          }
          // panel operand one column tile ahead of its MFMAs (pinned: hoisting all CT reads costs 4 CT VGPRs)
          double pr = B[s * 128 + rd_re], pi = B[s * 128 + rd_im];
#pragma unroll
          for (int ci = 0; ci < CT; ++ci) {
            const double cr = pr, cm = pi, psm = pr + pi;
            if (ci + 1 < CT) {
              pr = B[(ci + 1) * TILE_CH + s * 128 + rd_re];
              pi = B[(ci + 1) * TILE_CH + s * 128 + rd_im];
            }
            if (CT > 4) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int t = 0; t < RT; ++t) {
              a1[t][ci] = mfma64(cr, br[t], a1[t][ci]);
              a2[t][ci] = mfma64(cm, bm[t], a2[t][ci]);
              a3[t][ci] = mfma64(psm, bd[t], a3[t][ci]);
            }
            if (CT > 4) __builtin_amdgcn_sched_barrier(0);
          }
        }
      }
      bi = (bi + 1 == C::NBUF) ? 0 : bi + 1;
    }
  }
#undef HPX_PASS_STAGE
  // ---- blocked triangular solve; block operands over the staging buffers
  wait_vm<0>();
  __builtin_amdgcn_s_barrier();
  const lds_f64* LD = (const lds_f64*)lds;
  const lds_v2d* LD2 = (const lds_v2d*)lds;
  // W_ss fragment image -> lds + off (12 pieces of 1 KB)
#define HPX_PASS_DMA_W(s_, off_)                                                                 \
  for (int p_ = wave; p_ < 12; p_ += 4)                                                          \
    glds16(Wf + (s_) * W_FRAG + p_ * 128 + 2 * lane, lds + (off_) + p_ * 128);
  // columns [c0 + 32 s, + 32) of the block column's own row tiles ci0 .. CT-1 (8 pieces per tile)
#define HPX_PASS_DMA_L(s_, ci0_, off_)                                                           \
  {                                                                                              \
    const double* lb_ = Lb + (long)(c0 >> 4) * ptile + (long)(c0 + 32 * (s_)) * 32 + src_lane;   \
    for (int p_ = wave; p_ < (CT - (ci0_)) * 8; p_ += 4)                                         \
      glds16(lb_ + (long)((ci0_) + (p_ >> 3)) * ptile + (p_ & 7) * 128,                          \
             lds + (off_) + (p_ >> 3) * 1024 + (p_ & 7) * 128);                                  \
  }
  // X_s = W_ss a_s for the column tiles 2s, 2s+1: stored, and kept in a1 (re), a2 (im), a3 (re - im)
#define HPX_PASS_X(s_, woff_)                                                                    \
  if (active) {                                                                                  \
    _Pragma("unroll") for (int t = 0; t < RT; ++t) {                                             \
      _Pragma("unroll") for (int cj = 0; cj < 2; ++cj) {      /* in place: a1 = re, a2 = im, a3 = re + im */ \
        const d4 re_ = -(a1[t][2 * (s_) + cj] + a2[t][2 * (s_) + cj]);                           \
        const d4 im_ = a3[t][2 * (s_) + cj] - a1[t][2 * (s_) + cj] + a2[t][2 * (s_) + cj];       \
        a1[t][2 * (s_) + cj] = re_;                                                              \
        a2[t][2 * (s_) + cj] = im_;                                                              \
        a3[t][2 * (s_) + cj] = re_ + im_;                                                        \
      }                                                                                          \
      _Pragma("unroll") for (int ci = 1; ci >= 0; --ci) {     /* tile 1 first: it overwrites only its own input */ \
        __builtin_amdgcn_sched_barrier(0);                                                       \
        d4 x1 = {0., 0., 0., 0.}, x2 = {0., 0., 0., 0.}, x3 = {0., 0., 0., 0.};                  \
        _Pragma("unroll") for (int cj = 0; cj <= ci; ++cj)                                       \
          _Pragma("unroll") for (int v = 0; v < 4; ++v) {                                        \
            const v2d w_ = LD2[((woff_) >> 1) + ((ci + cj) * 4 + v) * 64 + lane];                \
            x1 = mfma64(w_.x, a1[t][2 * (s_) + cj][v], x1);                                      \
            x2 = mfma64(w_.y, a2[t][2 * (s_) + cj][v], x2);                                      \
            x3 = mfma64(w_.x + w_.y, a3[t][2 * (s_) + cj][v], x3);                               \
          }                                                                                      \
        const d4 xr_ = x1 - x2, xi_ = x3 - x1 - x2;                                              \
        a1[t][2 * (s_) + ci] = xr_;                                                              \
        a2[t][2 * (s_) + ci] = xi_;                                                              \
        a3[t][2 * (s_) + ci] = xr_ - xi_;                                                        \
        double* o_ = Lb + HPX_LIDX((rt0 + t) * 16 + li, c0 + 32 * (s_) + 16 * ci + g, npad);     \
        _Pragma("unroll") for (int v = 0; v < 4; ++v) {                                          \
          o_[(4 * v) * 32] = xr_[v];                                                             \
          o_[(4 * v) * 32 + 16] = xi_[v];                                                        \
        }                                                                                        \
      }                                                                                          \
    }                                                                                            \
  }
  // later column tiles: a[ci] += conj(L[c0 + 16 ci .., c0 + 32 s ..]) X_s  (8 k-steps per tile)
#define HPX_PASS_UPD(s_, loff_)                                                                  \
  if (active) {                                                                                  \
    _Pragma("unroll") for (int ci = 2 * (s_) + 2; ci < CT; ++ci)                                 \
      _Pragma("unroll") for (int cj = 0; cj < 2; ++cj) {                                         \
        __builtin_amdgcn_sched_barrier(0);                                                       \
        _Pragma("unroll") for (int v = 0; v < 4; ++v) {                                          \
          const int o_ = (loff_) + (ci - 2 * (s_) - 2) * 1024 + (4 * cj + v) * 128;              \
          const double pr = LD[o_ + rd_re], pi = LD[o_ + rd_im];                                 \
          const double psm = pr + pi;                                                            \
          _Pragma("unroll") for (int t = 0; t < RT; ++t) {                                       \
            a1[t][ci] = mfma64(pr, a1[t][2 * (s_) + cj][v], a1[t][ci]);                          \
            a2[t][ci] = mfma64(pi, a2[t][2 * (s_) + cj][v], a2[t][ci]);                          \
            a3[t][ci] = mfma64(psm, a3[t][2 * (s_) + cj][v], a3[t][ci]);                         \
          }                                                                                      \
        }                                                                                        \
      }                                                                                          \
  }
  if constexpr ((DG & 8) != 0) {
    if (active) {
#pragma unroll
      for (int t = 0; t < RT; ++t)
#pragma unroll
        for (int ci = 0; ci < CT; ++ci) {
          double* o_ = Lb + HPX_LIDX((rt0 + t) * 16 + li, c0 + 16 * ci + g, npad);
#pragma unroll
          for (int v = 0; v < 4; ++v) {
            o_[(4 * v) * 32] = a1[t][ci][v] + a2[t][ci][v];
            o_[(4 * v) * 32 + 16] = a3[t][ci][v];
          }
        }
    }
  } else if constexpr (CT == 4) {
    HPX_PASS_DMA_W(0, 0)
    HPX_PASS_DMA_W(1, W_FRAG)
    HPX_PASS_DMA_L(0, 2, 2 * W_FRAG)
    wait_vm<0>();
    __builtin_amdgcn_s_barrier();
    HPX_PASS_X(0, 0)
    HPX_PASS_UPD(0, 2 * W_FRAG)
    HPX_PASS_X(1, W_FRAG)
  } else {
    HPX_PASS_DMA_W(0, 0)
    HPX_PASS_DMA_W(1, W_FRAG)
    HPX_PASS_DMA_L(0, 2, 2 * W_FRAG)
    wait_vm<0>();
    __builtin_amdgcn_s_barrier();
    HPX_PASS_X(0, 0)
    HPX_PASS_UPD(0, 2 * W_FRAG)
    HPX_PASS_X(1, W_FRAG)
    wait_vm<0>();                                // (this wave's X stores; the barrier covers the LDS reads)
    __builtin_amdgcn_s_barrier();
    HPX_PASS_DMA_L(1, 4, 0)
    HPX_PASS_DMA_W(2, 4 * 1024)
    HPX_PASS_DMA_W(3, 4 * 1024 + W_FRAG)
    HPX_PASS_DMA_L(2, 6, 4 * 1024 + 2 * W_FRAG)
    wait_vm<0>();
    __builtin_amdgcn_s_barrier();
    HPX_PASS_UPD(1, 0)
    HPX_PASS_X(2, 4 * 1024)
    HPX_PASS_UPD(2, 4 * 1024 + 2 * W_FRAG)
    HPX_PASS_X(3, 4 * 1024 + W_FRAG)
  }
#undef HPX_PASS_DMA_W
#undef HPX_PASS_DMA_L
#undef HPX_PASS_X
#undef HPX_PASS_UPD
}

// Update of a 128 x 128 diagonal block by the columns before it,  D = K[j, j] - sum_{k < c0} L[j, k] L[j, k]^H,
// lower 16 x 16 tiles only, written (unfactored) to the block's place in the factor buffer; the fused
// kernel then factors it in place.  Same staging as the tile pass, but both MFMA operands are the block's
// own row tiles.  The 36 lower tiles are dealt to the 8 waves of two workgroups, at most five each (tile =
// (row tile, column tile); a wave's tiles lie in one or two rows):
//   workgroup 0 (stages tiles 0..7): (7,0-4) | (7,5-7)(1,0-1) | (6,0-4) | (6,5-6)(2,0-2)
//   workgroup 1 (stages tiles 0..5): (5,0-4) | (5,5)(3,0-3)   | (4,0-4) | (0,0)
struct SyrkDeal {
  int n, row[5], col[5], nstage, stage0;      // tiles; panel tiles this wave stages: stage0 .. stage0 + nstage - 1
};
__device__ constexpr SyrkDeal syrk_deal(const int q) {
  switch (q) {
    case 0: return {5, {7, 7, 7, 7, 7}, {0, 1, 2, 3, 4}, 2, 0};
    case 1: return {5, {7, 7, 7, 1, 1}, {5, 6, 7, 0, 1}, 2, 2};
    case 2: return {5, {6, 6, 6, 6, 6}, {0, 1, 2, 3, 4}, 2, 4};
    case 3: return {5, {6, 6, 2, 2, 2}, {5, 6, 0, 1, 2}, 2, 6};
    case 4: return {5, {5, 5, 5, 5, 5}, {0, 1, 2, 3, 4}, 2, 0};
    case 5: return {5, {5, 3, 3, 3, 3}, {5, 0, 1, 2, 3}, 2, 2};
    case 6: return {5, {4, 4, 4, 4, 4}, {0, 1, 2, 3, 4}, 1, 4};
    default: return {1, {0, 0, 0, 0, 0}, {0, 0, 0, 0, 0}, 1, 5};
  }
}
constexpr int SYRK_KC = 8;
constexpr int SYRK_LDS_D = 3 * 8 * SYRK_KC * 32;       // doubles: three buffers of eight tiles

template <int Q, bool GEN>
__device__ __forceinline__ void syrk_wave(double* __restrict__ Lb, double* lds, const int npad, const int c0,
                                          const int lane, const hpx_gen& G) {
  constexpr SyrkDeal D = syrk_deal(Q);
  constexpr int KC = SYRK_KC, PP = KC / 4, TILE_CH = KC * 32, BUF_D = 8 * TILE_CH, NBUF = 3;
  const int li = lane & 15, g = lane >> 4;
  const long ptile = (long)npad * 32;
  const int src_lane = g * 32 + 2 * ((li + 8 * (g & 1)) & 15);
  const int rd_re = g * 32 + li + 16 * (g & 1), rd_im = g * 32 + li + 16 * (1 - (g & 1));
  d4 a1[D.n], a2[D.n], a3[D.n];
#pragma unroll
  for (int i = 0; i < D.n; ++i)
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      const int r = c0 + 16 * D.row[i] + li, c = c0 + 16 * D.col[i] + HPX_ACC_ROW(g, v);
      double vr, vi;
      if (GEN) hpx_gen_entry(G, r, c, npad, vr, vi);
      else { const long off = HPX_LIDX(r, c, npad); vr = Lb[off]; vi = Lb[off + 16]; }
      a1[i][v] = -0.5 * vr;
      a2[i][v] = -0.5 * vr;
      a3[i][v] = vi;
    }
  const double* pan = Lb + (long)((c0 >> 4) + D.stage0) * ptile + src_lane;
  const int nch = c0 / KC;
#define HPX_SYRK_STAGE(chunk_, bufi_)                                                            \
  {                                                                                              \
    const long ko_ = (long)(chunk_) * TILE_CH;                                                   \
    double* bb_ = lds + (bufi_) * BUF_D;                                                         \
    _Pragma("unroll") for (int pt = 0; pt < D.nstage; ++pt)                                      \
      _Pragma("unroll") for (int pp = 0; pp < PP; ++pp)                                          \
        glds16(pan + pt * ptile + ko_ + pp * 128, bb_ + (D.stage0 + pt) * TILE_CH + pp * 128);   \
  }
  if (nch > 0) {
    HPX_SYRK_STAGE(0, 0)
    HPX_SYRK_STAGE(nch > 1 ? 1 : 0, 1)
    int bi = 0;
    for (int ch = 0; ch < nch; ++ch) {
      wait_vm<D.nstage * PP>();
      __builtin_amdgcn_s_barrier();
      const int nx = min(ch + 2, nch - 1);
      int bn = bi + 2;
      if (bn >= NBUF) bn -= NBUF;
      HPX_SYRK_STAGE(nx, bn)
      const lds_f64* B = (const lds_f64*)(lds + bi * BUF_D);
#pragma unroll
      for (int s = 0; s < PP; ++s) {
        double br = 0., bm = 0., bd = 0.;
#pragma unroll
        for (int i = 0; i < D.n; ++i) {
          if (i == 0 || D.row[i] != D.row[i - 1]) {          // (compile time) next row operand
            br = B[D.row[i] * TILE_CH + s * 128 + rd_re];
            bm = B[D.row[i] * TILE_CH + s * 128 + rd_im];
            bd = br - bm;
          }
          const double pr = B[D.col[i] * TILE_CH + s * 128 + rd_re], pi = B[D.col[i] * TILE_CH + s * 128 + rd_im];
          a1[i] = mfma64(pr, br, a1[i]);
          a2[i] = mfma64(pi, bm, a2[i]);
          a3[i] = mfma64(pr + pi, bd, a3[i]);
        }
      }
      bi = (bi + 1 == NBUF) ? 0 : bi + 1;
    }
  }
#undef HPX_SYRK_STAGE
  // D[r][c] = acc^T[c][r]: re = -(a1 + a2), im = a3 - a1 + a2
#pragma unroll
  for (int i = 0; i < D.n; ++i) {
    double* o_ = Lb + HPX_LIDX(c0 + 16 * D.row[i] + li, c0 + 16 * D.col[i] + g, npad);
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      o_[(4 * v) * 32] = -(a1[i][v] + a2[i][v]);
      o_[(4 * v) * 32 + 16] = a3[i][v] - a1[i][v] + a2[i][v];
    }
  }
}
// q = 4 * (workgroup of the pair) + wave
template <bool GEN>
__device__ __forceinline__ void syrk_workgroup(double* __restrict__ Lb, double* lds, const int npad, const int c0,
                                               const int q, const int lane, const hpx_gen& G) {
  switch (q) {
    case 0: syrk_wave<0, GEN>(Lb, lds, npad, c0, lane, G); break;
    case 1: syrk_wave<1, GEN>(Lb, lds, npad, c0, lane, G); break;
    case 2: syrk_wave<2, GEN>(Lb, lds, npad, c0, lane, G); break;
    case 3: syrk_wave<3, GEN>(Lb, lds, npad, c0, lane, G); break;
    case 4: syrk_wave<4, GEN>(Lb, lds, npad, c0, lane, G); break;
    case 5: syrk_wave<5, GEN>(Lb, lds, npad, c0, lane, G); break;
    case 6: syrk_wave<6, GEN>(Lb, lds, npad, c0, lane, G); break;
    default: syrk_wave<7, GEN>(Lb, lds, npad, c0, lane, G); break;
  }
}

// One launch = the tile pass of block column c0 for every baseline: workgroups of one baseline share
// an XCD (its L2 then serves the panel rows they all read); the row tiles first_rt .. nrt-1 are
// spread evenly over the 4 nw waves of the baseline's nw workgroups (at most RTMAX each).
template <int CT, int RTMAX, int KC, bool GEN, int DG = 0>
__device__ __forceinline__ void pass_workgroup(double* __restrict__ Lb, const double* __restrict__ Wf,
                                               double* lds, const int npad, const int c0,
                                               const int first_rt, const int nrt, const int wg,
                                               const int nw, const int wave, const int lane,
                                               const hpx_gen& G) {
  const int nt = nrt - first_rt, nwv = 4 * nw, q = wg * 4 + wave;
  const int base = nt / nwv, extra = nt % nwv;
  const int cnt = base + (q < extra ? 1 : 0);
  const int rt0 = first_rt + q * base + min(q, extra);
  if (RTMAX >= 2 && cnt >= 2)
    tile_pass<CT, RTMAX, RTMAX, KC, GEN, DG>(Lb, Wf, lds, npad, c0, rt0, wave, lane, true, G);
  else      // one tile, or none: an idle wave still stages (a valid tile) and keeps the barriers
    tile_pass<CT, RTMAX, 1, KC, GEN, DG>(Lb, Wf, lds, npad, c0, (cnt <= 0) ? nrt - 1 : rt0, wave, lane, cnt > 0, G);
}

}  // namespace hpx_pass
