// Batched complex-Hermitian blocked Cholesky on FP64 MFMA, with the right-hand
// sides carried as extra rows (forward substitution for free), and the
// backward substitution.  One workgroup (4 waves) per baseline.
//
// Storage ("16-row panel major"): the ld = npad + TP rows are cut into 16-row panels;
// element (r, c) has its real part at HPX_LIDX(r, c, npad) and its imaginary part 16
// doubles further, i.e. panel (r>>4) is a contiguous [npad columns][re 16 | im 16] strip.
// A wave streaming the k range of one row tile therefore reads 256 contiguous bytes per
// column (sequential DRAM pages), instead of 128-byte pieces 8*ld bytes apart as in a
// column-major layout.  Rows npad..ld-1 hold the conjugated right-hand sides, so that
// after the factorisation they hold Z^H with Z = L^-1 R.
//
// Algorithm (left-looking by block columns of HPX_NB = 32):
//   for each block column j:
//     1. diagonal block  D = K[j,j] - sum_k L[j,k] L[j,k]^H   (MFMA, one of its
//        three lower 16x16 tiles per wave, written straight to LDS)
//     2. D = Ljj Ljj^H and Ljj^-1 in LDS (fused right-looking elimination)
//     3. every 16-row tile below:  X = (K[r,j] - sum_k L[r,k] L[j,k]^H) Ljj^-H
//        computed TRANSPOSED so that the accumulator tile is directly the
//        B operand of the multiplication by conj(Ljj^-1): no data movement.
#include "hpx_internal.h"

#define HPX_INL __forceinline__
// Complex products as THREE real MFMAs (Gauss / "3M"): with p = pr + i pi, b = br + i bi,
//   conj(p) b = (S1 + S2) + i (S1 - S2 - S3),  S1 = pr br, S2 = pi bi, S3 = (pr + pi)(br - bi),
// so a tile keeps three accumulators (A1 = Kre/2 - S1, A2 = Kre/2 - S2, A3 = Kim + S3;
// re = A1 + A2, im = A1 - A2 + A3) and a k-step costs 3 MFMAs and one fp64 add per operand
// instead of 4 MFMAs.  Norm-wise as accurate as the four-product form (the imaginary part
// carries an error of a few ulp of |p||b| rather than of |Im|; Higham, "Stability of a method
// for multiplying complex matrices with three real matrix multiplications", 1992), which is
// what a Cholesky factorisation's backward error bound needs.
//
// The tile groups, the diagonal-block update and the elimination are out-of-line functions: each
// gets a register allocation of its own (out-of-line functions get the full 256-VGPR budget).
//
// Cache policy (k_factor sits on the CUs' memory path, docs/HISTORY.md section 9.3): the row-tile (B) operand
// loads of the k-loops are non-temporal -- each tile is streamed once per block column by one wave, and
// the hint keeps it from displacing the panel operand, which every wave of the workgroup re-reads
// (4.63 -> 4.30 ms at C3).  The other streams were measured too and left cached; the variants tried and
// rejected in rounds 1-3 (with their numbers) are under tools/experiments/.
#define HPX_LD(base, off) __builtin_nontemporal_load(&(base)[off])
#define HPX_LDA(base, off) (base)[off]

namespace {

constexpr int WLD = HPX_WLD;

struct FactorShared {
  double part[3][2][4][64];      // next diagonal block: update over the columns done so far (12 KB);
                                 // doubles as scratch of the 16 x 16 elimination (8.6 KB)
  double Dre[32 * WLD], Dim[32 * WLD];   // diagonal block (lower), row-major [r][c]
  double Yre[32 * WLD], Yim[32 * WLD];   // running inverse; finally W = conj(Ljj^-1)
};

// Pointers handed to the out-of-line functions carry their address space: a plain `double*`
// argument is a generic pointer there, and hipcc then emits FLAT loads/stores for the LDS
// traffic of the elimination (several hundred cycles per round trip instead of ~64, and every
// wait becomes vmcnt(0)+lgkmcnt(0)).
typedef __attribute__((address_space(3))) FactorShared lds_FactorShared;
typedef __attribute__((address_space(3))) double lds_f64;
typedef double cplx __attribute__((ext_vector_type(2)));     // (re, im): one 16-byte LDS access
typedef __attribute__((address_space(3))) cplx lds_cplx;
typedef __attribute__((address_space(1))) double glb_f64;
template <bool GLDS> struct gen_ptr { typedef glb_f64 type; };
template <> struct gen_ptr<true> { typedef lds_f64 type; };
template <bool GLDS>
struct gen_signal {             // the part of hpx_gen the diagonal block needs (by value)
  const typename gen_ptr<GLDS>::type *ia, *cre, *cim;
  int rmin;
};

// Tail of a tile group in the three-product form: turn the accumulators into the tiles, multiply by
// W = conj(Ljj^-1) (LDS) and store.
template <int CT, int RT>
__device__ HPX_INL void trsm_store_3m(d4 (&a1)[RT][CT], d4 (&a2)[RT][CT], d4 (&a3)[RT][CT],
                                      double* __restrict__ Lre, double* __restrict__ Lim, const int npad,
                                      const int c0, const int r0, const int rstride, const double* Wre,
                                      const double* Wim, const int lane) {
  const int li = lane & 15, g = lane >> 4;
  // the tiles themselves, in place: a1 <- re, a2 <- im, a3 <- re + im;  then X^T = W acc^T again
  // as three real products per complex one: X1 = wr re, X2 = wi im, X3 = (wr + wi)(re + im);
  // Re X = X1 - X2, Im X = X3 - X1 - X2.  Each 16-column result is stored as soon as it is done.
#pragma unroll
  for (int t = 0; t < RT; ++t)
#pragma unroll
    for (int ci = 0; ci < CT; ++ci) {
      const d4 re_ = a1[t][ci] + a2[t][ci];
      const d4 im_ = a1[t][ci] - a2[t][ci] + a3[t][ci];
      a1[t][ci] = re_;
      a2[t][ci] = im_;
      a3[t][ci] = re_ + im_;
    }
#pragma unroll
  for (int t = 0; t < RT; ++t) {
#pragma unroll
    for (int ci = 0; ci < CT; ++ci) {
      d4 x1 = {0., 0., 0., 0.}, x2 = {0., 0., 0., 0.}, x3 = {0., 0., 0., 0.};
#pragma unroll
      for (int cj = 0; cj <= ci; ++cj)
#pragma unroll
        for (int v = 0; v < 4; ++v) {
          const int kq = 16 * cj + HPX_ACC_ROW(g, v);
          const double wr = Wre[(16 * ci + li) * WLD + kq];
          const double wi = Wim[(16 * ci + li) * WLD + kq];
          x1 = mfma64(wr, a1[t][cj][v], x1);
          x2 = mfma64(wi, a2[t][cj][v], x2);
          x3 = mfma64(wr + wi, a3[t][cj][v], x3);
        }
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        const long off = HPX_LIDX(r0 + t * rstride + li, c0 + 16 * ci + HPX_ACC_ROW(g, v), npad);
        Lre[off] = x1[v] - x2[v];
        Lim[off] = x3[v] - x1[v] - x2[v];
      }
    }
  }
}

// RT off-diagonal 16-row tiles (rows r0 + i*rstride) of block column (c0, CT*16 wide),
// processed together so that the panel operand conj(L[c][k]) is fetched once per k-step
// for all of them (the panel rows are the re-read-heavy operand: without this reuse every
// tile streams the whole 32 x c0 panel again and the kernel becomes HBM/L2 bound).
template <int CT, int RT, bool GEN>
__device__ HPX_INL void offdiag_group(double* __restrict__ Lre, double* __restrict__ Lim,
                                              const int npad, const int c0, const int r0,
                                              const int rstride, const double* Wre,
                                              const double* Wim, const int lane,
                                              const hpx_gen& G) {
  const int li = lane & 15, g = lane >> 4;
  // rows >= rmin of a block column below rmin: straight from the plan's edge tiles when it has them
  const bool use_e = GEN && G.ere != nullptr && c0 + 16 * CT <= G.rmin;
  d4 a1[RT][CT], a2[RT][CT], a3[RT][CT];
  // acc^T[c][r] -= conj(L[c][k]) * L[r][k].  c0 is a multiple of 32, so the k range is a
  // whole number of chunk pairs; operands of the next chunk are fetched into the other
  // register buffer while the current one feeds the MFMAs (explicit double buffering:
  // hipcc does not software-pipeline across loop iterations).
  constexpr int KC = (RT >= 3) ? 1 : 2;     // k-steps per chunk (c0 / 4 is a multiple of 8: KC | 4)
  const int nch = (c0 >> 2) / KC;
  const double* pre = Lre + (long)g * 32;      // column k = 4 ks + g of every panel
  const double* pim = Lim + (long)g * 32;
  const long kstep = 128;                       // 4 columns x 32 doubles
  const long ptile = (long)npad * 32;           // doubles per 16-row panel
  const long boff = (long)(r0 >> 4) * ptile + li, bstr = (long)(rstride >> 4) * ptile;
  const long aoff = (long)(c0 >> 4) * ptile + li;
  double b0r[RT][KC], b0i[RT][KC], b1r[RT][KC], b1i[RT][KC];
  double p0r[CT][KC], p0i[CT][KC], p1r[CT][KC], p1i[CT][KC];
#define HPX_LOAD_CHUNK(br_, bi_, pr_, pi_, base_re, base_im)                     \
  _Pragma("unroll") for (int s = 0; s < KC; ++s) {                               \
    _Pragma("unroll") for (int t = 0; t < RT; ++t) {                             \
      br_[t][s] = HPX_LD((base_re), s * kstep + boff + t * bstr);                \
      bi_[t][s] = HPX_LD((base_im), s * kstep + boff + t * bstr);                \
    }                                                                            \
    _Pragma("unroll") for (int ci = 0; ci < CT; ++ci) {                          \
      pr_[ci][s] = HPX_LDA((base_re), s * kstep + aoff + ci * ptile);            \
      pi_[ci][s] = HPX_LDA((base_im), s * kstep + aoff + ci * ptile);            \
    }                                                                            \
  }
#define HPX_MMA_CHUNK(br_, bi_, pr_, pi_)                                        \
  _Pragma("unroll") for (int s = 0; s < KC; ++s) {                               \
    double bd_[RT];                                                              \
    _Pragma("unroll") for (int t = 0; t < RT; ++t) bd_[t] = br_[t][s] - bi_[t][s]; \
    _Pragma("unroll") for (int ci = 0; ci < CT; ++ci) {                          \
      const double npr = -pr_[ci][s], npi = -pi_[ci][s];                         \
      const double psm = pr_[ci][s] + pi_[ci][s];                                \
      _Pragma("unroll") for (int t = 0; t < RT; ++t) {                           \
        a1[t][ci] = mfma64(npr, br_[t][s], a1[t][ci]);                           \
        a2[t][ci] = mfma64(npi, bi_[t][s], a2[t][ci]);                           \
        a3[t][ci] = mfma64(psm, bd_[t], a3[t][ci]);                              \
      }                                                                          \
    }                                                                            \
  }
  // the first chunk's operands are requested before the accumulators are initialised, so
  // that the two load latencies overlap
  if (nch > 0) {
    HPX_LOAD_CHUNK(b0r, b0i, p0r, p0i, pre, pim)
  }
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int t = 0; t < RT; ++t)
#pragma unroll
    for (int ci = 0; ci < CT; ++ci) {
      if (GEN && r0 + t * rstride < G.rmin) {      // signal x signal tile: closed form
#pragma unroll
        for (int v = 0; v < 4; ++v) {
          double vr, vi;
          hpx_gen_signal(G, r0 + t * rstride + li, c0 + 16 * ci + HPX_ACC_ROW(g, v), vr, vi);
          a1[t][ci][v] = 0.5 * vr;
          a2[t][ci][v] = 0.5 * vr;
          a3[t][ci][v] = vi;
        }
      } else {
#pragma unroll
        for (int v = 0; v < 4; ++v) {
          double vr, vi;
          hpx_edge_init<GEN>(G, Lre, Lim, r0 + t * rstride + li, c0 + 16 * ci + HPX_ACC_ROW(g, v), npad, use_e,
                             vr, vi);
          a1[t][ci][v] = 0.5 * vr;
          a2[t][ci][v] = 0.5 * vr;
          a3[t][ci][v] = vi;
        }
      }
    }
  if (nch > 0) {
    for (int ch = 0; ch < nch; ch += 2) {
      const double* qre = pre + KC * kstep;
      const double* qim = pim + KC * kstep;
      HPX_LOAD_CHUNK(b1r, b1i, p1r, p1i, qre, qim)        // chunk ch+1 always exists (nch even)
      // sched_barrier: without it hipcc hoists the second group of loads up here as well and
      // then waits for loads issued in the SAME iteration (no prefetch distance left).
      __builtin_amdgcn_sched_barrier(0);
      HPX_MMA_CHUNK(b0r, b0i, p0r, p0i)
      __builtin_amdgcn_sched_barrier(0);
      // Branch-free prefetch of chunk ch+2: on the last pair the current chunk is simply
      // fetched again (in bounds, unused).  A conditional here gives the consuming block two
      // predecessors and hipcc then waits with vmcnt(0), draining the prefetch as well.
      const long adv = (ch + 2 < nch) ? 2 * KC * kstep : 0;
      pre += adv;
      pim += adv;
      HPX_LOAD_CHUNK(b0r, b0i, p0r, p0i, pre, pim)
      __builtin_amdgcn_sched_barrier(0);
      HPX_MMA_CHUNK(b1r, b1i, p1r, p1i)
      __builtin_amdgcn_sched_barrier(0);
    }
  }
#undef HPX_LOAD_CHUNK
#undef HPX_MMA_CHUNK
  // X^T = W * acc^T,  W = conj(Ljj^-1) lower triangular (LDS), acc^T as B operand
  trsm_store_3m<CT, RT>(a1, a2, a3, Lre, Lim, npad, c0, r0, rstride, Wre, Wim, lane);
}

// ---- tile groups with the panel operand shared through LDS (HPX_PANEL_LDS) -------------------
// k_factor is bound by what a CU can pull through its memory path (a workgroup alone on a CU
// factors a baseline in 1.3 ms, two share the CU's bandwidth and take 2.3 ms for two; 8 waves on one
// baseline are no faster than 4: round-2 measurements, docs/HISTORY.md section 6), and a third of that
// traffic is the 32 x c0 panel operand, which every wave fetches again for each of its tile groups
// (L2 hit rate 0.24).  Here the four waves of the workgroup take their groups in rounds, sweep the
// k range together in chunks of 16 columns, and the panel chunk (2 tiles x 16 columns = 8 KB) is
// brought in ONCE per workgroup by LDS-DMA (global_load_lds, two 1 KB pieces per wave, no VGPRs),
// double buffered, one s_barrier per chunk.  The row-tile operand still goes global -> registers,
// one k-step ahead.  Odd columns are stored [im | re] so that the two lane halves of a ds_read_b64
// hit disjoint banks.
// number of groups the 3 / 2+2 / 1 rule cuts `cnt` tiles into
__device__ __forceinline__ int hpx_group_count(int cnt) {
  int n = 0;
  while (cnt > 0) {
    if (cnt >= 3 && cnt != 4) cnt -= 3;
    else if (cnt >= 2) cnt -= 2;
    else cnt -= 1;
    ++n;
  }
  return n;
}

// Measured round 2 (C3, 1024 baselines; tools/build_variant.sh + tools/stamps.py): fusing the next
// diagonal block's update into the pass cuts k_factor's FETCH_SIZE by 12 % and the separate partial
// phase from 8.9 % to 1.8 % of a wave's cycles, but the closing barrier and the per-group fixed costs
// take it back (4.69 ms fused vs 4.58); dealing groups by equal COUNT per wave is worse still (5.21):
// both stay available as build switches, off by default.
// HPX_PANEL_LDS (round 2, measured, off): tile groups taken in rounds with the panel operand brought in
// once per workgroup by LDS-DMA (offdiag_group_lds).  Parity-green, FETCH_SIZE down, but the lock-step
// (one s_barrier per 16 columns, counted vmcnt waits that also drain the previous group's stores) costs
// more than the traffic it saves: 5.31 ms against 4.58 at C3, 1.84 against 1.34 ms for a workgroup alone
// on its CU.

template <bool GEN>
__device__ __noinline__ void offdiag_narrow(double* Lre, double* Lim, const int npad, const int c0,
                                            const int r0, const double* Wre, const double* Wim,
                                            const int lane, const hpx_gen& G) {
  offdiag_group<1, 1, GEN>(Lre, Lim, npad, c0, r0, 64, Wre, Wim, lane, G);
}

// K-split partial sums of the (up to) three lower tiles of the diagonal block.
// tile 0 = (c-tile 0, r-tile 0), 1 = (0,1), 2 = (1,1); acc^T[c][r].

// 1/sqrt(d): v_rsq_f64 (24 bits) + two Newton steps (measured 2.2e-16 max relative error,
// tools/rcp64_probe.hip) -- a third of the fp64 ops of sqrt followed by an IEEE division.
__device__ __forceinline__ double rsqrt_nr(const double d) {
  double q = __builtin_amdgcn_rsq(d);
  q = fma(q * 0.5, fma(-d * q, q, 1.0), q);
  q = fma(q * 0.5, fma(-d * q, q, 1.0), q);
  return q;
}

// One 16 x 16 tile of the diagonal block's update, acc^T[c][r] = -sum_{k<c0} conj(L[c][k]) L[r][k],
// over the whole k range on one wave (tile 0 = (c-tile 0, r-tile 0), 1 = (0,1), 2 = (1,1)).
// The three tiles of a 32-wide block go to three waves: nothing to reduce afterwards.  Operands
// are fetched four k-steps at a time into two register sets (the next chunk is in flight while
// the current one is multiplied); the k range is a multiple of 32, so the chunk count is even.
// The bulk of the sum (all columns but the last 32) is formed one block column EARLY, by the
// waves with the fewest tiles in the off-diagonal pass of the previous block column
// (diag_partial_next), and parked in LDS: the critical path between two passes only adds
// the last 32 columns.
template <int TILE>
__device__ __forceinline__ void diag_tile(const glb_f64* __restrict__ Lre,
                                          const glb_f64* __restrict__ Lim, const int npad,
                                          const int c0, const int k0, const int k1, const int lane,
                                          d4& ar, d4& ai) {
  // rows c0.. of the block, columns k0 <= k < k1 (both multiples of 32); ar/ai are added to
  const int li = lane & 15, g = lane >> 4;
  const int nch = ((k1 - k0) >> 4);   // chunks of 4 k-steps (16 columns)
  if (nch == 0) return;
  const glb_f64* pAr = Lre + HPX_LIDX(c0 + (TILE == 2 ? 16 : 0) + li, k0 + g, npad);
  const glb_f64* pAi = Lim + HPX_LIDX(c0 + (TILE == 2 ? 16 : 0) + li, k0 + g, npad);
  const glb_f64* pBr = Lre + HPX_LIDX(c0 + 16 + li, k0 + g, npad);
  const glb_f64* pBi = Lim + HPX_LIDX(c0 + 16 + li, k0 + g, npad);
  double xr[2][4], xi[2][4], yr[2][4], yi[2][4];
#define HPX_DT_LD(p_, o_) (p_)[o_]
#define HPX_DT_LOAD(buf, ch)                                   \
  _Pragma("unroll") for (int u = 0; u < 4; ++u) {              \
    const int o_ = ((ch) * 16 + 4 * u) << 5;                   \
    xr[buf][u] = HPX_DT_LD(pAr, o_);                           \
    xi[buf][u] = HPX_DT_LD(pAi, o_);                           \
    if (TILE == 1) {                                           \
      yr[buf][u] = HPX_DT_LD(pBr, o_);                         \
      yi[buf][u] = HPX_DT_LD(pBi, o_);                         \
    }                                                          \
  }
  d4 q1 = {0., 0., 0., 0.}, q2 = {0., 0., 0., 0.}, q3 = {0., 0., 0., 0.};   // -S1, -S2, S3 (see HPX_3M)
#define HPX_DT_MUL(buf)                                        \
  _Pragma("unroll") for (int u = 0; u < 4; ++u) {              \
    const double cr_ = xr[buf][u], ci_ = xi[buf][u];           \
    const double rr_ = (TILE == 1) ? yr[buf][u] : cr_;         \
    const double ri_ = (TILE == 1) ? yi[buf][u] : ci_;         \
    q1 = mfma64(-cr_, rr_, q1);                                \
    q2 = mfma64(-ci_, ri_, q2);                                \
    q3 = mfma64(cr_ + ci_, rr_ - ri_, q3);                     \
  }
  HPX_DT_LOAD(0, 0);
  for (int ch = 0; ch < nch; ch += 2) {
    HPX_DT_LOAD(1, ch + 1);
    __builtin_amdgcn_sched_barrier(0);
    HPX_DT_MUL(0);
    __builtin_amdgcn_sched_barrier(0);
    const int nx = (ch + 2 < nch) ? ch + 2 : ch;              // branch-free tail: harmless re-read
    HPX_DT_LOAD(0, nx);
    __builtin_amdgcn_sched_barrier(0);
    HPX_DT_MUL(1);
    __builtin_amdgcn_sched_barrier(0);
  }
#undef HPX_DT_LOAD
#undef HPX_DT_MUL
  ar += q1 + q2;
  ai += q1 - q2 + q3;
}

// Tile `t` of the diagonal block at c1, summed over the columns k < kend that are final already.
__device__ __noinline__ void diag_partial_next(const glb_f64* __restrict__ Lre,
                                              const glb_f64* __restrict__ Lim,
                                              lds_FactorShared* __restrict__ shp, const int npad,
                                              const int c1, const int kend, const int t,
                                              const int lane) {
  d4 ar = {0., 0., 0., 0.}, ai = {0., 0., 0., 0.};
  if (t == 0) diag_tile<0>(Lre, Lim, npad, c1, 0, kend, lane, ar, ai);
  else if (t == 1) diag_tile<1>(Lre, Lim, npad, c1, 0, kend, lane, ar, ai);
  else diag_tile<2>(Lre, Lim, npad, c1, 0, kend, lane, ar, ai);
#pragma unroll
  for (int v = 0; v < 4; ++v) {
    shp->part[t][0][v][lane] = ar[v];
    shp->part[t][1][v][lane] = ai[v];
  }
}

// Diagonal block at c0 (width wj = 16 or 32): D = K[j,j] - sum_{k<c0} L[j,k] L[j,k]^H
// (one 16 x 16 tile per wave), then D = Ljj Ljj^H and Ljj^-1 by the fused in-LDS elimination.
// On return (after the trailing barrier) Yre/Yim hold W = conj(Ljj^-1); Ljj and Ljj^-1 are
// in global memory.
template <bool GEN, bool GLDS>
__device__ __noinline__ bool diag_panel(glb_f64* __restrict__ Lre, glb_f64* __restrict__ Lim,
                                           glb_f64* __restrict__ Wgre, glb_f64* __restrict__ Wgim,
                                           lds_FactorShared* __restrict__ shp,
                                           const int npad, const int c0, const int wj,
                                           const int tid, const gen_signal<GLDS> G) {
  const bool act = tid < 256;
  lds_FactorShared& sh = *shp;
  lds_f64* const Yre = sh.Yre;
  lds_f64* const Yim = sh.Yim;
  // scratch of the elimination, over `part` (dead between the reads below and the next pass)
  lds_cplx* const Dm = reinterpret_cast<lds_cplx*>(&sh.part[0][0][0][0]);
  lds_cplx* const Ym = Dm + 16 * 17;
  lds_f64* const dg = reinterpret_cast<lds_f64*>(Ym + 16 * 17);
  const int wave = tid >> 6, lane = tid & 63;
  const int CT = wj >> 4;
  bool bad = false;
  d4 ar = {0., 0., 0., 0.}, ai = {0., 0., 0., 0.};
  if (c0 > 0 && wave < (CT == 2 ? 3 : 1)) {
    const int kb = c0 - 32;
    if (kb > 0) {
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        ar[v] = sh.part[wave][0][v][lane];
        ai[v] = sh.part[wave][1][v][lane];
      }
    }
    if (wave == 0) diag_tile<0>(Lre, Lim, npad, c0, kb, c0, lane, ar, ai);
    else if (wave == 1) diag_tile<1>(Lre, Lim, npad, c0, kb, c0, lane, ar, ai);
    else diag_tile<2>(Lre, Lim, npad, c0, kb, c0, lane, ar, ai);
  }
  // the elimination is a chain of dependent fp64 vector ops that queue behind the co-resident
  // workgroup's 64-cycle MFMAs on the shared DP pipe: let them go first
  __builtin_amdgcn_s_setprio(2);
  for (int e = tid; act && e < 32 * 32; e += 256) {     // identity for the running inverse
    const int i = e >> 5, q = e & 31;
    Yre[i * WLD + q] = (i == q) ? 1.0 : 0.0;
    Yim[i * WLD + q] = 0.0;
  }
  // D[r][c] = K[r][c] + acc^T[c][r], written by the wave that owns the tile
  if (wave < (CT == 2 ? 3 : 1)) {
    const int ci = (wave == 2) ? 1 : 0, ri = (wave == 0) ? 0 : 1;
    const int r = 16 * ri + (lane & 15);
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      const int c = 16 * ci + HPX_ACC_ROW(lane >> 4, v);
      double kr, ki;
      if (GEN && c0 + 32 <= G.rmin) {
        if (r > c) {
          kr = G.cre[r - c];
          ki = G.cim[r - c];
        } else { const double ic = G.ia[c0 + c]; kr = fma(ic, ic, G.cre[0]); ki = 0.0; }
      } else {
        const long off = HPX_LIDX(c0 + r, c0 + c, npad);
        kr = Lre[off];
        ki = Lim[off];
      }
      sh.Dre[r * WLD + c] = kr + ar[v];
      sh.Dim[r * WLD + c] = ki + ai[v];
    }
  }
  // fused Cholesky + inverse of the wj x wj block (unscaled columns; column q of L is
  // D[:,q]/sqrt(D[q][q]) and row i of L^-1 is Y[i,:]/sqrt(D[i][i]))
  // Blocked elimination of the wj x wj block as 2 x 2 blocks of 16 (wj = 16: one block):
  //   A. fused Cholesky + inverse of D00 on the vector ALU (16 steps, one D and one Y entry per
  //      thread in registers; only column k of D and row k of Y go through LDS each step)
  //   B. L10 = D10 L00^-H            C. D11 -= L10 L10^H          (MFMA, wave 0, operands in LDS)
  //   D. as A for D11                E. W10 = -W11 L10 W00         (MFMA, wave 0)
  // fp64 vector FMAs issue at ~10 cycles per wave: a 32-step scalar elimination of the whole
  // 32 x 32 block costs ~1e5 cycles, this form ~2e4.
  __syncthreads();                         // the combined block is complete
  {
    const int q = tid & 15, ib = tid >> 4;
    const int lane_ = tid & 63, li = lane_ & 15, g = lane_ >> 4;
    const int nhalf = wj >> 4;
    for (int hb = 0; hb < nhalf; ++hb) {
      const int o = 16 * hb;                // block offset inside the 32 x 32 block
      if (hb == 1 && wave == 0) {
        // ---- B: L10^T[c][r] = sum_c' conj(W00)[c][c'] D10[r][c']   (Y holds conj(W00))
        d4 xr = {0., 0., 0., 0.}, xi = {0., 0., 0., 0.};
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) {
          const int kk = 4 * s4 + g;
          const double wr = Yre[li * WLD + kk], wi = Yim[li * WLD + kk];
          const double br = sh.Dre[(16 + li) * WLD + kk], bi = sh.Dim[(16 + li) * WLD + kk];
          xr = mfma64(wr, br, xr);
          xr = mfma64(-wi, bi, xr);
          xi = mfma64(wr, bi, xi);
          xi = mfma64(wi, br, xi);
        }
        // L10[r][c] -> LDS (over D10) and global
#pragma unroll
        for (int v = 0; v < 4; ++v) {
          const int c = HPX_ACC_ROW(g, v);
          sh.Dre[(16 + li) * WLD + c] = xr[v];
          sh.Dim[(16 + li) * WLD + c] = xi[v];
          const long off = HPX_LIDX(c0 + 16 + li, c0 + c, npad);
          Lre[off] = xr[v];
          Lim[off] = xi[v];
        }
        // ---- C: D11^T[c][r] -= sum_k conj(L10[c][k]) L10[r][k]  (B operand = L10^T accumulators)
        d4 dr4, di4;
#pragma unroll
        for (int v = 0; v < 4; ++v) {
          dr4[v] = sh.Dre[(16 + li) * WLD + 16 + HPX_ACC_ROW(g, v)];
          di4[v] = sh.Dim[(16 + li) * WLD + 16 + HPX_ACC_ROW(g, v)];
        }
        __builtin_amdgcn_s_waitcnt(0xc07f);   // lgkmcnt(0): L10 stores above are visible to this wave
#pragma unroll
        for (int v = 0; v < 4; ++v) {
          const int kk = HPX_ACC_ROW(g, v);
          const double pr = sh.Dre[(16 + li) * WLD + kk], pi = sh.Dim[(16 + li) * WLD + kk];   // L10[c=li][k]
          dr4 = mfma64(-pr, xr[v], dr4);
          dr4 = mfma64(-pi, xi[v], dr4);
          di4 = mfma64(-pr, xi[v], di4);
          di4 = mfma64(pi, xr[v], di4);
        }
#pragma unroll
        for (int v = 0; v < 4; ++v) {
          sh.Dre[(16 + li) * WLD + 16 + HPX_ACC_ROW(g, v)] = dr4[v];
          sh.Dim[(16 + li) * WLD + 16 + HPX_ACC_ROW(g, v)] = di4[v];
        }
      }
      if (hb == 1) __syncthreads();
      // ---- A / D: fused Cholesky + inverse of the 16 x 16 block at (o, o)
      // Every thread stores its D and Y entry into LDS matrices each step, and D is kept
      // STRICTLY lower there (the diagonal goes to a side vector): retired rows and columns
      // then read back as zeros and a step needs no per-step masks, compares or selects --
      // 13 fp64 vector ops and nothing else on the vector ALU, which a co-resident workgroup
      // streaming MFMAs leaves only scraps of (tools/elim_step_probe.hip).  Retired entries
      // never change, so one copy of the matrices is enough (a wave that runs ahead rewrites
      // what the others still read with the same values).
      const bool dia = act && (q == ib), low = act && (q < ib);
      double dr = 0.0, di = 0.0;
      if (act) {
        dr = sh.Dre[(o + ib) * WLD + o + q];
        di = sh.Dim[(o + ib) * WLD + o + q];
        if (!low) Dm[ib * 17 + q] = (cplx){0.0, 0.0};      // diagonal and above stay zero in LDS
      }
      double yr = (ib == q) ? 1.0 : 0.0, yi = 0.0;
      const int nsteps = 16;
#pragma unroll
      for (int k = 0; k < nsteps; ++k) {
        if (dia) dg[ib] = dr; else if (low) Dm[ib * 17 + q] = (cplx){dr, di};
        if (act) Ym[ib * 17 + q] = (cplx){yr, yi};
        __syncthreads();
        if (!act) continue;
        const double dkk = dg[k];
        const cplx c = Dm[ib * 17 + k], cq = Dm[q * 17 + k], sy = Ym[k * 17 + q];
        // v_rcp_f64 (24 bits) + one Newton step (2e-15) instead of the IEEE division sequence
        const double r0 = __builtin_amdgcn_rcp(dkk);
        const double rinv = fma(r0, fma(-dkk, r0, 1.0), r0);
        const double lr = c.x * rinv, lm = c.y * rinv;
        yr = fma(-lr, sy.x, yr);
        yr = fma(lm, sy.y, yr);
        yi = fma(-lr, sy.y, yi);
        yi = fma(-lm, sy.x, yi);
        dr = fma(-lr, cq.x, dr);        // threads above the diagonal compute junk that is
        dr = fma(-lm, cq.y, dr);        // never stored: cheaper than masking the update
        di = fma(-lm, cq.x, di);
        di = fma(lr, cq.y, di);
      }
      if (nsteps == 0 && dia) dg[ib] = dr;
      __syncthreads();
      // scaling: L block to global, W = conj(L^-1) block to LDS (Y), L^-1 block to the side buffer
      if (act) {
        double wr = 0.0, wi = 0.0;
        if (q <= ib) {
          const double pq = dg[q], pib = dg[ib];
          if (!(pq > 0.0) || !(pib > 0.0)) bad = true;
          const double sq = rsqrt_nr(pq);
          const long off = HPX_LIDX(c0 + o + ib, c0 + o + q, npad);
          Lre[off] = dr * sq;
          Lim[off] = (ib == q) ? 0.0 : di * sq;
          const double sv = rsqrt_nr(pib);
          wr = yr * sv;
          wi = yi * sv;
        }
        Wgre[(o + ib) * 32 + o + q] = wr;
        Wgim[(o + ib) * 32 + o + q] = wi;
        Yre[(o + ib) * WLD + o + q] = wr;       // LDS copy is conjugated
        Yim[(o + ib) * WLD + o + q] = -wi;
        if (hb == 0) {                           // upper-right block of the inverse is zero
          Wgre[ib * 32 + 16 + q] = 0.0;
          Wgim[ib * 32 + 16 + q] = 0.0;
          Yre[ib * WLD + 16 + q] = 0.0;
          Yim[ib * WLD + 16 + q] = 0.0;
          if (nhalf == 1) {                      // 16-wide block: nothing below either
            Wgre[(16 + ib) * 32 + q] = 0.0; Wgim[(16 + ib) * 32 + q] = 0.0;
            Wgre[(16 + ib) * 32 + 16 + q] = 0.0; Wgim[(16 + ib) * 32 + 16 + q] = 0.0;
            Yre[(16 + ib) * WLD + q] = 0.0; Yim[(16 + ib) * WLD + q] = 0.0;
            Yre[(16 + ib) * WLD + 16 + q] = 0.0; Yim[(16 + ib) * WLD + 16 + q] = 0.0;
          }
        }
      }
      __syncthreads();
    }
    if (nhalf == 2 && wave == 0) {
      // ---- E: W10 = -W11 (L10 W00).  T[r][c'] = sum_k L10[r][k] W00[k][c'], W00 = conj(Y00)
      d4 tr = {0., 0., 0., 0.}, ti = {0., 0., 0., 0.};
#pragma unroll
      for (int s4 = 0; s4 < 4; ++s4) {
        const int kk = 4 * s4 + g;
        const double pr = sh.Dre[(16 + li) * WLD + kk], pi = sh.Dim[(16 + li) * WLD + kk];      // L10[r=li][k]
        const double wr = Yre[kk * WLD + li], wi = -Yim[kk * WLD + li];                        // W00[k][c'=li]
        tr = mfma64(pr, wr, tr);
        tr = mfma64(-pi, wi, tr);
        ti = mfma64(pr, wi, ti);
        ti = mfma64(pi, wr, ti);
      }
      // W10[i][c'] = -sum_r W11[i][r] T[r][c'],  W11 = conj(Y11); B operand = T accumulators
      d4 zr = {0., 0., 0., 0.}, zi = {0., 0., 0., 0.};
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        const int rr = HPX_ACC_ROW(g, v);
        const double wr = Yre[(16 + li) * WLD + 16 + rr], wi = -Yim[(16 + li) * WLD + 16 + rr];   // W11[i=li][r]
        zr = mfma64(-wr, tr[v], zr);
        zr = mfma64(wi, ti[v], zr);
        zi = mfma64(-wr, ti[v], zi);
        zi = mfma64(-wi, tr[v], zi);
      }
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        const int i = HPX_ACC_ROW(g, v);
        Wgre[(16 + i) * 32 + li] = zr[v];
        Wgim[(16 + i) * 32 + li] = zi[v];
        Yre[(16 + i) * WLD + li] = zr[v];
        Yim[(16 + i) * WLD + li] = -zi[v];
      }
    }
  }
  __syncthreads();
  __builtin_amdgcn_s_setprio(0);
  return bad;
}

// GLDS: the per-baseline vectors the closed-form entries are made of (1/a, circ: 3 N doubles)
// are staged in LDS behind FactorShared.  From global memory every tile group's initialisation
// waits for loads that miss L1 and L2 (the factor streams through both), ~1e4 cycles a group.
template <bool GEN, bool GLDS>
__global__ __launch_bounds__(256, 2) void k_factor(double* __restrict__ L_all, double* __restrict__ Wre_all,
                                                   double* __restrict__ Wim_all, int32_t* __restrict__ info,
                                                   const int npad, const int ld, const int iter_tag,
                                                   const hpx_gen_batch GB) {
  constexpr int NW = 4;
  extern __shared__ double lds_raw[];
  FactorShared& sh = *reinterpret_cast<FactorShared*>(lds_raw);
  const int b = blockIdx.x;
  hpx_gen G;
  if (GEN) G = hpx_gen_for(GB, b);
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  // which wave takes the first tile below the diagonal block (the one that ends up with the
  // extra tile) rotates with the block column, two steps apart for the two workgroups that
  // share a CU's SIMDs (dispatch puts blocks b and b + 256 on the same CU)
  const int rot0 = 2 * ((b >> 8) & 1);
  double* Lre = L_all + (long)b * npad * ld * 2;
  double* Lim = Lre + 16;
  const int nblk = (npad + HPX_NB - 1) / HPX_NB;
  double* Wgre = Wre_all + (long)b * nblk * 1024;
  double* Wgim = Wim_all + (long)b * nblk * 1024;
  const int nrt = ld >> 4;
  bool bad = false;
  typedef typename gen_ptr<GLDS>::type gsrc;
  gen_signal<GLDS> GS = {nullptr, nullptr, nullptr, 0};
  if (GEN) {
    if (GLDS) {
      double* ga = lds_raw + sizeof(FactorShared) / sizeof(double);
      double* gcr = ga + G.N;
      double* gci = gcr + G.N;
      for (int i = tid; i < G.N; i += NW * 64) {
        ga[i] = G.ia[i];
        gcr[i] = G.cre[i];
        gci[i] = G.cim[i];
      }
      G.ia = ga;
      G.cre = gcr;
      G.cim = gci;
      __syncthreads();
    }
    GS = {(const gsrc*)G.ia, (const gsrc*)G.cre, (const gsrc*)G.cim, G.rmin};
  }

  int c0 = 0;
  // ---- 32-wide block columns (the last one may be 16 wide)
  while (c0 < npad) {
    const int wj = min(HPX_NB, npad - c0);
    const int jb = c0 >> 5;
    bad |= diag_panel<GEN, GLDS>((glb_f64*)Lre, (glb_f64*)Lim, (glb_f64*)(Wgre + jb * 1024),
                                 (glb_f64*)(Wgim + jb * 1024), (lds_FactorShared*)&sh, npad, c0, wj, tid,
                                 GS);
    // tiles below the diagonal block (incl. the right-hand-side rows)
    const int pos = (wave + rot0 + jb) & (NW - 1);
    int rt = ((c0 + wj) >> 4) + pos;
    if (wj == 32) {
      const int c1 = c0 + 32;
      // this wave's tiles rt + NW i in groups of 3 (2 + 2 rather than 3 + 1: a single-tile
      // pass costs nearly as much as a three-tile one, its k-loop is bound by load latency)
      int cnt = (nrt - rt + NW - 1) / NW;
      if (nrt <= rt) cnt = 0;
      while (cnt > 0) {
        if (cnt >= 3 && cnt != 4) {
          offdiag_group<2, 3, GEN>(Lre, Lim, npad, c0, rt << 4, 16 * NW, sh.Yre, sh.Yim, lane, G);
          rt += 3 * NW;
          cnt -= 3;
        } else if (cnt >= 2) {
          offdiag_group<2, 2, GEN>(Lre, Lim, npad, c0, rt << 4, 16 * NW, sh.Yre, sh.Yim, lane, G);
          rt += 2 * NW;
          cnt -= 2;
        } else {
          offdiag_group<2, 1, GEN>(Lre, Lim, npad, c0, rt << 4, 16 * NW, sh.Yre, sh.Yim, lane, G);
          rt += NW;
          cnt -= 1;
        }
      }
      // early part of the next diagonal block (columns < c0): its tiles go to the waves that got the
      // fewest off-diagonal tiles in this pass
      if (c0 > 0 && c1 < npad) {
        const int t = NW - 1 - pos;
        if (t < ((npad - c1 >= 32) ? 3 : 1)) {
          diag_partial_next((const glb_f64*)Lre, (const glb_f64*)Lim, (lds_FactorShared*)&sh, npad, c1, c0,
                            t, lane);
        }
      }
    } else {
      for (; rt < nrt; rt += NW) offdiag_narrow<GEN>(Lre, Lim, npad, c0, rt << 4, sh.Yre, sh.Yim, lane, G);
    }
    __syncthreads();
    c0 += wj;
  }
  if (bad && info) atomicCAS(&info[b], 0, iter_tag);
}

// ---------------------------------------------------------------------------
// Backward substitution, second form (round 2): every element of L is fetched by exactly one wave
// and the solution X by one super-block pass instead of one pass per block column.
//
// The block columns are taken four at a time (a "super-block" of 128 columns); wave w owns block
// column 4 J + w and BOTH t-tiles of a pair (two accumulator sets against one L operand), and runs
// over all rows below the super-block on its own (phase A: no k-split, nothing to reduce, no LDS).
// Inside the super-block (phase B) the four block columns are finished from the last to the first:
// the owner multiplies by its inverse diagonal block and stores its 32 rows of X, and after a
// barrier the waves to its left add those rows' contribution (two 16-row chunks).  Against the
// first form (k_backsolve_v1: one pass over the rows below per 32-wide block column, t-tiles on
// different waves) this reads X n/128 instead of n/32 times and never streams a tile of L through
// two waves.  Complex products are three real MFMAs each (HPX_3M, see the top of this file):
// A1 = Zr/2 - S1, A2 = Zr/2 - S2, A3 = Zi + S3 with S1 = lr xr, S2 = lm xi, S3 = (lr + lm)(xr - xi).
// the factor is streamed exactly once by the back substitution: non-temporal loads (HPX_NT bit 4) keep it
// from displacing X, which is re-read
typedef double hpx_v2d __attribute__((ext_vector_type(2)));
#define HPX_BS_LD2(p_) (*reinterpret_cast<const double2*>(p_))
template <int CT, int NT>
__device__ __forceinline__ void bs_accumulate(d4 (&a1)[2][2], d4 (&a2)[2][2], d4 (&a3)[2][2],
                                              const double* __restrict__ Lre, const double* __restrict__ Lim,
                                              const double* __restrict__ Xre, const double* __restrict__ Xim,
                                              const int npad, const int TP, const int c0, const int t0,
                                              const int rbeg, const int nch, const int lane) {
  if (nch <= 0) return;
  const int li = lane & 15, g = lane >> 4;
  double lr0[CT][4], lm0[CT][4], lr1[CT][4], lm1[CT][4];
  double xr0[NT][4], xi0[NT][4], xr1[NT][4], xi1[NT][4];
#define HPX_BS_LOAD(lr, lm, xr_, xi_, ch_)                                               \
  {                                                                                      \
    const int rb_ = rbeg + ((ch_) << 4);                                                 \
    _Pragma("unroll") for (int ci = 0; ci < CT; ++ci) {                                  \
      const long off = HPX_LIDX(rb_ + 4 * g, c0 + 16 * ci + li, npad);                   \
      const double2 q0 = HPX_BS_LD2(Lre + off);                                          \
      const double2 q1 = HPX_BS_LD2(Lre + off + 2);                                      \
      const double2 q2 = HPX_BS_LD2(Lim + off);                                          \
      const double2 q3 = HPX_BS_LD2(Lim + off + 2);                                      \
      lr[ci][0] = q0.x; lr[ci][1] = q0.y; lr[ci][2] = q1.x; lr[ci][3] = q1.y;            \
      lm[ci][0] = q2.x; lm[ci][1] = q2.y; lm[ci][2] = q3.x; lm[ci][3] = q3.y;            \
    }                                                                                    \
    _Pragma("unroll") for (int tt = 0; tt < NT; ++tt)                                    \
      _Pragma("unroll") for (int s = 0; s < 4; ++s) {                                    \
        const long xo = (long)(rb_ + 4 * g + s) * TP + t0 + 16 * tt + li;                \
        xr_[tt][s] = Xre[xo];                                                            \
        xi_[tt][s] = Xim[xo];                                                            \
      }                                                                                  \
  }
#define HPX_BS_MMA(lr, lm, xr_, xi_)                                                     \
  _Pragma("unroll") for (int s = 0; s < 4; ++s) {                                        \
    double xd_[NT];                                                                      \
    _Pragma("unroll") for (int tt = 0; tt < NT; ++tt) xd_[tt] = xr_[tt][s] - xi_[tt][s]; \
    _Pragma("unroll") for (int ci = 0; ci < CT; ++ci) {                                  \
      const double nlr = -lr[ci][s], nlm = -lm[ci][s], lsm = lr[ci][s] + lm[ci][s];      \
      _Pragma("unroll") for (int tt = 0; tt < NT; ++tt) {                                \
        a1[tt][ci] = mfma64(nlr, xr_[tt][s], a1[tt][ci]);                                \
        a2[tt][ci] = mfma64(nlm, xi_[tt][s], a2[tt][ci]);                                \
        a3[tt][ci] = mfma64(lsm, xd_[tt], a3[tt][ci]);                                   \
      }                                                                                  \
    }                                                                                    \
  }
  HPX_BS_LOAD(lr0, lm0, xr0, xi0, 0)
  for (int ch = 0; ch + 1 < nch; ch += 2) {
    HPX_BS_LOAD(lr1, lm1, xr1, xi1, ch + 1)
    __builtin_amdgcn_sched_barrier(0);
    HPX_BS_MMA(lr0, lm0, xr0, xi0)
    __builtin_amdgcn_sched_barrier(0);
    const int nx = min(ch + 2, nch - 1);                 // branch-free: a harmless re-read at the end
    HPX_BS_LOAD(lr0, lm0, xr0, xi0, nx)
    __builtin_amdgcn_sched_barrier(0);
    HPX_BS_MMA(lr1, lm1, xr1, xi1)
    __builtin_amdgcn_sched_barrier(0);
  }
  if (nch & 1) HPX_BS_MMA(lr0, lm0, xr0, xi0)            // the odd chunk sits in set 0
#undef HPX_BS_LOAD
#undef HPX_BS_MMA
}

template <int CT, int NT>
__device__ __forceinline__ void bs_init(d4 (&a1)[2][2], d4 (&a2)[2][2], d4 (&a3)[2][2],
                                        const double* __restrict__ Lre, const double* __restrict__ Lim,
                                        const int npad, const int c0, const int t0, const int lane) {
  const int li = lane & 15, g = lane >> 4;
#pragma unroll
  for (int tt = 0; tt < NT; ++tt)
#pragma unroll
    for (int ci = 0; ci < CT; ++ci)
#pragma unroll
      for (int v = 0; v < 4; ++v) {      // Z[c][t] = conj(Laug[npad + t][c])
        const long off = HPX_LIDX(npad + t0 + 16 * tt + li, c0 + 16 * ci + HPX_ACC_ROW(g, v), npad);
        const double zr = Lre[off];
        a1[tt][ci][v] = 0.5 * zr;
        a2[tt][ci][v] = 0.5 * zr;
        a3[tt][ci][v] = -Lim[off];
      }
}

// X[c][t] = sum_{c' >= c in the block} conj(Linv[c'][c]) Y[c'][t], stored to X
template <int CT, int NT>
__device__ __forceinline__ void bs_finish(d4 (&a1)[2][2], d4 (&a2)[2][2], d4 (&a3)[2][2],
                                          const double* __restrict__ Wgre, const double* __restrict__ Wgim,
                                          double* __restrict__ Xre, double* __restrict__ Xim, const int TP,
                                          const int c0, const int t0, const int lane) {
  const int li = lane & 15, g = lane >> 4;
#pragma unroll
  for (int tt = 0; tt < NT; ++tt) {
    d4 yr[CT], yi[CT];
#pragma unroll
    for (int ci = 0; ci < CT; ++ci) {
      yr[ci] = a1[tt][ci] + a2[tt][ci];
      yi[ci] = a1[tt][ci] - a2[tt][ci] + a3[tt][ci];
    }
#pragma unroll
    for (int ci = 0; ci < CT; ++ci) {
      d4 xr = {0., 0., 0., 0.}, xi = {0., 0., 0., 0.};
#pragma unroll
      for (int cj = ci; cj < CT; ++cj)
#pragma unroll
        for (int v = 0; v < 4; ++v) {
          const int kq = 16 * cj + HPX_ACC_ROW(g, v);
          const double wr = Wgre[kq * 32 + 16 * ci + li], wi = Wgim[kq * 32 + 16 * ci + li];
          xr = mfma64(wr, yr[cj][v], xr);
          xr = mfma64(wi, yi[cj][v], xr);
          xi = mfma64(wr, yi[cj][v], xi);
          xi = mfma64(-wi, yr[cj][v], xi);
        }
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        const long xo = (long)(c0 + 16 * ci + HPX_ACC_ROW(g, v)) * TP + t0 + 16 * tt + li;
        Xre[xo] = xr[v];
        Xim[xo] = xi[v];
      }
    }
  }
}

template <int NT>
__device__ __forceinline__ void bs_pair(const double* __restrict__ Lre, const double* __restrict__ Lim,
                                        const double* __restrict__ Wgre_all, const double* __restrict__ Wgim_all,
                                        double* __restrict__ Xre, double* __restrict__ Xim, const int npad,
                                        const int TP, const int t0, const int wave, const int lane) {
  const int nblk = (npad + HPX_NB - 1) / HPX_NB, nsb = (nblk + 3) >> 2;
  d4 a1[2][2], a2[2][2], a3[2][2];
  for (int J = nsb - 1; J >= 0; --J) {
    const int jb = 4 * J + wave;
    const bool have = jb < nblk;
    const int c0 = jb * HPX_NB;
    const int ct = have ? (min(HPX_NB, npad - c0) >> 4) : 0;      // 16-column tiles of this wave's block
    // phase A: rows below the super-block
    const int rbeg = min(npad, 128 * (J + 1));
    if (ct == 2) {
      bs_init<2, NT>(a1, a2, a3, Lre, Lim, npad, c0, t0, lane);
      bs_accumulate<2, NT>(a1, a2, a3, Lre, Lim, Xre, Xim, npad, TP, c0, t0, rbeg, (npad - rbeg) >> 4, lane);
    } else if (ct == 1) {
      bs_init<1, NT>(a1, a2, a3, Lre, Lim, npad, c0, t0, lane);
      bs_accumulate<1, NT>(a1, a2, a3, Lre, Lim, Xre, Xim, npad, TP, c0, t0, rbeg, (npad - rbeg) >> 4, lane);
    }
    // phase B: the super-block's own block columns, last to first
    for (int w = 3; w >= 0; --w) {
      const int jw = 4 * J + w;
      if (jw >= nblk) continue;                                    // uniform over the workgroup
      if (wave == w) {
        if (ct == 2) bs_finish<2, NT>(a1, a2, a3, Wgre_all + (long)jw * 1024, Wgim_all + (long)jw * 1024, Xre, Xim, TP, c0, t0, lane);
        else bs_finish<1, NT>(a1, a2, a3, Wgre_all + (long)jw * 1024, Wgim_all + (long)jw * 1024, Xre, Xim, TP, c0, t0, lane);
      }
      __syncthreads();                                             // X rows of block jw are in memory
      if (wave < w) {                                              // (the waves to the left always hold 32-wide blocks)
        const int rw = jw * HPX_NB, nchw = min(HPX_NB, npad - rw) >> 4;
        bs_accumulate<2, NT>(a1, a2, a3, Lre, Lim, Xre, Xim, npad, TP, c0, t0, rw, nchw, lane);
      }
    }
  }
}

__global__ __launch_bounds__(256, 2) void k_backsolve(const double* __restrict__ L_all,
                                                      const double* __restrict__ Wre_all,
                                                      const double* __restrict__ Wim_all,
                                                      double* __restrict__ Xre_all,
                                                      double* __restrict__ Xim_all,
                                                      const int npad, const int TP, const int ld) {
  const int b = blockIdx.x;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const double* Lre = L_all + (long)b * npad * ld * 2;
  const double* Lim = Lre + 16;
  const int nblk = (npad + HPX_NB - 1) / HPX_NB;
  const double* Wgre = Wre_all + (long)b * nblk * 1024;
  const double* Wgim = Wim_all + (long)b * nblk * 1024;
  double* Xre = Xre_all + (long)b * npad * TP;
  double* Xim = Xim_all + (long)b * npad * TP;
  const int TT = TP >> 4;
  for (int tp = 0; tp < TT; tp += 2) {                 // pairs of t-tiles (T <= 32: one pass)
    if (tp + 1 < TT) bs_pair<2>(Lre, Lim, Wgre, Wgim, Xre, Xim, npad, TP, tp << 4, wave, lane);
    else bs_pair<1>(Lre, Lim, Wgre, Wgim, Xre, Xim, npad, TP, tp << 4, wave, lane);
  }
}

// ---------------------------------------------------------------------------
// stand-alone helpers: interleaved row-major <-> planar column-major
__global__ void k_pack_herm(const double* __restrict__ a, const double* __restrict__ rhs,
                            double* __restrict__ L, const int n,
                            const int nrhs, const int npad, const int ld) {
  // a (nb,n,n) c128 row-major; rhs (nb,n,nrhs) c128 row-major or NULL
  const int b = blockIdx.y;
  const long tot = (long)npad * ld;
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < tot;
       e += (long)gridDim.x * blockDim.x) {
    const int c = (int)(e / ld), r = (int)(e % ld);
    double vr = 0.0, vi = 0.0;
    if (r < npad) {
      if (r >= c) {
        if (r < n && c < n) {
          const long o = ((long)b * n * n + (long)r * n + c) * 2;
          vr = a[o];
          vi = a[o + 1];
        } else if (r == c) vr = 1.0;
      }
    } else if (rhs && c < n && (r - npad) < nrhs) {   // row npad+t, col c = conj(rhs[c][t])
      const long o = ((long)b * n * nrhs + (long)c * nrhs + (r - npad)) * 2;
      vr = rhs[o];
      vi = -rhs[o + 1];
    }
    const long o = (long)b * tot * 2 + HPX_LIDX(r, c, npad);
    L[o] = vr;
    L[o + 16] = vi;
  }
}

__global__ void k_unpack_lower(const double* __restrict__ L,
                               double* __restrict__ out, const int n, const int npad,
                               const int ld) {
  const int b = blockIdx.y;
  const long tot = (long)n * n;
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < tot;
       e += (long)gridDim.x * blockDim.x) {
    const int r = (int)(e / n), c = (int)(e % n);
    double vr = 0.0, vi = 0.0;
    if (r >= c) {
      const long o = (long)b * npad * ld * 2 + HPX_LIDX(r, c, npad);
      vr = L[o];
      vi = L[o + 16];
    }
    out[((long)b * tot + e) * 2] = vr;
    out[((long)b * tot + e) * 2 + 1] = vi;
  }
}

__global__ void k_unpack_x(const double* __restrict__ Xre, const double* __restrict__ Xim,
                           double* __restrict__ out, const int n, const int nrhs, const int npad,
                           const int TP) {
  const int b = blockIdx.y;
  const long tot = (long)n * nrhs;
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < tot;
       e += (long)gridDim.x * blockDim.x) {
    const int r = (int)(e / nrhs), t = (int)(e % nrhs);
    const long o = (long)b * npad * TP + (long)r * TP + t;
    out[((long)b * tot + e) * 2] = Xre[o];
    out[((long)b * tot + e) * 2 + 1] = Xim[o];
  }
}

}  // namespace

// Waves per workgroup: 4 (two workgroups per CU) or 8 (one per CU, the baseline's tile passes spread over
// twice the waves; HPX_FACTOR_WAVES in the environment, read once -- see docs/HISTORY.md section 6 for the
// measurements behind the default).
template <bool GEN, bool GLDS>
static int launch_factor_t(int nbl, size_t lds, int npad, int ld, double* L, double* Wre, double* Wim,
                           int32_t* info, int iter_tag, const hpx_gen_batch& gen, hipStream_t st) {
  static hpx_lds_limit limit;      // per instantiation
  HPX_TRY(limit.ensure(reinterpret_cast<const void*>(&k_factor<GEN, GLDS>), lds));
  hipLaunchKernelGGL((k_factor<GEN, GLDS>), dim3(nbl), dim3(256), lds, st, L, Wre, Wim, info, npad, ld, iter_tag, gen);
  HPX_HIP(hipGetLastError());
  return HPX_OK;
}

// Which form runs: the wide one (hpx_factor_wide.hip: 128-column super-blocks, LDS-staged panels) from HPX_WIDE_MIN
// columns on, this file's 32-wide one below.  Measured on one MI355X, 1024 systems (docs/HISTORY.md section 11): order 1040
// 24.7 against 27.9 ms, order 528 4.06 against 4.19 ms, order 272 1.05 against 0.99 ms, order 144 0.92 against 0.73 ms
// -- the wide form pays from four super-blocks on.
#ifndef HPX_WIDE_MIN
#define HPX_WIDE_MIN 400
#endif
int hpx_launch_factor(int nbl, int npad, int ld, double* L, double* Wre, double* Wim, double* Vt,
                      int32_t* info, int iter_tag, const hpx_gen_batch* gen, hipStream_t st, int allow_split) {
  if (Vt && allow_split) {      // a batch too small for one workgroup per CU: several workgroups per system (hpx_factor_split.hip)
    int took = 0;
    HPX_TRY(hpx_launch_factor_split(nbl, npad, ld, L, Wre, Wim, Vt, info, iter_tag, gen, st, &took));
    if (took) return HPX_OK;    // (else: not applicable, or no room beside the split launches in flight on other streams)
  }
  if (Vt && npad >= HPX_WIDE_MIN)
    return hpx_launch_factor_wide(nbl, npad, ld, L, Wre, Wim, Vt, info, iter_tag, gen, st);
  const size_t base = sizeof(FactorShared);
  if (!gen) {
    hpx_gen_batch none = {};
    return launch_factor_t<false, false>(nbl, base, npad, ld, L, Wre, Wim, info, iter_tag, none, st);
  }
  // stage a / circ in LDS while two workgroups still fit on a CU (160 KB)
  const size_t staged = base + (size_t)3 * gen->N * sizeof(double);
  if (staged <= (size_t)80 * 1024)
    return launch_factor_t<true, true>(nbl, staged, npad, ld, L, Wre, Wim, info, iter_tag, *gen, st);
  return launch_factor_t<true, false>(nbl, base, npad, ld, L, Wre, Wim, info, iter_tag, *gen, st);
}

// Which form a batch of nb systems of order n with nrhs right-hand sides takes on the current device (an idle one):
// 0 the 32-wide kernel, 1 the wide kernel, 2 the split kernel (*parts = workgroups per system)
extern "C" int hpx_factor_form(int nb, int n, int nrhs, int* parts) {
  const int npad = ceil16(n), ld = npad + (nrhs > 0 ? ceil16(nrhs) : 0);
  const int pr = hpx_factor_split_parts(nb, npad, ld);
  if (parts) *parts = pr;
  if (pr) return 2;
  return npad >= HPX_WIDE_MIN ? 1 : 0;
}

int hpx_launch_backsolve(int nbl, int npad, int TP, int ld, const double* L, const double* Wre,
                         const double* Wim, double* Xre, double* Xim, hipStream_t st) {
  if (hpx_backsolve_reg_ok(npad, TP)) return hpx_launch_backsolve_reg(nbl, npad, TP, ld, L, Wre, Wim, Xre, Xim, st);
  if (hpx_backsolve_x_ok(npad, TP)) return hpx_launch_backsolve_x(nbl, npad, ld, L, Wre, Wim, Xre, Xim, st);
  hipLaunchKernelGGL(k_backsolve, dim3(nbl), dim3(256), 0, st, L, Wre, Wim, Xre, Xim, npad, TP, ld);
  HPX_HIP(hipGetLastError());
  return HPX_OK;
}

// ---- stand-alone C-ABI entry points (tests, other callers) -----------------
namespace {
struct Scratch {
  double *L = nullptr, *Wre = nullptr, *Wim = nullptr, *Vt = nullptr, *Xre = nullptr, *Xim = nullptr;
  ~Scratch() {
    (void)hipFree(L); (void)hipFree(Wre); (void)hipFree(Wim); (void)hipFree(Vt); (void)hipFree(Xre); (void)hipFree(Xim);
  }
};
}  // namespace

static int potr_common(int nb, int n, int nrhs, const double* a, const double* rhs, double* l_out,
                       double* x_out, int32_t* info, hipStream_t st) {
  HPX_REQUIRE(nb > 0 && n > 0 && a, "hpx_zpotr*: bad dimensions or null matrix");
  const int npad = ceil16(n), TP = nrhs > 0 ? ceil16(nrhs) : 0, ld = npad + TP;
  const int nblk = (npad + HPX_NB - 1) / HPX_NB;
  Scratch s;
  const size_t lbytes = (size_t)nb * npad * ld * sizeof(double);
  HPX_HIP(hipMalloc(&s.L, 2 * lbytes));
  HPX_HIP(hipMalloc(&s.Wre, (size_t)nb * nblk * 1024 * sizeof(double)));
  HPX_HIP(hipMalloc(&s.Wim, (size_t)nb * nblk * 1024 * sizeof(double)));
  HPX_HIP(hipMalloc(&s.Vt, (size_t)nb * HPX_VT_STRIDE(npad) * sizeof(double)));
  HPX_HIP(hipMemsetAsync(s.Vt, 0, (size_t)nb * HPX_VT_STRIDE(npad) * sizeof(double), st));
  if (info) HPX_HIP(hipMemsetAsync(info, 0, (size_t)nb * sizeof(int32_t), st));
  hipLaunchKernelGGL(k_pack_herm, dim3(64, nb), dim3(256), 0, st, a, rhs, s.L, n, nrhs, npad, ld);
  HPX_HIP(hipGetLastError());
  HPX_TRY(hpx_launch_factor(nb, npad, ld, s.L, s.Wre, s.Wim, s.Vt, info, 1, nullptr, st));
  if (l_out) {
    hipLaunchKernelGGL(k_unpack_lower, dim3(64, nb), dim3(256), 0, st, s.L, l_out, n, npad, ld);
    HPX_HIP(hipGetLastError());
  }
  if (x_out) {
    const size_t xbytes = (size_t)nb * npad * TP * sizeof(double);
    HPX_HIP(hipMalloc(&s.Xre, xbytes));
    HPX_HIP(hipMalloc(&s.Xim, xbytes));
    HPX_HIP(hipMemsetAsync(s.Xre, 0, xbytes, st));
    HPX_HIP(hipMemsetAsync(s.Xim, 0, xbytes, st));
    HPX_TRY(hpx_launch_backsolve(nb, npad, TP, ld, s.L, s.Wre, s.Wim, s.Xre, s.Xim, st));
    hipLaunchKernelGGL(k_unpack_x, dim3(64, nb), dim3(256), 0, st, s.Xre, s.Xim, x_out, n, nrhs,
                       npad, TP);
    HPX_HIP(hipGetLastError());
  }
  HPX_HIP(hipStreamSynchronize(st));
  return HPX_OK;
}


extern "C" int hpx_zpotrf_batched(int nb, int n, const double* a, double* l_out, int32_t* info,
                                  void* stream) {
  HPX_REQUIRE(l_out, "hpx_zpotrf_batched: null output");
  return potr_common(nb, n, 0, a, nullptr, l_out, nullptr, info, (hipStream_t)stream);
}

extern "C" int hpx_zpotrs_batched(int nb, int n, int nrhs, const double* a, const double* b,
                                  double* x_out, int32_t* info, void* stream) {
  HPX_REQUIRE(nrhs > 0 && b && x_out, "hpx_zpotrs_batched: bad right-hand side");
  return potr_common(nb, n, nrhs, a, b, nullptr, x_out, info, (hipStream_t)stream);
}
