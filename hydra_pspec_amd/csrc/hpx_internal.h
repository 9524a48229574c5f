// Internal declarations shared by the libhpx translation units (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <vector>
#include "../../include/hpx.h"

// ---- error plumbing -------------------------------------------------------
void hpx_set_error(const char* fmt, ...);
#define HPX_HIP(call)                                                              \
  do {                                                                             \
    hipError_t e_ = (call);                                                        \
    if (e_ != hipSuccess) {                                                        \
      hpx_set_error("%s:%d %s -> %s", __FILE__, __LINE__, #call, hipGetErrorString(e_)); \
      return HPX_EHIP;                                                             \
    }                                                                              \
  } while (0)
#define HPX_REQUIRE(cond, msg)                                                     \
  do {                                                                             \
    if (!(cond)) { hpx_set_error("%s:%d %s", __FILE__, __LINE__, msg); return HPX_EINVAL; } \
  } while (0)
#define HPX_TRY(call)                                                              \
  do { int rc_ = (call); if (rc_ != HPX_OK) return rc_; } while (0)

static inline int ceil16(int v) { return (v + 15) & ~15; }

// ---- f64 MFMA -------------------------------------------------------------
// v_mfma_f64_16x16x4_f64: D(16x16) += A(16x4) * B(4x16), one f64 of A and of B
// per lane:  lane l holds A[l&15][l>>4] and B[l>>4][l&15];
// D register v of lane l holds D[HPX_ACC_ROW(l>>4, v)][l&15]
// (cdna_hip_programming.md section 3; checked on hardware by tests/test_gpu_mfma.py).
typedef double d4 __attribute__((ext_vector_type(4)));
#define HPX_ACC_ROW(g, v) ((g) + 4 * (v))
__device__ __forceinline__ d4 mfma64(double a, double b, d4 c) {
  return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
}

// Factor storage: 16-row panels, each a contiguous [npad columns][re 16 | im 16] strip.
#define HPX_LIDX(r, c, npad) ((((long)((r) >> 4) * (npad) + (c)) << 5) + ((r) & 15))
#define HPX_NB 32          // block-column width of the factorisation
#define HPX_NPART 16       // slots for per-block partial sums of the residual kernel
#define HPX_WLD 34         // LDS leading dimension (doubles) of 32x32 blocks

// ---- plan -------------------------------------------------------------------
struct hpx_plan {
  int nbl, T, N, M;
  int n;        // N + M
  int npad;     // ceil16(n): order of the padded system
  int TP;       // ceil16(T)
  int ld;       // npad + TP: rows of the augmented factor (RHS rows appended)
  int NP;       // ceil16(N): DFT padding
  int MP;       // ceil16(M)
  int ncolR;    // columns of the invariant block R: TP (Q) + MP (G) + 16 (circ)
  int nblk;     // number of block columns
  int ngrid, nxrows, niter_tab;
  int fg_shared, prior_shared, has_omega, any_flags, have_static, profiling;
  int have_edge; // E / P2T hold the edge tiles of the current static inputs (the factor reads them directly)
  int have_ps;  // the chain state (ps_cur, ia) holds bandpowers: a run may continue without ps0
  // dense (non-diagonal, Hermitian) inverse noise covariance (hpx_plan_set_static_dense): C = U^H Ni U
  // is a general Hermitian matrix instead of a circulant
  int dense_noise;         // 1: unflagged; 2: with flags (Woodbury correction of the unflagged-noise system)
  // dense noise with flags: flagged channels per baseline, the f x (f + T) correction systems, the masked residual
  int wb_fmax;
  int32_t *wb_flist, *wb_fcount;     // [nbl][wb_fmax], [nbl]
  double *wb_W;                      // [nbl][wb_fmax][wb_fmax + T] interleaved complex
  double *RDre, *RDim;               // [nbl][NP][TP] residual w (d - model)
  double *NIre, *NIim;     // [nbl][NP][NP] Ninv, planar row-major (zero padded)
  double *CDre, *CDim;     // [nbl][NP][NP] C = U^H Ninv U
  // time-dependent flags / noise (hpx_plan_set_static_pertime): every time sample has its own system.
  // `child` is a plan over the nbl*T units (one right-hand side each) that holds the per-unit
  // operators and factors; this (parent) plan keeps the chain state and everything after the solve.
  int per_time;            // 1: diagonal noise per time; 2: full noise matrix per time (dense-noise units)
  int omega_mod;           // > 0 (child plans): unit u takes the noise draws of time u % omega_mod
  hpx_plan* child;
  uint8_t* flags_t;        // [nbl][T][N]
  double* ninv_t;          // [nbl][T][N]
  double *PTre, *PTim;     // [T][NP][16]: column 0 = U^H omega_a of time t (the child's P2, per unit u % T)
  double *PTTre, *PTTim;   // the same by row tile, [T][NP][16] with row 0 = that time (the child's P2T)
  // iteration-invariant part of the augmented matrix's rows >= rmin (foreground rows, padding, right-hand
  // sides; signal rows rmin..N-1 when N % 32 != 0) for the columns c < rmin, in the factor's tile layout:
  // the factor reads them directly (hpx_edge_init) instead of a per-iteration copy into its buffer
  double *E;               // [nbl][(ld - rmin)/16 tiles][rmin columns][re16|im16]; RHS rows hold Q (not conjugated)
  double *P2Tre, *P2Tim;   // [TP/16 tiles][NP columns][16 rows]: P2 by row tile (shared by the baselines)
  int solver;   // HPX_SOLVER_DENSE / HPX_SOLVER_FLAT / HPX_SOLVER_LOWRANK (hpx_plan_set_solver)
  int allow_split;  // HPX_OPT_FACTOR_SPLIT of this plan: small batches may take the split factor (default 1)
  int split_retry;  // HPX_OPT_SPLIT_RETRY: a run whose split factor timed out is repeated once without the form (default 1)
  int split_fallbacks;   // how often that happened
  double* ps_start; // [nbl][N] the bandpowers a run continued from (kept for that repeat; allocated on first use)
  // HPX_SOLVER_LOWRANK: flagged channels per baseline, the small Schur system and its solution
  int lr_fmax, lr_npad;
  int32_t *lr_flist, *lr_fcount;
  double *lr_Vt;
  double *lr_c, *lr_L, *lr_Wre, *lr_Wim, *lr_Yre, *lr_Yim, *lr_Bre, *lr_Bim, *lr_Tre, *lr_Tim;
  // FFT form of the low-rank solver (hpx_lowrank.hip): transform input / output, foreground
  // blocks, channel -> flagged index
  int lr_fft, lr_cp;
  double *lr_Ire, *lr_Iim, *lr_Ore, *lr_Oim, *lr_Sre, *lr_Sim;
  int32_t* lr_finv;
  int64_t bytes;
  // factor / solution
  double *L;               // [nbl][ld/16 panels][npad][re16|im16]  (HPX_LIDX)
  double *Wre, *Wim;       // [nbl][nblk][32][32] inverse diagonal blocks
  double *Vt;              // [nbl][npad/16 tiles][16][re16|im16] inverse diagonal 16 x 16 tiles (wide factor)
  double *Xre, *Xim;       // [nbl][npad][TP] solution [y' ; f]
  int32_t *info;           // [nbl]
  // chain state
  double *ia, *ps_cur;     // [nbl][N]: sqrt(N / ps) = 1/a, and the bandpowers themselves
  double *beta, *betam;    // [nbl][N]
  double *bpart, *lnpart;  // [nbl][HPX_NPART][N], [nbl][HPX_NPART]: partial sums of the residual kernel
  double *lnp1;            // [nbl]
  double *lnhist;          // sliced k_draw (small batches): [niter_tab][nbl][ceil(N / 16) + 1] group sums per iteration
  int draw_slices;         // workgroups per baseline of k_draw (hpx_plan_set_rng)
  // invariants
  double *Rre, *Rim;       // [nbl][NP][ncolR]
  double *Cre, *Cim;       // [nbl][N] generator of the circulant C = U^H Ni U
  double *P2re, *P2im;     // [NP][TP] (shared)
  double *Hre, *Him;       // [nbl][M][M]
  double *P4re, *P4im;     // [nbl][M][TP]
  double *Fopre, *Fopim;   // [NP][NP] zero padded
  double *Dre, *Dim;       // [nbl][NP][TP] masked data, transposed (x major)
  double *Fre, *Fim;       // [nbl|1][N][M]
  double *ninv, *ni;       // [nbl][N]
  uint8_t *flags;          // [nbl][N]
  int32_t *pmap;           // [nbl|1][N] row of xgrid or -1
  double *xgrid;           // [nxrows][ngrid]
  double *uni, *igy;       // [niter_tab][N]
  // per-iteration work
  double *Sre, *Sim;       // [nbl][NP][TP] signal realisation s[x][t]
  double *Gre, *Gim;       // [nbl][NP][TP] scratch (DFT input / masked DFT output)
  double *Zre, *Zim;       // [nbl][NP][ncolR] scratch for the invariant transform
  // general (non-Fourier S_initial) first step
  double *SHre, *SHim, *CMre, *CMim, *Y1re, *Y1im, *XTre, *XTim, *RSre, *RSim;   // general step (on demand)
  std::vector<hipEvent_t> events;   // profiling: (HPX_NSTAGE+1) per iteration
  int ev_used;
  float stage_ms[HPX_NSTAGE];
  double lgam_T;           // lgamma(T)
  std::vector<std::pair<void*, size_t>> allocs;   // (device pointer, bytes)
};

// ---- augmented-system generator ------------------------------------------------
// The system solved is K' [y'; f] = r' of DESIGN.md section 2 scaled symmetrically by
// A^-1 = diag(1/a, 1), a = sqrt(ps/N):   M [z; f] = A^-1 r',  z = a . y',
//   M = A^-1 K' A^-1 = [[C + diag(N/ps), G], [G^H, H]].
// Only the diagonal of M depends on the bandpowers; Cholesky is insensitive to the symmetric
// scaling, and s = U z, beta_k = N sum_t |z_kt|^2 need no further scaling.
// Entry (r, c), r >= c, of the augmented matrix of one baseline (ia = 1/a):
//   r, c < N        : circ[r-c] + delta_rc ia_c^2
//   N <= r < N+M    : conj(G[c][m])  /  H[r-N][c-N]
//   N+M <= r < npad : identity padding
//   r >= npad       : conj of right-hand side t = r - npad     (Q[c][t] + ia_c P2[c][t] ; P4[c-N][t])
// Pointers are per baseline (already offset), except P2 (shared).
struct hpx_gen {
  const double *ia, *cre, *cim, *rre, *rim, *p2re, *p2im, *hre, *him, *p4re, *p4im;
  const double *cdre, *cdim;   // dense C[r][c] (leading dimension NP) or NULL: circulant circ[r-c]
  const double *ere, *eim;     // edge tiles (rows >= rmin, columns < rmin) or NULL; p2t: P2 in tile layout
  const double *p2tre, *p2tim;
  int N, M, NP, TP, ncol, has_omega, rmin;
};
// offset of (r, c), r >= rmin, c < rmin in the edge tiles (imaginary part 16 doubles further)
#define HPX_EIDX(r, c, rmin) ((((long)(((r) - (rmin)) >> 4) * (rmin) + (c)) << 5) + ((r) & 15))
// Entry (r, c) of a row tile r >= rmin in a block column c < rmin: from the edge tiles when the plan has
// them (GEN), adding the bandpower-dependent P2 / a term to the right-hand-side rows -- the arithmetic
// of hpx_gen_entry, operation for operation -- otherwise from the factor buffer (assembled by the caller).
template <bool GEN>
__device__ __forceinline__ void hpx_edge_init(const hpx_gen& G, const double* __restrict__ Lre,
                                              const double* __restrict__ Lim, const int r, const int c,
                                              const int npad, const bool use_e, double& vr, double& vi) {
  if (GEN && use_e) {
    const long o = HPX_EIDX(r, c, G.rmin);
    vr = G.ere[o];
    vi = G.eim[o];
    if (r >= npad) {
      if (G.has_omega) {
        const int t = r - npad;
        const long q = (((long)(t >> 4) * G.NP + c) << 4) + (t & 15);
        const double ic = G.ia[c];
        vr = fma(ic, G.p2tre[q], vr);
        vi = fma(ic, G.p2tim[q], vi);
      }
      vi = -vi;
    }
  } else {
    const long off = HPX_LIDX(r, c, npad);
    vr = Lre[off];
    vi = Lim[off];
  }
}
__device__ __forceinline__ void hpx_gen_entry(const hpx_gen& G, const int r, const int c,
                                              const int npad, double& vr, double& vi) {
  vr = 0.0;
  vi = 0.0;
  const int N = G.N, M = G.M;
  if (r < c) return;
  if (c < N) {
    if (r < N) {
      const double ic = G.ia[c];
      if (G.cdre) {
        // C is Hermitian: entry (r, c) is read as the conjugate of (c, r) -- the callers walk down a column (r fastest),
        // which in the row-major matrix is a stride of NP doubles for (r, c) and contiguous for (c, r)
        vr = G.cdre[(long)c * G.NP + r] + (r == c ? ic * ic : 0.0);
        vi = (r == c) ? 0.0 : -G.cdim[(long)c * G.NP + r];
      } else {
        vr = G.cre[r - c] + (r == c ? ic * ic : 0.0);
        vi = (r == c) ? 0.0 : G.cim[r - c];
      }
    } else if (r < N + M) {
      const long o = (long)c * G.ncol + G.TP + (r - N);
      vr = G.rre[o];
      vi = -G.rim[o];
    } else if (r >= npad) {
      const int t = r - npad;
      const long o = (long)c * G.ncol + t;
      vr = G.rre[o];
      vi = G.rim[o];
      if (G.has_omega) {
        const double ic = G.ia[c];
        vr = fma(ic, G.p2re[(long)c * G.TP + t], vr);
        vi = fma(ic, G.p2im[(long)c * G.TP + t], vi);
      }
      vi = -vi;
    }
  } else if (c < N + M) {
    const int mc = c - N;
    if (r < N + M) {
      vr = G.hre[(long)(r - N) * M + mc];
      vi = (r == c) ? 0.0 : G.him[(long)(r - N) * M + mc];
    } else if (r >= npad) {
      const int t = r - npad;
      vr = G.p4re[(long)mc * G.TP + t];
      vi = -G.p4im[(long)mc * G.TP + t];
    }
  } else if (r == c) {
    vr = 1.0;
  }
}
// Signal x signal entry strictly below the diagonal: circ[r-c] (no branches).  Used by
// the factor kernel for all row tiles below `rmin` (= 32*floor(N/32)); rows >= rmin (foreground
// modes, padding, right-hand sides) are written to the factor buffer by k_assemble beforehand.
__device__ __forceinline__ void hpx_gen_signal(const hpx_gen& G, const int r, const int c,
                                               double& vr, double& vi) {
  vr = G.cre[r - c];
  vi = G.cim[r - c];
}

// batch-level description: per-baseline strides are implied by the plan dimensions
struct hpx_gen_batch {
  const double *ia, *cre, *cim, *rre, *rim, *p2re, *p2im, *hre, *him, *p4re, *p4im;
  const double *cdre, *cdim;   // dense C, [nbl][NP][NP], or NULL
  const double *ere;           // edge tiles [nbl][e_bstride] or NULL
  const double *p2tre, *p2tim; // P2 in tile layout; per-time units: block (b % p2_mod) * p2t_stride
  long e_bstride, p2t_stride;
  int N, M, NP, TP, ncol, has_omega, rmin;
  int ia_div;                  // baseline b reads 1/a of chain b / ia_div (per-time units share their baseline's)
  int p2_mod;                  // ... and the omega_a block (b % p2_mod) * p2_stride (0 / 1: one shared block)
  long p2_stride;
};
__device__ __forceinline__ hpx_gen hpx_gen_for(const hpx_gen_batch& B, const int b) {
  hpx_gen G;
  G.ia = B.ia + (long)(B.ia_div > 1 ? b / B.ia_div : b) * B.N;
  G.cre = B.cre + (long)b * B.N;
  G.cim = B.cim + (long)b * B.N;
  G.rre = B.rre + (long)b * B.NP * B.ncol;
  G.rim = B.rim + (long)b * B.NP * B.ncol;
  G.p2re = B.p2re + (B.p2_mod > 1 ? (long)(b % B.p2_mod) * B.p2_stride : 0);
  G.p2im = B.p2im + (B.p2_mod > 1 ? (long)(b % B.p2_mod) * B.p2_stride : 0);
  G.hre = B.hre + (long)b * B.M * B.M;
  G.him = B.him + (long)b * B.M * B.M;
  G.p4re = B.p4re + (long)b * B.M * B.TP;
  G.p4im = B.p4im + (long)b * B.M * B.TP;
  G.ere = B.ere ? B.ere + (long)b * B.e_bstride : nullptr;
  G.eim = B.ere ? G.ere + 16 : nullptr;
  G.p2tre = B.p2tre ? B.p2tre + (B.p2_mod > 1 ? (long)(b % B.p2_mod) * B.p2t_stride : 0) : nullptr;
  G.p2tim = B.p2tim ? B.p2tim + (B.p2_mod > 1 ? (long)(b % B.p2_mod) * B.p2t_stride : 0) : nullptr;
  G.cdre = B.cdre ? B.cdre + (long)b * B.NP * B.NP : nullptr;
  G.cdim = B.cdim ? B.cdim + (long)b * B.NP * B.NP : nullptr;
  G.N = B.N; G.M = B.M; G.NP = B.NP; G.TP = B.TP; G.ncol = B.ncol; G.has_omega = B.has_omega;
  G.rmin = B.rmin;
  return G;
}

// Scratch device buffer of a stand-alone entry point: freed on every return path.
struct hpx_devbuf {
  double* p = nullptr;
  hpx_devbuf() = default;
  hpx_devbuf(const hpx_devbuf&) = delete;
  hpx_devbuf& operator=(const hpx_devbuf&) = delete;
  ~hpx_devbuf() { if (p) (void)hipFree(p); }
  int alloc(size_t doubles) {
    HPX_HIP(hipMalloc(&p, (doubles ? doubles : 1) * sizeof(double)));
    return HPX_OK;
  }
};
struct hpx_event {
  hipEvent_t e = nullptr;
  ~hpx_event() { if (e) (void)hipEventDestroy(e); }
  int create() { HPX_HIP(hipEventCreate(&e)); return HPX_OK; }
};

// Raise a kernel's dynamic-LDS limit when a launch needs more than the current one.  The limit
// is a property of the function ON A DEVICE: remembered per (call site, device), since one
// process may drive several GPUs (hpx_set_device).
struct hpx_lds_limit {
  size_t set[32] = {};
  int ensure(const void* func, size_t bytes) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 32) dev = 0;
    if (bytes <= set[dev]) return HPX_OK;
    HPX_HIP(hipFuncSetAttribute(func, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
    set[dev] = bytes;
    return HPX_OK;
  }
};

// ---- launchers (each returns HPX_OK / HPX_EHIP) -----------------------------
// gen == nullptr: factor the matrix stored in L in place; otherwise the augmented matrix is generated on the
// fly from *gen and L is write-only.
// Vt [nbl][HPX_VT_STRIDE(npad)]: per system the inverses of the diagonal 16 x 16 tiles in tile layout (npad * 32
// doubles; workspace of the wide and split forms) and HPX_VT_SYNC doubles of hand-off flags for the split form --
// zero when allocated, and left zero by every launch.
#define HPX_VT_SYNC 64
#define HPX_VT_STRIDE(npad) ((size_t)(npad) * 32 + HPX_VT_SYNC)
// allow_split: batches too small for one workgroup per CU may take the split form (another order of operations:
// a system's factor then depends, in its last bits, on the size of the batch it is in -- callers that promise
// "alone == inside the batch" for batches of any size pass 0).
int hpx_launch_factor(int nbl, int npad, int ld, double* L, double* Wre, double* Wim, double* Vt,
                      int32_t* info, int iter_tag, const hpx_gen_batch* gen, hipStream_t st, int allow_split = 1);
// the wide (128-column super-block, LDS-staged) form, hpx_factor_wide.hip
int hpx_launch_factor_wide(int nbl, int npad, int ld, double* L, double* Wre, double* Wim, double* Vt,
                           int32_t* info, int iter_tag, const hpx_gen_batch* gen, hipStream_t st);
// Hermitian positive semi-definite eigendecomposition, orders 128 .. (hpx_eigh.hip): planar in, eigenvalues on the
// diagonal of gr, unit eigenvectors as the columns of (vr, vi); n a multiple of 16
int hpx_eigh_padded_order(int n0);
int hpx_eigh_psd_planar(int nb, int n, double* gr, const double* gi, double* vr, double* vi, int* sweeps_out,
                        hipStream_t st);
// the split form (hpx_factor_split.hip): several workgroups per system, for batches too small to fill the chip
int hpx_factor_split_parts(int nbl, int npad, int ld);     // workgroups per system on an idle device, 0 = not applicable
// *took = workgroups per system, or 0 when the form was not taken (not applicable; switched off; no room on the device
// beside the split launches in flight on other streams): the caller then launches a one-workgroup kernel
int hpx_launch_factor_split(int nbl, int npad, int ld, double* L, double* Wre, double* Wim, double* Vt,
                            int32_t* info, int iter_tag, const hpx_gen_batch* gen, hipStream_t st, int* took);
int hpx_split_set_option(int key, int value);
int hpx_eigh_set_option(int key, int value);
// `info` words: the iteration tag (k > 0: first seen at iteration k - 1) with HPX_INFO_TIMEOUT on top when the split
// factor's hand-off between workgroups timed out (the factor is then incomplete -- not a statement about the matrix)
#define HPX_INFO_TIMEOUT 0x40000000
// structured solve for flat noise with flags (hpx_lowrank.hip): writes X = [z; f]
int hpx_launch_solve_lowrank(hpx_plan* p, int iter_tag, hipStream_t st);
size_t hpx_lowrank_lds_bytes(const hpx_plan* p);
int hpx_launch_flat_blocks(hpx_plan* p, const double* cval, double* ore, double* oim, double* xin_re,
                           double* xin_im, int cp, hipStream_t st);
int hpx_lowrank_prepare(hpx_plan* p, hipStream_t st);
// structured solve for flat noise without flags (hpx_flat.hip): writes X = [z; f]
int hpx_launch_solve_flat(hpx_plan* p, int iter_tag, hipStream_t st);
size_t hpx_flat_lds_bytes(const hpx_plan* p);
int hpx_launch_backsolve(int nbl, int npad, int TP, int ld, const double* L, const double* Wre,
                         const double* Wim, double* Xre, double* Xim, hipStream_t st);
// the register-resident form (hpx_backsolve.hip): small orders, up to 32 right-hand sides
int hpx_backsolve_reg_ok(int npad, int TP);
// hpx_backsolve_lds.hip: X through LDS, L three chunks ahead (TP = 32, orders beyond the register form)
int hpx_backsolve_x_ok(int npad, int TP);
int hpx_launch_backsolve_x(int nbl, int npad, int ld, const double* L, const double* Wre, const double* Wim, double* Xre,
                           double* Xim, hipStream_t st);
int hpx_launch_backsolve_reg(int nbl, int npad, int TP, int ld, const double* L, const double* Wre, const double* Wim,
                             double* Xre, double* Xim, hipStream_t st);
// out[b][x][c] = scale * sum_k W[x][k] in[b][k][c] (W = fop or conj(fop)), optional
// row scaling of the input by rs[b][k]; ncol multiple of 16; matrices [NP][NP].
extern int hpx_dft_use_fft;
int hpx_launch_dft(int nbl, int NP, int ncol, const double* Wre, const double* Wim,
                   int conjW, const double* inre, const double* inim, long in_bstride,
                   int in_ld, const double* rs, int rs_n, double* outre, double* outim,
                   long out_bstride, int out_ld, double scale, hipStream_t st, int fft_ok = 1,
                   long W_bstride = 0);
extern int hpx_dft_use_fft;
int hpx_fop_to_planar(const double* fop, double* re, double* im, int N, int NP, hipStream_t st);
