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

#define HPX_NB 32          // block-column width of the factorisation
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
  int64_t bytes;
  // factor / solution
  double *Lre, *Lim;       // [nbl][npad][ld] column-major planar
  double *Wre, *Wim;       // [nbl][nblk][32][32] inverse diagonal blocks
  double *Xre, *Xim;       // [nbl][npad][TP] solution [y' ; f]
  int32_t *info;           // [nbl]
  // chain state
  double *a, *ps_cur;      // [nbl][N]
  double *beta, *betam;    // [nbl][N]
  double *lnp1;            // [nbl]
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
  double *SHre, *SHim;     // [nbl][N][N] (allocated on demand)
  std::vector<hipEvent_t> events;   // profiling: (HPX_NSTAGE+1) per iteration
  int ev_used;
  float stage_ms[HPX_NSTAGE];
  double lgam_T;           // lgamma(T)
  std::vector<void*> allocs;
};

// ---- launchers (each returns HPX_OK / HPX_EHIP) -----------------------------
int hpx_launch_factor(int nbl, int npad, int ld, double* Lre, double* Lim, double* Wre,
                      double* Wim, int32_t* info, int iter_tag, hipStream_t st);
int hpx_launch_backsolve(int nbl, int npad, int TP, int ld, const double* Lre,
                         const double* Lim, const double* Wre, const double* Wim,
                         double* Xre, double* Xim, hipStream_t st);
// out[b][x][c] = scale * sum_k W[x][k] in[b][k][c] (W = fop or conj(fop)), optional
// row scaling of the input by rs[b][k]; ncol multiple of 16; matrices [NP][NP].
int hpx_launch_dft(int nbl, int NP, int ncol, const double* Wre, const double* Wim,
                   int conjW, const double* inre, const double* inim, long in_bstride,
                   int in_ld, const double* rs, int rs_n, double* outre, double* outim,
                   long out_bstride, int out_ld, double scale, hipStream_t st);
int hpx_fop_to_planar(const double* fop, double* re, double* im, int N, int NP, hipStream_t st);
