// Plan management, iteration-invariant operators, and the per-iteration
// kernels around the factor/solve: assembly of the augmented system, residual / chi^2 /
// log-posterior reductions, and the inverse-gamma bandpower draw.
#include <math.h>
#include <stdarg.h>
#include <string.h>
#include "hpx_internal.h"
#include <string>
#include <stdlib.h>
#include "hpx_fft.h"

// ---------------------------------------------------------------------------
static thread_local char g_err[512] = "";
void hpx_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
extern "C" const char* hpx_last_error(void) { return g_err; }
extern "C" int hpx_version(void) { return HPX_VERSION; }
extern "C" int hpx_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}
extern "C" int hpx_set_device(int dev) {
  HPX_HIP(hipSetDevice(dev));
  return HPX_OK;
}

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

// ---- setup ------------------------------------------------------------------
// Z[b][j][col]: col<TP: Ni (w d)_t + Ni^1/2 omega_b,t ; TP..: Ni F[:,m] ; TP+MP: Ni
// D[b][j][t] = w_j d[t][j]
__global__ void k_prep(const double* __restrict__ vis, const uint8_t* __restrict__ flags,
                       const double* __restrict__ ninv, const double* __restrict__ fg,
                       const int fg_shared, const double* __restrict__ omega,
                       double* __restrict__ Zre, double* __restrict__ Zim,
                       double* __restrict__ Dre, double* __restrict__ Dim,
                       double* __restrict__ ni_out, const int T, const int N, const int M,
                       const int NP, const int TP, const int MP, const int ncol, const int omega_mod,
                       const int mask_data_only) {
  // mask_data_only: the flags mask the data columns only (dense noise with flags: the matrix blocks are
  // those of the unflagged noise, the mask comes in through the Woodbury correction)
  const int b = blockIdx.y;
  const int tom = omega_mod > 0 ? b % omega_mod : 0;      // per-time units: the draws of "their" time
  const long tot = (long)NP * ncol;
  const double* F = fg + (fg_shared ? 0 : (long)b * N * M * 2);
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < tot;
       e += (long)gridDim.x * blockDim.x) {
    const int j = (int)(e / ncol), col = (int)(e % ncol);
    double zr = 0.0, zi = 0.0;
    if (j < N) {
      const double w = flags[(long)b * N + j] ? 1.0 : 0.0;
      const double ni = ninv[(long)b * N + j] * ((mask_data_only && col >= TP) ? 1.0 : w);
      if (col < TP) {
        const int t = col;
        double dr = 0.0, di = 0.0;
        if (t < T) {
          const long o = (((long)b * T + t) * N + j) * 2;
          dr = vis[o] * w;
          di = vis[o + 1] * w;
          zr = ninv[(long)b * N + j] * dr;
          zi = ninv[(long)b * N + j] * di;
          if (omega) {
            const double nih = sqrt(ni);
            zr += nih * (omega[((long)(t + tom) * 4 + 2) * N + j] / SQRT2);
            zi += nih * (omega[((long)(t + tom) * 4 + 3) * N + j] / SQRT2);
          }
        }
        Dre[((long)b * NP + j) * TP + t] = dr;
        Dim[((long)b * NP + j) * TP + t] = di;
      } else if (col < TP + MP) {
        const int m = col - TP;
        if (m < M) {
          zr = ni * F[((long)j * M + m) * 2];
          zi = ni * F[((long)j * M + m) * 2 + 1];
        }
      } else if (col == TP + MP) {
        zr = ni;
        ni_out[(long)b * N + j] = ni;
      }
    } else if (col < TP) {
      Dre[((long)b * NP + j) * TP + col] = 0.0;
      Dim[((long)b * NP + j) * TP + col] = 0.0;
    }
    Zre[(long)b * tot + e] = zr;
    Zim[(long)b * tot + e] = zi;
  }
}

// shared omega_a block: Z2[j][t] = (omi + i omj)/sqrt2
__global__ void k_prep_omega(const double* __restrict__ omega, double* __restrict__ Zre,
                             double* __restrict__ Zim, const int T, const int N, const int NP,
                             const int TP) {
  const long tot = (long)NP * TP;
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < tot;
       e += (long)gridDim.x * blockDim.x) {
    const int j = (int)(e / TP), t = (int)(e % TP);
    double zr = 0.0, zi = 0.0;
    if (j < N && t < T) {
      zr = omega[((long)t * 4 + 0) * N + j] / SQRT2;
      zi = omega[((long)t * 4 + 1) * N + j] / SQRT2;
    }
    Zre[e] = zr;
    Zim[e] = zi;
  }
}

// circ[m] = R[(m + N/2) mod N][colC] / sqrt(N)
__global__ void k_circ(const double* __restrict__ Rre, const double* __restrict__ Rim,
                       double* __restrict__ Cre, double* __restrict__ Cim, const int N,
                       const int NP, const int ncol, const int colC) {
  const int b = blockIdx.y;
  const double s = 1.0 / sqrt((double)N);
  for (int m = blockIdx.x * blockDim.x + threadIdx.x; m < N; m += gridDim.x * blockDim.x) {
    const int x = (m + N / 2) % N;
    const long o = ((long)b * NP + x) * ncol + colC;
    Cre[(long)b * N + m] = Rre[o] * s;
    Cim[(long)b * N + m] = Rim[o] * s;
  }
}

// H = F^H Ni F (M x M), P4 = F^H (Ni d + Ni^1/2 omega_b) (M x TP); Z holds the operands.
__global__ void k_small(const double* __restrict__ fg, const int fg_shared,
                        const double* __restrict__ Zre, const double* __restrict__ Zim,
                        double* __restrict__ Hre, double* __restrict__ Him,
                        double* __restrict__ P4re, double* __restrict__ P4im, const int N,
                        const int M, const int NP, const int TP, const int MP, const int ncol) {
  const int b = blockIdx.x;
  const double* F = fg + (fg_shared ? 0 : (long)b * N * M * 2);
  const double* zr = Zre + (long)b * NP * ncol;
  const double* zi = Zim + (long)b * NP * ncol;
  const int nh = M * M, np4 = M * TP;
  for (int e = threadIdx.x; e < nh + np4; e += blockDim.x) {
    int m, col;
    if (e < nh) { m = e / M; col = TP + e % M; }
    else { m = (e - nh) / TP; col = (e - nh) % TP; }
    double sr = 0.0, si = 0.0;
    for (int j = 0; j < N; ++j) {
      const double fr = F[((long)j * M + m) * 2], fi = -F[((long)j * M + m) * 2 + 1];   // conj
      const double ar = zr[(long)j * ncol + col], ai = zi[(long)j * ncol + col];
      sr += fr * ar - fi * ai;
      si += fr * ai + fi * ar;
    }
    if (e < nh) {
      Hre[(long)b * nh + e] = sr;
      Him[(long)b * nh + e] = si;
    } else {
      P4re[(long)b * np4 + (e - nh)] = sr;
      P4im[(long)b * np4 + (e - nh)] = si;
    }
  }
}

__global__ void k_fg_planar(const double* __restrict__ fg, double* __restrict__ Fre,
                            double* __restrict__ Fim, const long tot) {
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < tot;
       e += (long)gridDim.x * blockDim.x) {
    Fre[e] = fg[2 * e];
    Fim[e] = fg[2 * e + 1];
  }
}

// ---- dense (Hermitian, non-diagonal) inverse noise covariance -----------------------------------
// (nbl|1, N, N) c128 row-major -> planar [b][NP][NP] zero padded; herm != 0: out[k][x] = conj(in[x][k])
__global__ void k_dense_planar(const double* __restrict__ m, const int shared, double* __restrict__ re,
                               double* __restrict__ im, const int N, const int NP) {
  const int b = blockIdx.y;
  const long tot = (long)NP * NP;
  const double* src = m + (shared ? 0 : (long)b * N * N * 2);
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < tot; e += (long)gridDim.x * blockDim.x) {
    const int k = (int)(e / NP), x = (int)(e % NP);
    double vr = 0.0, vi = 0.0;
    if (k < N && x < N) {
      vr = src[((long)k * N + x) * 2];
      vi = src[((long)k * N + x) * 2 + 1];
    }
    re[(long)b * tot + e] = vr;
    im[(long)b * tot + e] = vi;
  }
}
// planar [b][NP][NP]: out[k][x] = conj(in[x][k])
__global__ void k_conj_transpose(const double* __restrict__ ire, const double* __restrict__ iim,
                                 double* __restrict__ ore, double* __restrict__ oim, const int NP) {
  const int b = blockIdx.y;
  const long tot = (long)NP * NP;
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < tot; e += (long)gridDim.x * blockDim.x) {
    const int k = (int)(e / NP), x = (int)(e % NP);
    ore[(long)b * tot + e] = ire[(long)b * tot + (long)x * NP + k];
    oim[(long)b * tot + e] = -iim[(long)b * tot + (long)x * NP + k];
  }
}
// the real diagonal of the planar matrices -> (nbl, N)
__global__ void k_take_diag(const double* __restrict__ re, double* __restrict__ dg, const int N, const int NP) {
  const int b = blockIdx.y;
  for (int k = blockIdx.x * blockDim.x + threadIdx.x; k < N; k += gridDim.x * blockDim.x)
    dg[(long)b * N + k] = re[(long)b * NP * NP + (long)k * NP + k];
}
// Z[b][j][t] += A[b][j][t] for t < TP (both with leading dimension ld_z / ld_a)
__global__ void k_add_block(double* __restrict__ zre, double* __restrict__ zim, const long z_bs, const int ld_z,
                            const double* __restrict__ are, const double* __restrict__ aim, const long a_bs,
                            const int ld_a, const int NP, const int TP) {
  const int b = blockIdx.y;
  const long tot = (long)NP * TP;
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < tot; e += (long)gridDim.x * blockDim.x) {
    const int j = (int)(e / TP), t = (int)(e % TP);
    zre[(long)b * z_bs + (long)j * ld_z + t] += are[(long)b * a_bs + (long)j * ld_a + t];
    zim[(long)b * z_bs + (long)j * ld_z + t] += aim[(long)b * a_bs + (long)j * ld_a + t];
  }
}
// omega_b block: O[j][t] = (omk + i oml)/sqrt2, replicated per baseline (the dense product is batched)
__global__ void k_prep_omega_b(const double* __restrict__ omega, double* __restrict__ Ore,
                               double* __restrict__ Oim, const int T, const int N, const int NP, const int TP,
                               const int omega_mod) {
  const int b = blockIdx.y;
  const int tom = omega_mod > 0 ? b % omega_mod : 0;      // per-time units: the draws of "their" time
  const long tot = (long)NP * TP;
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < tot; e += (long)gridDim.x * blockDim.x) {
    const int j = (int)(e / TP), t = (int)(e % TP);
    double zr = 0.0, zi = 0.0;
    if (j < N && t < T) {
      zr = omega[((long)(t + tom) * 4 + 2) * N + j] / SQRT2;
      zi = omega[((long)(t + tom) * 4 + 3) * N + j] / SQRT2;
    }
    Ore[(long)b * tot + e] = zr;
    Oim[(long)b * tot + e] = zi;
  }
}
// lnpart[b][0] = sum_{x,t} Re( conj(r[x][t]) v[x][t] )  (r^H Ninv r summed over the times; v = Ninv r)
__global__ __launch_bounds__(256) void k_quadform(const double* __restrict__ rre, const double* __restrict__ rim,
                                                  const long r_bs, const int ld_r, const double* __restrict__ vre,
                                                  const double* __restrict__ vim, const long v_bs, const int ld_v,
                                                  double* __restrict__ lnpart, const int N, const int T) {
  __shared__ double red[4];
  const int b = blockIdx.x;
  double acc = 0.0;
  for (int e = threadIdx.x; e < N * T; e += 256) {
    const int x = e / T, t = e % T;
    const long o1 = (long)b * r_bs + (long)x * ld_r + t, o2 = (long)b * v_bs + (long)x * ld_v + t;
    acc += rre[o1] * vre[o2] + rim[o1] * vim[o2];
  }
  const double tot = block_sum(acc, red);
  if (threadIdx.x == 0) lnpart[(long)b * HPX_NPART] = tot;
}

// ia = 1/a = sqrt(N / ps); bandpowers below HPX_PS_FLOOR (incl. zero) are treated as the floor:
// the channel's signal is then pinned to ~0, which is what a -> 0 means in the unscaled system.
#define HPX_PS_FLOOR 1e-280
__device__ __forceinline__ double inv_a(const double ps, const double dN) {
  // (ps < floor) is false for a NaN, which therefore propagates into the pivots and is reported
  return sqrt(dN / ((ps < HPX_PS_FLOOR) ? HPX_PS_FLOOR : ps));
}
__global__ void k_set_a(const double* __restrict__ ps, double* __restrict__ ia,
                        double* __restrict__ ps_cur, const long tot, const double dN) {
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < tot;
       e += (long)gridDim.x * blockDim.x) {
    const double v = ps[e];
    ps_cur[e] = v;
    ia[e] = inv_a(v, dN);
  }
}

// ---- assembly of the augmented (scaled) system M_aug ----------------------------
// rows >= rlo only (rlo = 0: the whole lower triangle + right-hand sides)
__global__ __launch_bounds__(256) void k_assemble(const hpx_gen_batch B, double* __restrict__ L_all,
                                                  const int npad, const int ld, const int rlo) {
  const int b = blockIdx.y, cb = blockIdx.x;
  const hpx_gen G = hpx_gen_for(B, b);
  double* L = L_all + (long)b * npad * ld * 2;
  const int rbeg = max(cb * 16, rlo), nrow = ld - rbeg;
  for (int e = threadIdx.x; e < 16 * nrow; e += 256) {
    const int c = cb * 16 + e / nrow, r = rbeg + e % nrow;
    double vr, vi;
    hpx_gen_entry(G, r, c, npad, vr, vi);
    const long o = HPX_LIDX(r, c, npad);
    L[o] = vr;
    L[o + 16] = vi;
  }
}

// The rows the factor does not generate itself (r >= rmin: foreground rows, padding, right-hand
// sides), one workgroup per baseline.  The bulk -- rows >= N of the signal columns -- is copied
// from the invariant block R with unit-stride reads along the row index (16 columns x 16 rows per
// round of the block); what is left (signal rows rmin..N-1 when N is not a multiple of 32, and
// the last columns c >= N) goes through the generic entry function.
__global__ __launch_bounds__(256) void k_assemble_edge(const hpx_gen_batch B, double* __restrict__ L_all,
                                                       const int npad, const int ld) {
  const int b = blockIdx.x, tid = threadIdx.x;
  const hpx_gen G = hpx_gen_for(B, b);
  const int N = B.N, M = B.M, TP = B.TP, ncol = B.ncol, rmin = B.rmin;
  double* L = L_all + (long)b * npad * ld * 2;
  const int ci = tid >> 4, ri = tid & 15;
  for (int c0 = 16 * blockIdx.y; c0 < N; c0 += 16 * gridDim.y) {     // column blocks over grid.y
    const int c = c0 + ci;
    if (c >= N) continue;
    const double ic = G.ia[c];
    for (int r = N + ri; r < ld; r += 16) {
      double vr = 0.0, vi = 0.0;
      if (r < N + M) {
        const long o = (long)c * ncol + TP + (r - N);
        vr = G.rre[o];
        vi = -G.rim[o];
      } else if (r >= npad) {
        const int t = r - npad;
        const long o = (long)c * ncol + t;
        vr = G.rre[o];
        vi = G.rim[o];
        if (G.has_omega) {
          vr = fma(ic, G.p2re[(long)c * TP + t], vr);
          vi = fma(ic, G.p2im[(long)c * TP + t], vi);
        }
        vi = -vi;
      }
      const long o = HPX_LIDX(r, c, npad);
      L[o] = vr;
      L[o + 16] = vi;
    }
  }
  if (blockIdx.y != 0) return;
  // signal rows rmin <= r < N of the signal columns (N not a multiple of 32)
  const int nsr = N - rmin;
  for (int e = tid; e < nsr * N; e += 256) {
    const int c = e / nsr, r = rmin + e - c * nsr;
    if (r < c) continue;
    double vr, vi;
    hpx_gen_entry(G, r, c, npad, vr, vi);
    const long o = HPX_LIDX(r, c, npad);
    L[o] = vr;
    L[o + 16] = vi;
  }
  // columns c >= N (foreground x foreground block, padding, their right-hand sides)
  const int ncl = npad - N, nrl = ld - N;
  for (int e = tid; e < ncl * nrl; e += 256) {
    const int c = N + e / nrl, r = N + e % nrl;
    if (r < c) continue;
    double vr, vi;
    hpx_gen_entry(G, r, c, npad, vr, vi);
    const long o = HPX_LIDX(r, c, npad);
    L[o] = vr;
    L[o + 16] = vi;
  }
}

// Edge tiles (hpx_internal.h: E): the iteration-invariant part of rows >= rmin for the columns c < rmin,
// in the factor's tile layout.  Right-hand-side rows hold Q un-conjugated (hpx_edge_init adds P2 / a and
// conjugates, as hpx_gen_entry does); everything else is the entry itself.
__global__ __launch_bounds__(256) void k_build_edge(const hpx_gen_batch B, double* __restrict__ E_all,
                                                    const int npad, const int ld) {
  const int b = blockIdx.y;
  hpx_gen_batch B0 = B;
  B0.has_omega = 0;                                   // the invariant part only
  const hpx_gen G = hpx_gen_for(B0, b);
  const int rmin = B.rmin, nrow = ld - rmin;
  double* E = E_all + (long)b * B.e_bstride;
  const long tot = (long)nrow * rmin;
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < tot; e += (long)gridDim.x * blockDim.x) {
    const int c = (int)(e / nrow), r = rmin + (int)(e % nrow);
    double vr, vi;
    hpx_gen_entry(G, r, c, npad, vr, vi);
    if (r >= npad) vi = -vi;                          // Q itself: the conjugation happens at the use
    const long o = HPX_EIDX(r, c, rmin);
    E[o] = vr;
    E[o + 16] = vi;
  }
}
// P2T[(t >> 4)][c][t & 15] = P2[c][t]
__global__ void k_p2_tiles(const double* __restrict__ p2re, const double* __restrict__ p2im,
                           double* __restrict__ tre, double* __restrict__ tim, const int NP, const int TP) {
  const long tot = (long)NP * TP;
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < tot; e += (long)gridDim.x * blockDim.x) {
    const int c = (int)(e / TP), t = (int)(e % TP);
    const long q = (((long)(t >> 4) * NP + c) << 4) + (t & 15);
    tre[q] = p2re[e];
    tim[q] = p2im[e];
  }
}
// What is left to lay out per iteration once the factor reads the edge tiles itself: the columns
// c >= rmin (foreground x foreground block, identity padding, their right-hand sides; signal columns
// rmin..N-1 when N % 32 != 0), rows r >= c.
__global__ __launch_bounds__(256) void k_assemble_tail(const hpx_gen_batch B, double* __restrict__ L_all,
                                                       const int npad, const int ld) {
  const int b = blockIdx.x, tid = threadIdx.x;
  const hpx_gen G = hpx_gen_for(B, b);
  double* L = L_all + (long)b * npad * ld * 2;
  const int rmin = B.rmin, ncl = npad - rmin, nrl = ld - rmin;
  for (int e = tid; e < ncl * nrl; e += 256) {
    const int c = rmin + e / nrl, r = rmin + e % nrl;
    if (r < c) continue;
    double vr, vi;
    hpx_gen_entry(G, r, c, npad, vr, vi);
    const long o = HPX_LIDX(r, c, npad);
    L[o] = vr;
    L[o + 16] = vi;
  }
}

__global__ void k_kaug_out(const double* __restrict__ L, double* __restrict__ out, const int npad,
                           const int ld) {
  // (nbl, ld, npad) c128 row-major
  const int b = blockIdx.y;
  const long tot = (long)ld * npad;
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < tot;
       e += (long)gridDim.x * blockDim.x) {
    const int r = (int)(e / npad), c = (int)(e % npad);
    const long o = (long)b * npad * ld * 2 + HPX_LIDX(r, c, npad);
    const bool keep = (r >= c);
    out[((long)b * tot + e) * 2] = keep ? L[o] : 0.0;
    out[((long)b * tot + e) * 2 + 1] = keep ? L[o + 16] : 0.0;
  }
}

// ---- residual, chi^2, first part of ln posterior, beta --------------------------
struct ResArgs {
  const double *Xre, *Xim, *Sre, *Sim, *Dre, *Dim, *Fre, *Fim, *ninv;
  const uint8_t* flags;
  double *bpart, *lnpart, *Gre, *Gim;   // partial sums (HPX_NPART slots per baseline); G: masked
                                        // signal (only if any_flags)
  const double *twre, *twim;            // centred Fourier operator (twiddles of the fused kernel)
  double isn;                           // 1 / sqrt(N)
  int logN, tcs;                        // fused kernel: log2 N, log2 of the time columns per block
  double *cr_out, *fg_out, *chisq_out;  // already offset to the slot; may be NULL
  long cr_bstride, fg_bstride, chisq_bstride;
  int N, M, T, NP, TP, npad, fg_shared, any_flags;
  int nbl, npart;                       // fused kernel: batch size, column groups per baseline
  const uint8_t* flags_t;               // k_resid, per-time mode: [nbl][T][N] flags and inverse noise
  const double* ninv_t;                 // variances (NULL: the time-independent ones above)
  double *Rdre, *Rdim;                  // k_resid: where the masked residual w (d - model) goes, or NULL (dense
                                        // noise: the quadratic form r^H Ninv r over the unflagged channels is
                                        // taken afterwards).  May be G itself when there are no flags.
};

__global__ __launch_bounds__(256) void k_resid(const ResArgs A) {
  extern __shared__ double rl[];      // f_re[M][TP], f_im[M][TP], part[N][TP/16]
  __shared__ double red[4];
  // blockIdx.x: one of A.npart slices of the channels (a workgroup per baseline walked N TP / 256 dependent rounds of
  // loads: 1.1 ms at C5; the slices leave their chi^2 term in their own lnpart slot, k_draw adds the slots)
  const int b = blockIdx.y, jp = blockIdx.x, tid = threadIdx.x;
  const int N = A.N, M = A.M, T = A.T, TP = A.TP, TG = TP >> 4;
  const int xper = N / A.npart, x0 = jp * xper;
  double* lfr = rl;
  double* lfi = rl + (long)M * TP;
  double* part = rl + 2L * M * TP;                     // [xper][TG]
  const double* xre = A.Xre + (long)b * A.npad * TP;
  const double* xim = A.Xim + (long)b * A.npad * TP;
  const double* sre = A.Sre + (long)b * A.NP * TP;
  const double* sim = A.Sim + (long)b * A.NP * TP;
  const double* dre = A.Dre + (long)b * A.NP * TP;
  const double* dim_ = A.Dim + (long)b * A.NP * TP;
  const double* fre = A.Fre + (A.fg_shared ? 0 : (long)b * N * M);
  const double* fim = A.Fim + (A.fg_shared ? 0 : (long)b * N * M);
  const double* ninv = A.ninv + (long)b * N;
  const uint8_t* fl = A.flags + (long)b * N;
  for (int e = tid; e < M * TP; e += 256) {          // foreground amplitudes f[m][t]
    lfr[e] = xre[(long)N * TP + e];
    lfi[e] = xim[(long)N * TP + e];
  }
  __syncthreads();
  double acc = 0.0;
  const int tot = (x0 + xper) * TP;                  // multiple of 16; threads of a 16-lane group
  for (int e0 = x0 * TP; e0 < tot; e0 += 256) {      // share x, so the shuffles stay in range
    const int e = e0 + tid;
    const bool in = e < tot;
    const int x = in ? e / TP : x0, t = in ? e % TP : 0;
    const long o = (long)x * TP + t;
    // beta partial: |z_xt|^2 summed over 16 consecutive times
    double v = 0.0;
    if (in && t < T) {          // (columns >= T are padding, or the Woodbury columns of the dense-noise-with-flags mode)
      const double yr = xre[o], yi = xim[o];
      v = yr * yr + yi * yi;
    }
    v += __shfl_xor(v, 8, 16);
    v += __shfl_xor(v, 4, 16);
    v += __shfl_xor(v, 2, 16);
    v += __shfl_xor(v, 1, 16);
    if (in && (t & 15) == 0) part[(x - x0) * TG + (t >> 4)] = v;
    if (!in) continue;
    if (t >= T) {
      if (A.any_flags) { A.Gre[(long)b * A.NP * TP + o] = 0.0; A.Gim[(long)b * A.NP * TP + o] = 0.0; }
      if (A.Rdre) { A.Rdre[(long)b * A.NP * TP + o] = 0.0; A.Rdim[(long)b * A.NP * TP + o] = 0.0; }
      continue;
    }
    const double sr = sre[o], si = sim[o];
    double mr = sr, mi = si;
    for (int m = 0; m < M; ++m) {
      const double fr = fre[(long)x * M + m], fi = fim[(long)x * M + m];
      const double gr = lfr[m * TP + t], gi = lfi[m * TP + t];
      mr += gr * fr - gi * fi;
      mi += gr * fi + gi * fr;
    }
    const double rr = dre[o] - mr, ri = dim_[o] - mi;
    const long ot = ((long)b * T + t) * N + x;
    const double w = (A.flags_t ? A.flags_t[ot] : fl[x]) ? 1.0 : 0.0;
    const double c2 = (rr * rr + ri * ri) * (A.ninv_t ? A.ninv_t[ot] : ninv[x]);
    acc += w * c2;
    if (A.any_flags) {
      A.Gre[(long)b * A.NP * TP + o] = w * sr;
      A.Gim[(long)b * A.NP * TP + o] = w * si;
    }
    if (A.Rdre) {
      A.Rdre[(long)b * A.NP * TP + o] = w * rr;
      A.Rdim[(long)b * A.NP * TP + o] = w * ri;
    }
    if (A.cr_out) {
      double* q = A.cr_out + (long)b * A.cr_bstride + ((long)t * N + x) * 2;
      q[0] = sr;
      q[1] = si;
    }
    if (A.chisq_out) A.chisq_out[(long)b * A.chisq_bstride + (long)t * N + x] = c2;
  }
  if (A.fg_out && jp == 0) {
    for (int e = tid; e < T * M; e += 256) {
      const int t = e / M, m = e % M;
      double* q = A.fg_out + (long)b * A.fg_bstride + (long)e * 2;
      q[0] = lfr[m * TP + t];
      q[1] = lfi[m * TP + t];
    }
  }
  const double total = block_sum(acc, red);          // (barrier inside: part[] is complete)
  // (dense noise: the quadratic form with the full matrix replaces this term afterwards, in slot 0 -- the other
  // slices' slots must then hold nothing)
  if (tid == 0) A.lnpart[(long)b * HPX_NPART + jp] = (A.Rdre && jp > 0) ? 0.0 : total;
  // sum_t |z_kt|^2; k_draw turns it into beta_k = N sum_t |z_kt|^2  ( |F s|^2 with s = U z ): slot 0 for this
  // slice's channels, nothing in the other slots
  for (int k = x0 + tid; k < x0 + xper; k += 256) {
    double sum = 0.0;
    for (int j = 0; j < TG; ++j) sum += part[(k - x0) * TG + j];
    A.bpart[(long)b * HPX_NPART * N + k] = sum;
    for (int j = 1; j < A.npart; ++j) A.bpart[((long)b * HPX_NPART + j) * N + k] = 0.0;
  }
}

#ifndef HPX_FR_NT
#define HPX_FR_NT 0          // bit 0: solution (z) loads non-temporal, bit 1: data loads
#endif
// Back transform s = U z and everything k_resid does, in one kernel (N a power of two): the
// block that holds TC time columns of the signal in LDS after the FFT goes straight on to the
// model, residual, chi^2 and the optional sample write-back for those columns, so s never goes
// to HBM and back.  Blocks of one baseline leave their partial sums (|z|^2 per channel, the
// chi^2 total) in slot blockIdx.x; k_draw adds the slots in a fixed order.
__global__ __launch_bounds__(256) void k_fft_resid(const ResArgs A) {
  extern __shared__ double fl[];
  __shared__ double red[4];
  const int N = A.N, M = A.M, T = A.T, TP = A.TP, tcs = A.tcs, TC = 1 << tcs, logN = A.logN;
  // column groups of one baseline on one XCD (they share cache lines of X, D and the outputs);
  // workgroup ids go round-robin over the 8 XCDs
  const int b = ((int)(blockIdx.x >> 3) / A.npart) * 8 + (int)(blockIdx.x & 7);
  if (b >= A.nbl) return;
  const int cg = (int)(blockIdx.x >> 3) % A.npart;
  const int c0 = cg * TC, tid = threadIdx.x, h = N >> 1;
  double* fre = fl;
  double* fim = fl + ((long)N << tcs);
  double* tw = fim + ((long)N << tcs);              // N doubles
  double* lfr = tw + N;                             // f[m][tc]: M * TC
  double* lfi = lfr + (M << tcs);
  const double* xre = A.Xre + (long)b * A.npad * TP;
  const double* xim = A.Xim + (long)b * A.npad * TP;
  for (int j = tid; j < h; j += 256) {
    tw[j] = A.twre[(long)(h + 1) * N + h + j];
    tw[h + j] = A.twim[(long)(h + 1) * N + h + j];
  }
  for (int e = tid; e < (M << tcs); e += 256) {
    const int m = e >> tcs, tc = e & (TC - 1);
    lfr[e] = xre[(long)(N + m) * TP + c0 + tc];
    lfi[e] = xim[(long)(N + m) * TP + c0 + tc];
  }
  double* bp = A.bpart + ((long)b * HPX_NPART + cg) * N;
  // loads in batches of 16 per thread, all in flight before the first use (one element at a time
  // every iteration waits out a memory round trip); N * TC is a multiple of 256
  {
    constexpr int UB = 16;
    const int total = N << tcs;
    for (int e0 = tid; e0 < total; e0 += 256 * UB) {
      double zr[UB], zi[UB];
#pragma unroll
      for (int u = 0; u < UB; ++u) {
        const int e = min(e0 + 256 * u, total - 1), k = e >> tcs, tc = e & (TC - 1);
#if HPX_FR_NT & 1
        zr[u] = __builtin_nontemporal_load(&xre[(long)k * TP + c0 + tc]);
        zi[u] = __builtin_nontemporal_load(&xim[(long)k * TP + c0 + tc]);
#else
        zr[u] = xre[(long)k * TP + c0 + tc];
        zi[u] = xim[(long)k * TP + c0 + tc];
#endif
      }
#pragma unroll
      for (int u = 0; u < UB; ++u) {
        const int e = e0 + 256 * u;
        if (e < total) {                                  // uniform over the workgroup
          const int k = e >> tcs, tc = e & (TC - 1);
          const double sg = (k & 1) ? -1.0 : 1.0;
          fre[e] = zr[u] * sg;
          fim[e] = zi[u] * sg;
          double v = zr[u] * zr[u] + zi[u] * zi[u];       // sum over this block's time columns
          for (int o = TC >> 1; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
          if (tc == 0) bp[k] = v;
        }
      }
    }
  }
  int s = 0;
  for (; s + 3 <= logN; s += 3) {
    __syncthreads();
    fft_pass<3, 1>(fre, fim, tw, N, h, logN, s, tcs, tid);
  }
  if (logN - s == 2) {
    __syncthreads();
    fft_pass<2, 1>(fre, fim, tw, N, h, logN, s, tcs, tid);
  } else if (logN - s == 1) {
    __syncthreads();
    fft_pass<1, 1>(fre, fim, tw, N, h, logN, s, tcs, tid);
  }
  __syncthreads();
  const double* dre = A.Dre + (long)b * A.NP * TP;
  const double* dim_ = A.Dim + (long)b * A.NP * TP;
  const double* fmr = A.Fre + (A.fg_shared ? 0 : (long)b * N * M);
  const double* fmi = A.Fim + (A.fg_shared ? 0 : (long)b * N * M);
  const double* ninv = A.ninv + (long)b * N;
  const uint8_t* fl8 = A.flags + (long)b * N;
  double acc = 0.0;
  if (M <= 16) {
    // Model term F f on the matrix pipe: tiles of 16 channels x the block's time columns, K =
    // the (padded) mode index.  The accumulator lane (li, g) then holds channels x0 + g + 4v,
    // column tc = li: the residual is finished from there with a handful of vector ops per
    // element instead of 8 per mode.
    const int wave = tid >> 6, lane = tid & 63, li = lane & 15, g = lane >> 4;
    double bfr[4], bfi[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {                  // B[m = 4 ks + g][tc = li] = f[m][tc]
      const int m = 4 * ks + g;
      const bool ok = (m < M) && (li < TC);
      bfr[ks] = ok ? lfr[(m << tcs) + li] : 0.0;
      bfi[ks] = ok ? lfi[(m << tcs) + li] : 0.0;
    }
    const int t = c0 + li;
    // Tiles run over channels in NATURAL order, so that everything in global memory (data, mode
    // rows, noise, outputs) is touched with unit stride; the bit reversal of the FFT output is
    // undone by the LDS read of s instead.  The global operands of the next tile are requested
    // before the current one is worked on (two workgroups per CU: nothing else hides them).
    const int ntile = N >> 4;
    const int tlast = wave + 4 * ((ntile - 1 - wave) >> 2);       // this wave's last tile
    const bool tvalid = (li < TC) && (t < T);
    double nfr[4], nfi[4], ndr[4], ndi[4], nnv[4], nw[4];
#if HPX_FR_NT & 2
#define HPX_FR_LDD(p_, o_) __builtin_nontemporal_load(&(p_)[o_])
#else
#define HPX_FR_LDD(p_, o_) (p_)[o_]
#endif
#define HPX_FR_LOAD(xt_)                                                              \
  {                                                                                   \
    const int x0_ = (xt_) << 4;                                                       \
    _Pragma("unroll") for (int ks = 0; ks < 4; ++ks) {                                \
      const int m = 4 * ks + g;                  /* unconditional (clamped) load, then the  */ \
      const long fo_ = (long)(x0_ + li) * M + min(m, M - 1);      /* select: no branch      */ \
      const double fr_ = fmr[fo_], fi_ = fmi[fo_];                                    \
      nfr[ks] = (m < M) ? fr_ : 0.0;                                                  \
      nfi[ks] = (m < M) ? fi_ : 0.0;                                                  \
    }                                                                                 \
    _Pragma("unroll") for (int v = 0; v < 4; ++v) {                                   \
      const int x_ = x0_ + HPX_ACC_ROW(g, v);                                         \
      const long o_ = (long)x_ * TP + (tvalid ? t : 0);                               \
      ndr[v] = HPX_FR_LDD(dre, o_);                                                   \
      ndi[v] = HPX_FR_LDD(dim_, o_);                                                  \
      nnv[v] = ninv[x_];                                                              \
      nw[v] = fl8[x_] ? 1.0 : 0.0;                                                    \
    }                                                                                 \
  }
    if (wave < ntile) HPX_FR_LOAD(wave)
    for (int xt = wave; xt < ntile; xt += 4) {
      const int x0 = xt << 4;
      double cfr[4], cfi[4], cdr[4], cdi[4], cnv[4], cw[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        cfr[q] = nfr[q]; cfi[q] = nfi[q]; cdr[q] = ndr[q]; cdi[q] = ndi[q]; cnv[q] = nnv[q]; cw[q] = nw[q];
      }
      HPX_FR_LOAD(min(xt + 4, tlast))                 // branch-free: re-read at the end
      d4 mr = {0., 0., 0., 0.}, mi = {0., 0., 0., 0.};
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {                // A[channel x0 + li][m = 4 ks + g] = F[x][m]
        if (4 * ks >= M) break;                       // k-steps made of padding only (M <= 12: one in four)
        mr = mfma64(cfr[ks], bfr[ks], mr);
        mr = mfma64(-cfi[ks], bfi[ks], mr);
        mi = mfma64(cfr[ks], bfi[ks], mi);
        mi = mfma64(cfi[ks], bfr[ks], mi);
      }
      if (li >= TC) continue;
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        const int x = x0 + HPX_ACC_ROW(g, v);
        const int pidx = (int)(__brev((unsigned)x) >> (32 - logN));
        const long o = (long)x * TP + t;
        if (t >= T) {
          if (A.any_flags) { A.Gre[(long)b * A.NP * TP + o] = 0.0; A.Gim[(long)b * A.NP * TP + o] = 0.0; }
          continue;
        }
        const double sc = (x & 1) ? -A.isn : A.isn;
        const double sr = fre[(pidx << tcs) + li] * sc, si = fim[(pidx << tcs) + li] * sc;
        const double rr = cdr[v] - (sr + mr[v]), ri = cdi[v] - (si + mi[v]);
        const double w = cw[v];
        const double c2 = (rr * rr + ri * ri) * cnv[v];
        acc += w * c2;
        if (A.any_flags) {
          A.Gre[(long)b * A.NP * TP + o] = w * sr;
          A.Gim[(long)b * A.NP * TP + o] = w * si;
        }
        if (A.cr_out) {
          double* q = A.cr_out + (long)b * A.cr_bstride + ((long)t * N + x) * 2;
          q[0] = sr;
          q[1] = si;
        }
        if (A.chisq_out) A.chisq_out[(long)b * A.chisq_bstride + (long)t * N + x] = c2;
      }
    }
#undef HPX_FR_LOAD
  } else {
    // 256 elements (256 / TC channels x TC times) per round; the foreground-mode rows F[x][:] of
    // the round's channels are staged in LDS first (TC threads share a channel: from global
    // memory each of them would fetch the same 2 M values again)
    double* sfr = lfi + (M << tcs);                   // [256 / TC][M]
    double* sfi = sfr + (256 >> tcs) * M;
    const int xper = 256 >> tcs;
    for (int e0 = 0; e0 < (N << tcs); e0 += 256) {
      __syncthreads();
      for (int q = tid; q < xper * M; q += 256) {
        const int xi = q / M, m = q - xi * M;
        const int xx = (int)(__brev((unsigned)((e0 >> tcs) + xi)) >> (32 - logN));
        sfr[q] = fmr[(long)xx * M + m];
        sfi[q] = fmi[(long)xx * M + m];
      }
      __syncthreads();
      const int e = e0 + tid;
      const int pidx = e >> tcs, tc = e & (TC - 1), t = c0 + tc;
      const int x = (int)(__brev((unsigned)pidx) >> (32 - logN));
      const long o = (long)x * TP + t;
      if (t >= T) {
        if (A.any_flags) { A.Gre[(long)b * A.NP * TP + o] = 0.0; A.Gim[(long)b * A.NP * TP + o] = 0.0; }
        continue;
      }
      const double sc = (x & 1) ? -A.isn : A.isn;
      const double sr = fre[e] * sc, si = fim[e] * sc;
      double mr = sr, mi = si;
      const double* myfr = sfr + (tid >> tcs) * M;
      const double* myfi = sfi + (tid >> tcs) * M;
      for (int m = 0; m < M; ++m) {
        const double fr = myfr[m], fi = myfi[m];
        const double gr = lfr[(m << tcs) + tc], gi = lfi[(m << tcs) + tc];
        mr += gr * fr - gi * fi;
        mi += gr * fi + gi * fr;
      }
      const double rr = dre[o] - mr, ri = dim_[o] - mi;
      const double w = fl8[x] ? 1.0 : 0.0;
      const double c2 = (rr * rr + ri * ri) * ninv[x];
      acc += w * c2;
      if (A.any_flags) {
        A.Gre[(long)b * A.NP * TP + o] = w * sr;
        A.Gim[(long)b * A.NP * TP + o] = w * si;
      }
      if (A.cr_out) {
        double* q = A.cr_out + (long)b * A.cr_bstride + ((long)t * N + x) * 2;
        q[0] = sr;
        q[1] = si;
      }
      if (A.chisq_out) A.chisq_out[(long)b * A.chisq_bstride + (long)t * N + x] = c2;
    }
  }
  if (A.fg_out) {
    for (int e = tid; e < (M << tcs); e += 256) {
      const int m = e >> tcs, t = c0 + (e & (TC - 1));
      if (t >= T) continue;
      double* q = A.fg_out + (long)b * A.fg_bstride + ((long)t * M + m) * 2;
      q[0] = lfr[e];
      q[1] = lfi[e];
    }
  }
  const double total = block_sum(acc, red);
  if (tid == 0) A.lnpart[(long)b * HPX_NPART + cg] = total;
}

// The same for channel counts without an in-LDS FFT (N not a power of two, e.g. the 120 channels of
// the reference's test data) and small enough for the dense transform to be cheap (NP <= 256): s = U z
// as a contraction with conj(Fop) on the MFMA, tile by tile (16 channels x 16 times), and each tile
// goes straight on to the model term, residual, chi^2 and outputs from its accumulators -- s is not
// written to and read back from HBM, and one launch replaces k_dft + k_resid.  Wave w of block j owns
// the channel tile 4 j + w and sweeps the time tiles; it also forms sum_t |z|^2 of its own channels.
__global__ __launch_bounds__(256) void k_dft_resid(const ResArgs A) {
  __shared__ double red[4];
  const int b = blockIdx.y, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, li = lane & 15, g = lane >> 4;
  const int N = A.N, M = A.M, T = A.T, TP = A.TP, NP = A.NP, TT = TP >> 4;
  const double* xre = A.Xre + (long)b * A.npad * TP;
  const double* xim = A.Xim + (long)b * A.npad * TP;
  const double* dre = A.Dre + (long)b * NP * TP;
  const double* dim_ = A.Dim + (long)b * NP * TP;
  const double* fmr = A.Fre + (A.fg_shared ? 0 : (long)b * N * M);
  const double* fmi = A.Fim + (A.fg_shared ? 0 : (long)b * N * M);
  const double* ninv = A.ninv + (long)b * N;
  const uint8_t* fl8 = A.flags + (long)b * N;
  const int xt = blockIdx.x * 4 + wave, x0 = xt << 4;
  double acc = 0.0;
  if (x0 < NP) {
    // A operand of the model term, F[x0 + li][m = 4 ks + g]: the same for every time tile
    double cfr[4], cfi[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const int m = 4 * ks + g;
      const long fo = (long)min(x0 + li, N - 1) * (M > 0 ? M : 1) + min(m, (M > 0 ? M : 1) - 1);
      const double fr = (M > 0) ? fmr[fo] : 0.0, fi = (M > 0) ? fmi[fo] : 0.0;
      const bool ok = (m < M) && (x0 + li < N);
      cfr[ks] = ok ? fr : 0.0;
      cfi[ks] = ok ? fi : 0.0;
    }
    double cnv[4], cw[4], zz[4];
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      const int x = min(x0 + HPX_ACC_ROW(g, v), N - 1);
      cnv[v] = ninv[x];
      cw[v] = fl8[x] ? 1.0 : 0.0;
      zz[v] = 0.0;
    }
    const int nks = NP >> 2;                       // a multiple of 4
    for (int tt = 0; tt < TT; ++tt) {
      const int t = (tt << 4) + li;
      // s^[x][t] = sum_k conj(Fop)[x][k] z[k][t]: A[x0 + li][k = 4 ks + g], B[k][t]; two operand sets
      d4 sr = {0., 0., 0., 0.}, si = {0., 0., 0., 0.};
      double w0r, w0i, w1r, w1i, b0r, b0i, b1r, b1i;
#define HPX_DR_LOAD(wr_, wi_, br_, bi_, ks_)                          \
  {                                                                   \
    const int k_ = 4 * (ks_) + g;                                     \
    wr_ = A.twre[(long)k_ * NP + x0 + li];                            \
    wi_ = A.twim[(long)k_ * NP + x0 + li];                            \
    br_ = xre[(long)k_ * TP + t];                                     \
    bi_ = xim[(long)k_ * TP + t];                                     \
  }
#define HPX_DR_MMA(wr_, wi_, br_, bi_)    /* conj(W) z */             \
  {                                                                   \
    sr = mfma64(wr_, br_, sr);                                        \
    sr = mfma64(wi_, bi_, sr);                                        \
    si = mfma64(wr_, bi_, si);                                        \
    si = mfma64(-wi_, br_, si);                                       \
  }
      HPX_DR_LOAD(w0r, w0i, b0r, b0i, 0)
      for (int ks = 0; ks < nks; ks += 2) {
        HPX_DR_LOAD(w1r, w1i, b1r, b1i, ks + 1)
        __builtin_amdgcn_sched_barrier(0);
        HPX_DR_MMA(w0r, w0i, b0r, b0i)
        __builtin_amdgcn_sched_barrier(0);
        HPX_DR_LOAD(w0r, w0i, b0r, b0i, min(ks + 2, nks - 1))
        __builtin_amdgcn_sched_barrier(0);
        HPX_DR_MMA(w1r, w1i, b1r, b1i)
        __builtin_amdgcn_sched_barrier(0);
      }
#undef HPX_DR_LOAD
#undef HPX_DR_MMA
      // model term F f: B[m = 4 ks + g][t] = f[m][t] (rows N + m of X)
      d4 mr = {0., 0., 0., 0.}, mi = {0., 0., 0., 0.};
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        if (4 * ks >= M) break;
        const int m = min(4 * ks + g, M - 1);
        const double fr = xre[(long)(N + m) * TP + t], fi = xim[(long)(N + m) * TP + t];
        const double br = (4 * ks + g < M) ? fr : 0.0, bi = (4 * ks + g < M) ? fi : 0.0;
        mr = mfma64(cfr[ks], br, mr);
        mr = mfma64(-cfi[ks], bi, mr);
        mi = mfma64(cfr[ks], bi, mi);
        mi = mfma64(cfi[ks], br, mi);
      }
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        const int x = x0 + HPX_ACC_ROW(g, v);
        const long o = (long)min(x, NP - 1) * TP + t;
        const double zr = xre[o], zi = xim[o];             // this channel's z (rows < N of X)
        if (x < N) zz[v] += zr * zr + zi * zi;
        if (x >= N) continue;
        if (t >= T) {
          if (A.any_flags) { A.Gre[(long)b * NP * TP + o] = 0.0; A.Gim[(long)b * NP * TP + o] = 0.0; }
          continue;
        }
        const double s_r = sr[v] * A.isn, s_i = si[v] * A.isn;
        const double rr = dre[o] - (s_r + mr[v]), ri = dim_[o] - (s_i + mi[v]);
        const double w = cw[v];
        const double c2 = (rr * rr + ri * ri) * cnv[v];
        acc += w * c2;
        if (A.any_flags) {
          A.Gre[(long)b * NP * TP + o] = w * s_r;
          A.Gim[(long)b * NP * TP + o] = w * s_i;
        }
        if (A.cr_out) {
          double* q = A.cr_out + (long)b * A.cr_bstride + ((long)t * N + x) * 2;
          q[0] = s_r;
          q[1] = s_i;
        }
        if (A.chisq_out) A.chisq_out[(long)b * A.chisq_bstride + (long)t * N + x] = c2;
      }
    }
    // sum_t |z_xt|^2 of the tile's channels: over the 16 lanes of a row group, then slot 0 of the
    // partial-sum table (the other slots of these channels are zero: every channel has one owner)
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      double s = zz[v];
      s += __shfl_xor(s, 8, 16);
      s += __shfl_xor(s, 4, 16);
      s += __shfl_xor(s, 2, 16);
      s += __shfl_xor(s, 1, 16);
      const int x = x0 + HPX_ACC_ROW(g, v);
      if (li == 0 && x < N)
        for (int j = 0; j < A.npart; ++j) A.bpart[((long)b * HPX_NPART + j) * N + x] = (j == 0) ? s : 0.0;
    }
  }
  if (A.fg_out && blockIdx.x == 0) {
    for (int e = tid; e < T * M; e += 256) {
      const int t = e / M, m = e % M;
      double* q = A.fg_out + (long)b * A.fg_bstride + (long)e * 2;
      q[0] = xre[(long)(N + m) * TP + t];
      q[1] = xim[(long)(N + m) * TP + t];
    }
  }
  const double total = block_sum(acc, red);
  if (tid == 0) A.lnpart[(long)b * HPX_NPART + blockIdx.x] = total;
}

// betam_k = sum_t |SK[k][t]|^2 (SK = F (w s), in the Z scratch with leading dim ncol).
// Sixteen lanes share a row (consecutive t: one 128-byte segment per load) and reduce by shuffles;
// a thread per row would touch 64 different cache lines with every load.
__global__ __launch_bounds__(256) void k_betam(const double* __restrict__ Kre, const double* __restrict__ Kim,
                                               double* __restrict__ betam, const int N, const int T,
                                               const int NP, const int ncol) {
  const int b = blockIdx.y, c = threadIdx.x & 15, r = threadIdx.x >> 4;
  for (int k0 = blockIdx.x * 16; k0 < N; k0 += gridDim.x * 16) {
    const int k = k0 + r;
    double s = 0.0;
    if (k < N) {
      const long o = ((long)b * NP + k) * ncol;
      for (int t = c; t < T; t += 16) s += Kre[o + t] * Kre[o + t] + Kim[o + t] * Kim[o + t];
    }
#pragma unroll
    for (int m = 8; m >= 1; m >>= 1) s += __shfl_xor(s, m, 16);
    if (c == 0 && k < N) betam[(long)b * N + k] = s;
  }
}

// ---- bandpower draw ---------------------------------------------------------------
// Regularised upper incomplete gamma Q(a, z) for integer a >= 1:
// Q = exp(-z) sum_{k<a} z^k / k!  (= scipy.special.gammaincc(a, z) = invgamma.cdf
// of pspec.py:51 at x = beta/z).  Forward sum for z < a, scaled Horner form
// around the leading term for z >= a (no overflow, all terms positive).
#define HPX_RK_MAX 512
// rk[k] = 1 / k (LDS table, k < a <= HPX_RK_MAX, else NULL): the forward sum multiplies by it instead of dividing --
// an fp64 division is ~10 dependent vector instructions, and the 1000-point CDF grids of the
// prior channels made that loop the bulk of k_draw (one more rounding per term: ~a ulp in Q).
__device__ double igamc_int(const int a, const double z, const double lgam_a, const double* rk) {
  if (!(z > 0.0)) return 1.0;
  if (z < (double)a) {
    double t = 1.0, s = 1.0;
    if (rk) {
      // eight terms per trip: their table reads are issued together (one LDS round trip per term
      // on the dependent chain otherwise, ~10x the latency of the multiply itself)
      int k = 1;
      for (; k + 8 <= a; k += 8) {
        double r[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) r[u] = z * rk[k + u];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          t *= r[u];
          s += t;
        }
      }
      for (; k < a; ++k) {
        t *= z * rk[k];
        s += t;
      }
    } else {                                   // shape beyond the table (Ntimes > HPX_RK_MAX)
      for (int k = 1; k < a; ++k) {
        t *= z / (double)k;
        s += t;
      }
    }
    return exp(-z) * s;
  }
  const double rz = 1.0 / z;
  double s = 1.0;
#pragma unroll 8
  for (int m = 1; m < a; ++m) s = 1.0 + s * ((double)m * rz);
  return exp(-z + (double)(a - 1) * log(z) - lgam_a) * s;
}

// One inversion draw by a group of 16 lanes; pspec.py:50-62.  The reference tabulates the CDF on the
// 1000-point grid, normalises it (cdf -= min; cdf /= max), de-duplicates and interpolates linearly
// at u.  The table is monotone, so min and max are its end points and the bracket
// [first occurrence of the previous distinct value, first value >= u] is found by two 16-ary searches
// (three rounds of one evaluation per lane each) instead of evaluating all 1000 points: ~13
// incomplete-gamma sums per lane and channel instead of 1000 per channel by the whole block, and
// the prior channels of a baseline are sampled side by side (16 groups per workgroup).
// All 16 lanes of the group return the sample.
template <class Pred>
__device__ __forceinline__ int first_true16(const int n, Pred pred) {
  // smallest i in [0, n) with pred(i), n if none; pred is monotone (false ... false true ... true)
  const int j = threadIdx.x & 15, gsh = (threadIdx.x & 63) & ~15;
  int lo = 0, hi = n;                   // the answer is in [lo, hi]; hi < n is known to be true
  while (hi > lo) {
    const int step = (hi - lo + 15) >> 4;
    const int idx = min(lo + (j + 1) * step - 1, hi - 1);
    const bool p = pred(idx);
    const int m = __popc((unsigned)((__ballot(!p) >> gsh) & 0xFFFFull));     // leading false probes
    if (m == 16) break;                 // every probe up to hi - 1 is false: the answer is hi
    const int nlo = lo + m * step;
    hi = min(lo + (m + 1) * step - 1, hi - 1);
    lo = nlo;
  }
  return hi;
}

__device__ double inversion_draw(const int alpha, const double lgam, const double beta, const double u,
                                 const double* __restrict__ xg, const int ngrid, const double* rk) {
  const double mn = igamc_int(alpha, beta / xg[0], lgam, rk);                    // cdf.min()
  const double mx = igamc_int(alpha, beta / xg[ngrid - 1], lgam, rk) - mn;       // (cdf - min).max()
  auto cval = [&](const int i) { return (igamc_int(alpha, beta / xg[i], lgam, rk) - mn) / mx; };
  // searchsorted(unique, u, 'left') in original indexing = number of table values < u
  int hi = first_true16(ngrid, [&](const int i) { return !(cval(i) < u); });
  if (hi >= ngrid) {                    // u above the table: last two distinct values
    const double top = cval(ngrid - 1);
    hi = first_true16(ngrid, [&](const int i) { return !(cval(i) < top); });
  }
  int lo;
  if (hi == 0) {                        // u at/below the first value: first two distinct values
    const double bot = cval(0);
    hi = first_true16(ngrid, [&](const int i) { return !(cval(i) <= bot); });
    lo = 0;
    if (hi >= ngrid) return xg[0];      // degenerate table (all equal): the reference would give NaN
  } else {
    const double below = cval(hi - 1);  // previous distinct value; its first occurrence:
    lo = first_true16(ngrid, [&](const int i) { return !(cval(i) < below); });
  }
  const double clo = cval(lo), chi = cval(hi), xlo = xg[lo], xhi = xg[hi];
  const double slope = (xhi - xlo) / (chi - clo);
  return slope * (u - clo) + xlo;
}

struct DrawArgs {
  const double *bpart, *lnpart, *betam, *uni, *igy, *xgrid, *ps_forced;
  double *beta, *lnp1;
  int npart;
  const int32_t* pmap;
  double *ia, *ps_cur, *ps_out;
  long ps_bstride, forced_bstride;   // strides between baselines in ps_out / ps_forced
  int N, T, ngrid, prior_shared, any_flags;
  double lgam_T;
  double* lnblk;                     // [nbl][ceil(N / 64)]: the second ln-posterior term by blocks of 64 channels
  unsigned* dcount;                  // [nbl]: slices of the baseline that have finished (wraps to 0)
  double* lnpost_out;                // the caller's ln-posterior history, offset to this iteration
  long lnpost_pitch;
};

// Grid (slices, baselines): a baseline's channels are dealt to `gridDim.x` workgroups in blocks of 64 -- one per
// baseline for large batches, up to eight for batches that would leave most CUs idle (config 2).  The ln-posterior's
// sum over the channels is formed per block of 64 (a fixed shuffle tree) and added in block order by the slice
// that finishes last, so the result does not depend on the number of slices.
__global__ __launch_bounds__(256) void k_draw(const DrawArgs A) {
  extern __shared__ double dyn[];     // N ints: the channels with a prior
  __shared__ int pcount;
  __shared__ double rk_s[HPX_RK_MAX];
  const int b = blockIdx.y, tid = threadIdx.x, N = A.N;
  const int nblk = (N + 63) >> 6;
  const int cb0 = (int)(((long)nblk * blockIdx.x) / gridDim.x), cb1 = (int)(((long)nblk * (blockIdx.x + 1)) / gridDim.x);
  const int k0 = cb0 * 64, k1 = min(N, cb1 * 64);
  for (int k = tid; k < HPX_RK_MAX; k += 256) rk_s[k] = 1.0 / (double)(k > 0 ? k : 1);
  const double* rk = (A.T <= HPX_RK_MAX) ? rk_s : nullptr;
  // beta_k = N sum_t |z_kt|^2 from the partial sums of the residual kernel (one slot per block of a baseline
  // there), added in slot order
  double* beta = A.beta + (long)b * N;
  for (int k = k0 + tid; k < k1; k += 256) {
    double sum = 0.0;
    for (int j = 0; j < A.npart; ++j) sum += A.bpart[((long)b * HPX_NPART + j) * N + k];
    beta[k] = (double)N * sum;
  }
  if (tid == 0) pcount = 0;
  __syncthreads();
  const double* bm = A.any_flags ? A.betam + (long)b * N : beta;
  const int32_t* pmap = A.pmap + (A.prior_shared ? 0 : (long)b * N);
  double* ps_out = A.ps_out + (long)b * A.ps_bstride;
  // channels without a prior: x = beta * invgamma.ppf(U, a=T-1)   (pspec.py:125)
  // channels with a prior: truncated draw with shape alpha+1 = T   (pspec.py:121-123).
  // They are collected first (a scan of pmap by one thread per channel would be N dependent
  // global loads); each draw depends only on its own channel, so their order is immaterial.
  int* plist = reinterpret_cast<int*>(dyn);
  for (int k = k0 + tid; k < k1; k += 256) {
    if (pmap[k] < 0) ps_out[k] = A.igy[k] * beta[k];
    else plist[atomicAdd(&pcount, 1)] = k;
  }
  __syncthreads();
  const int np = pcount;
  for (int i = tid >> 4; i < np; i += 16) {            // one prior channel per group of 16 lanes
    const int k = plist[i], row = pmap[k];
    const double v = inversion_draw(A.T, A.lgam_T, beta[k], A.uni[k], A.xgrid + (long)row * A.ngrid,
                                    A.ngrid, rk);
    if ((tid & 15) == 0) ps_out[k] = v;
  }
  __syncthreads();
  // second ln-posterior term, the next 1 / a: one wave per block of 64 channels
  __shared__ double part[64];
  for (int cb = cb0 + (tid >> 6); cb < cb1; cb += 4) {
    const int k = cb * 64 + (tid & 63);
    double v = 0.0;
    if (k < N) {
      const double pn = ps_out[k];
      v = bm[k] / pn;
      const double nx = A.ps_forced ? A.ps_forced[(long)b * A.forced_bstride + k] : pn;
      A.ps_cur[(long)b * N + k] = nx;
      A.ia[(long)b * N + k] = inv_a(nx, (double)N);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o);
    if ((tid & 63) == 0) part[(cb - cb0) & 63] = v;          // (a slice holds at most 64 blocks: N <= 4096 per slice)
  }
  __syncthreads();
  if (tid != 0) return;
  double tot = 0.0;                  // chi^2 total of the residual kernel's partial sums, in slot order
  for (int j = 0; j < A.npart; ++j) tot += A.lnpart[(long)b * HPX_NPART + j];
  double s = 0.0;
  if (gridDim.x == 1) {              // one slice: nothing to hand over
    for (int cb = 0; cb < nblk; ++cb) s += part[cb];
  } else {
    // several slices: this slice's block sums go to memory behind an agent-scope release; the slice that counts
    // itself in last acquires and adds all blocks in block order
    double* gl = A.lnblk + (long)b * nblk;
    for (int cb = cb0; cb < cb1; ++cb) __hip_atomic_store(&gl[cb], part[cb - cb0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    const unsigned old = atomicInc(&A.dcount[b], gridDim.x - 1);     // (wraps: zero again after the last slice)
    if (old != gridDim.x - 1) return;
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    for (int cb = 0; cb < nblk; ++cb) s += __hip_atomic_load(&gl[cb], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  const double lnp = -tot - s;       // -> ln posterior
  A.lnp1[b] = lnp;
  if (A.lnpost_out) A.lnpost_out[(long)b * A.lnpost_pitch] = lnp;
}

__global__ void k_inv_test(const int alpha, const double lgam, const double* __restrict__ beta,
                           const double* __restrict__ u, const double* __restrict__ xgrid,
                           const int ngrid, double* __restrict__ out) {
  __shared__ double rk_s[HPX_RK_MAX];
  const int i = blockIdx.x;
  for (int k = threadIdx.x; k < HPX_RK_MAX; k += 256) rk_s[k] = 1.0 / (double)(k > 0 ? k : 1);
  __syncthreads();
  if (threadIdx.x < 16) {
    const double v = inversion_draw(alpha, lgam, beta[i], u[i], xgrid + (long)i * ngrid, ngrid,
                                    alpha <= HPX_RK_MAX ? rk_s : nullptr);
    if (threadIdx.x == 0) out[i] = v;
  }
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

}  // namespace

// ---------------------------------------------------------------------------
static int plan_create_impl(hpx_plan** out, int nbl, int T, int N, int M, int extra_rhs);
extern "C" int hpx_plan_create(hpx_plan** out, int nbl, int T, int N, int M) {
  HPX_REQUIRE(out, "hpx_plan_create: null out");
  HPX_REQUIRE(nbl > 0 && T > 1 && N > 0 && M >= 0, "hpx_plan_create: need nbl>0, T>1, N>0, M>=0");
  return plan_create_impl(out, nbl, T, N, M, 0);
}
extern "C" int hpx_plan_create_ex(hpx_plan** out, int nbl, int T, int N, int M, int extra_rhs) {
  HPX_REQUIRE(out, "hpx_plan_create_ex: null out");
  HPX_REQUIRE(nbl > 0 && T > 1 && N > 0 && M >= 0 && extra_rhs >= 0 && extra_rhs <= N,
              "hpx_plan_create_ex: need nbl>0, T>1, N>0, M>=0, 0 <= extra_rhs <= N");
  return plan_create_impl(out, nbl, T, N, M, extra_rhs);
}
static int plan_create_impl(hpx_plan** out, int nbl, int T, int N, int M, int extra_rhs) {
  hpx_plan* p = new hpx_plan();
  p->nbl = nbl; p->T = T; p->N = N; p->M = M;
  p->n = N + M;
  p->npad = ceil16(p->n);
  p->TP = ceil16(T + extra_rhs);      // right-hand-side columns: the times (+ one per flagged channel, dense noise)
  p->ld = p->npad + p->TP;
  p->NP = ceil16(N);
  p->MP = ceil16(M > 0 ? M : 1);
  p->ncolR = p->TP + p->MP + 16;
  p->nblk = (p->npad + HPX_NB - 1) / HPX_NB;
  p->lgam_T = lgamma((double)T);
  p->ev_used = 0;
  p->allow_split = 1;
  const size_t nb = nbl, lsz = (size_t)p->npad * p->ld, xsz = (size_t)p->npad * p->TP,
               ssz = (size_t)p->NP * p->TP, rsz = (size_t)p->NP * p->ncolR;
  int rc = HPX_OK;
#define A_(ptr, cnt) if (rc == HPX_OK) rc = dev_alloc(p, &p->ptr, (cnt))
  A_(L, nb * lsz * 2);
  A_(Wre, nb * p->nblk * 1024); A_(Wim, nb * p->nblk * 1024);
  A_(Vt, nb * HPX_VT_STRIDE(p->npad));
  if (rc == HPX_OK && hipMemset(p->Vt, 0, nb * HPX_VT_STRIDE(p->npad) * sizeof(double)) != hipSuccess) rc = HPX_EHIP;
  A_(Xre, nb * xsz); A_(Xim, nb * xsz);
  A_(info, nb);
  A_(ia, nb * N); A_(ps_cur, nb * N); A_(beta, nb * N); A_(betam, nb * N); A_(lnp1, nb);
  A_(lnblk, nb * ((N + 63) / 64)); A_(dcount, nb);
  A_(bpart, nb * HPX_NPART * N); A_(lnpart, nb * HPX_NPART);
  A_(Rre, nb * rsz); A_(Rim, nb * rsz); A_(Zre, nb * rsz); A_(Zim, nb * rsz);
  A_(Cre, nb * N); A_(Cim, nb * N);
  A_(P2re, ssz); A_(P2im, ssz);
  {
    const size_t rmin = 32 * (size_t)(N / 32);
    A_(E, nb * ((size_t)(p->ld - rmin) / 16) * rmin * 32 + 8);
    A_(P2Tre, (size_t)(p->TP / 16) * p->NP * 16); A_(P2Tim, (size_t)(p->TP / 16) * p->NP * 16);
  }
  A_(Hre, nb * M * M); A_(Him, nb * M * M);
  A_(P4re, nb * M * p->TP); A_(P4im, nb * M * p->TP);
  A_(Fopre, (size_t)p->NP * p->NP); A_(Fopim, (size_t)p->NP * p->NP);
  A_(Dre, nb * ssz); A_(Dim, nb * ssz); A_(Sre, nb * ssz); A_(Sim, nb * ssz);
  A_(Gre, nb * ssz); A_(Gim, nb * ssz);
  A_(Fre, nb * N * (M > 0 ? M : 1)); A_(Fim, nb * N * (M > 0 ? M : 1));
  A_(ninv, nb * N); A_(ni, nb * N);
  A_(flags, nb * N);
  A_(pmap, nb * N);
#undef A_
  if (rc != HPX_OK) { hpx_plan_destroy(p); return rc; }
  hipError_t e = hipMemset(p->Xre, 0, nb * xsz * sizeof(double));
  if (e == hipSuccess) e = hipMemset(p->Xim, 0, nb * xsz * sizeof(double));
  if (e == hipSuccess) e = hipMemset(p->P2re, 0, ssz * sizeof(double));
  if (e == hipSuccess) e = hipMemset(p->P2im, 0, ssz * sizeof(double));
  if (e == hipSuccess) e = hipMemset(p->info, 0, nb * sizeof(int32_t));
  if (e == hipSuccess) e = hipMemset(p->dcount, 0, nb * sizeof(unsigned));
  if (e == hipSuccess) e = hipDeviceSynchronize();   // null-stream memsets vs. the caller's (non-blocking) streams
  if (e != hipSuccess) {
    hpx_set_error("hpx_plan_create: memset failed: %s", hipGetErrorString(e));
    hpx_plan_destroy(p);
    return HPX_EHIP;
  }
  *out = p;
  return HPX_OK;
}

extern "C" int hpx_plan_destroy(hpx_plan* p) {
  if (!p) return HPX_OK;
  if (p->child) { hpx_plan_destroy(p->child); p->child = nullptr; }
  for (auto& q : p->allocs) (void)hipFree(q.first);
  for (hipEvent_t ev : p->events) (void)hipEventDestroy(ev);
  delete p;
  return HPX_OK;
}

extern "C" int64_t hpx_plan_bytes(const hpx_plan* p) {
  return p ? p->bytes + (p->child ? p->child->bytes : 0) : 0;
}

extern "C" int hpx_plan_dims(const hpx_plan* p, int* npad, int* tpad, int* ld) {
  HPX_REQUIRE(p, "hpx_plan_dims: null plan");
  if (npad) *npad = p->npad;
  if (tpad) *tpad = p->TP;
  if (ld) *ld = p->ld;
  return HPX_OK;
}

static hpx_gen_batch gen_of(const hpx_plan* p);
// dense noise with flags (hpx_plan_set_static_dense_flagged): unit vectors of the flagged channels into the
// padded time columns T .. T+f-1 of the operand block, so that Z = Ninv [d | e_j ..] carries the Woodbury
// vectors P = B^H Ninv E through the same transforms as the data (their omega / P2 parts stay zero)
__global__ void k_wb_inject(double* __restrict__ Rre, const int32_t* __restrict__ flist,
                            const int32_t* __restrict__ fcount, const int fmax, const int T, const int NP,
                            const int ncol) {
  const int b = blockIdx.x;
  for (int kf = threadIdx.x; kf < fcount[b]; kf += blockDim.x)
    Rre[((long)b * NP + flist[(long)b * fmax + kf]) * ncol + T + kf] = 1.0;
}

static int set_static_impl(hpx_plan* p, const double* vis, const uint8_t* flags,
                           const double* ninv, const double* ninv_dense, const double* nih_dense,
                           int noise_shared, const double* fgmodes, int fg_shared,
                           const int32_t* prior_map, const double* xgrid, int nxrows,
                           int prior_shared, int ngrid, const double* omega,
                           const double* fop, int any_flags, void* stream, int wb = 0) {
  HPX_REQUIRE(p && vis && flags && (ninv || ninv_dense) && fop && prior_map, "hpx_plan_set_static: null argument");
  HPX_REQUIRE(p->M == 0 || fgmodes, "hpx_plan_set_static: fgmodes required when M > 0");
  HPX_REQUIRE(nxrows == 0 || (xgrid && ngrid >= 2 && ngrid <= 8192),
              "hpx_plan_set_static: bad prior grid");
  hipStream_t st = (hipStream_t)stream;
  const int nbl = p->nbl, N = p->N, M = p->M, T = p->T, NP = p->NP, TP = p->TP, MP = p->MP;
  p->fg_shared = fg_shared ? 1 : 0;
  p->prior_shared = prior_shared ? 1 : 0;
  p->has_omega = omega ? 1 : 0;
  p->any_flags = any_flags ? 1 : 0;
  p->ngrid = ngrid;
  p->nxrows = nxrows;
  HPX_HIP(hipMemcpyAsync(p->flags, flags, (size_t)nbl * N, hipMemcpyDeviceToDevice, st));
  p->dense_noise = ninv_dense ? (wb ? 2 : 1) : 0;
  hpx_devbuf ones, tmp;                    // dense noise only
  if (ninv_dense) {
    HPX_REQUIRE(nih_dense && (!any_flags || wb),
                "hpx_plan_set_static_dense: needs sqrtm(Ninv) and unflagged data (the reference's column-masked "
                "Ni = Ninv diag(w) is not Hermitian, pspec.py:361: hpx_plan_set_static_dense_flagged)");
    if (wb) {      // flagged channels per baseline (host lists), correction systems, masked residual
      std::vector<uint8_t> hf((size_t)nbl * N);
      HPX_HIP(hipMemcpyAsync(hf.data(), flags, hf.size(), hipMemcpyDeviceToHost, st));
      HPX_HIP(hipStreamSynchronize(st));
      std::vector<int32_t> cnt(nbl, 0);
      int fmax = 0;
      for (int b = 0; b < nbl; ++b) {
        for (int j = 0; j < N; ++j) cnt[b] += hf[(size_t)b * N + j] ? 0 : 1;
        fmax = std::max(fmax, cnt[b]);
      }
      HPX_REQUIRE(T + fmax <= TP, "hpx_plan_set_static_dense_flagged: the plan has too few right-hand-side columns "
                                  "(hpx_plan_create_ex with extra_rhs >= the largest number of flagged channels)");
      HPX_REQUIRE(fmax <= 512, "hpx_plan_set_static_dense_flagged: at most 512 flagged channels per baseline");
      {   // the residual kernel keeps (2 M + N / (16 slices)) x TP doubles in LDS (k_resid): say so HERE, not at the first run
        int P = 4;                       // (the slices of k_resid's launch, post_solve)
        while (P > 1 && (N % P != 0 || N / P < 64)) P >>= 1;
        const size_t lds = (size_t)(2 * M * TP + (N / P) * (TP / 16)) * sizeof(double);
        if (lds > (size_t)160 * 1024) {
          hpx_set_error("hpx_plan_set_static_dense_flagged: %d right-hand-side columns (%d times + %d flagged channels, "
                        "padded) need %zu bytes of LDS in the residual kernel, the CU has 160 KiB: at Nfreqs = %d and "
                        "%d modes at most %d columns", TP, T, fmax, lds, N, M,
                        (int)((160 * 1024 / sizeof(double)) / (2 * M + N / 16.0)) / 16 * 16);
          return HPX_EINVAL;
        }
      }
      p->wb_fmax = fmax > 0 ? fmax : 1;
      std::vector<int32_t> list((size_t)nbl * p->wb_fmax, 0);
      for (int b = 0; b < nbl; ++b) {
        int k = 0;
        for (int j = 0; j < N; ++j)
          if (!hf[(size_t)b * N + j]) list[(size_t)b * p->wb_fmax + k++] = j;
      }
      HPX_TRY(dev_alloc(p, &p->wb_flist, list.size()));
      HPX_TRY(dev_alloc(p, &p->wb_fcount, (size_t)nbl));
      HPX_TRY(dev_alloc(p, &p->wb_W, (size_t)nbl * p->wb_fmax * (p->wb_fmax + T) * 2));
      HPX_TRY(dev_alloc(p, &p->RDre, (size_t)nbl * NP * TP));
      HPX_TRY(dev_alloc(p, &p->RDim, (size_t)nbl * NP * TP));
      HPX_HIP(hipMemcpy(p->wb_flist, list.data(), list.size() * sizeof(int32_t), hipMemcpyHostToDevice));
      HPX_HIP(hipMemcpy(p->wb_fcount, cnt.data(), cnt.size() * sizeof(int32_t), hipMemcpyHostToDevice));
    }
    const size_t msz = (size_t)nbl * NP * NP;
    HPX_TRY(dev_alloc(p, &p->NIre, msz)); HPX_TRY(dev_alloc(p, &p->NIim, msz));
    HPX_TRY(dev_alloc(p, &p->CDre, msz)); HPX_TRY(dev_alloc(p, &p->CDim, msz));
    hipLaunchKernelGGL(k_dense_planar, dim3(64, nbl), dim3(256), 0, st, ninv_dense, noise_shared, p->NIre, p->NIim, N, NP);
    hipLaunchKernelGGL(k_take_diag, dim3(4, nbl), dim3(256), 0, st, p->NIre, p->ninv, N, NP);   // chi^2 uses Ninv.diagonal()
    HPX_HIP(hipGetLastError());
    HPX_TRY(ones.alloc((size_t)nbl * N));
    std::vector<double> h1((size_t)nbl * N, 1.0);
    HPX_HIP(hipMemcpyAsync(ones.p, h1.data(), h1.size() * sizeof(double), hipMemcpyHostToDevice, st));
    HPX_HIP(hipStreamSynchronize(st));     // h1 goes out of scope with this block's caller frame only at return; be explicit
  } else {
    HPX_HIP(hipMemcpyAsync(p->ninv, ninv, (size_t)nbl * N * sizeof(double), hipMemcpyDeviceToDevice, st));
  }
  HPX_HIP(hipMemcpyAsync(p->pmap, prior_map, (size_t)(prior_shared ? 1 : nbl) * N * sizeof(int32_t),
                         hipMemcpyDeviceToDevice, st));
  if (nxrows > 0) {
    HPX_TRY(dev_alloc(p, &p->xgrid, (size_t)nxrows * ngrid));
    HPX_HIP(hipMemcpyAsync(p->xgrid, xgrid, (size_t)nxrows * ngrid * sizeof(double),
                           hipMemcpyDeviceToDevice, st));
  }
  HPX_TRY(hpx_fop_to_planar(fop, p->Fopre, p->Fopim, N, NP, st));
  const double* fgp = fgmodes ? fgmodes : vis;   // never dereferenced when M == 0
  if (M > 0) {
    const long tot = (long)(fg_shared ? 1 : nbl) * N * M;
    hipLaunchKernelGGL(k_fg_planar, dim3(256), dim3(256), 0, st, fgmodes, p->Fre, p->Fim, tot);
    HPX_HIP(hipGetLastError());
  }
  const double isn = 1.0 / sqrt((double)N);
  if (!ninv_dense) {
    hipLaunchKernelGGL(k_prep, dim3(128, nbl), dim3(256), 0, st, vis, flags, ninv, fgp, p->fg_shared,
                       omega, p->Zre, p->Zim, p->Dre, p->Dim, p->ni, T, N, M, NP, TP, MP, p->ncolR, p->omega_mod, 0);
    HPX_HIP(hipGetLastError());
  } else {
    // Z = Ninv [d | F | .] + Ninv^1/2 [omega_b | 0]: the operand block with unit weights (into R as
    // scratch), then two dense products on the MFMA (the stored planar matrices are Hermitian:
    // buffer[k][x] = conj(W[x][k]), hence conjW = 1)
    const long mstr = (long)NP * NP, zstr = (long)NP * p->ncolR;
    hipLaunchKernelGGL(k_prep, dim3(128, nbl), dim3(256), 0, st, vis, flags, ones.p, fgp, p->fg_shared,
                       (const double*)nullptr, p->Rre, p->Rim, p->Dre, p->Dim, p->ni, T, N, M, NP, TP, MP, p->ncolR, 0,
                       wb);
    if (wb) hipLaunchKernelGGL(k_wb_inject, dim3(nbl), dim3(256), 0, st, p->Rre, p->wb_flist, p->wb_fcount,
                               p->wb_fmax, T, NP, p->ncolR);
    HPX_HIP(hipGetLastError());
    HPX_TRY(hpx_launch_dft(nbl, NP, p->ncolR, p->NIre, p->NIim, 1, p->Rre, p->Rim, zstr, p->ncolR, nullptr, 0,
                           p->Zre, p->Zim, zstr, p->ncolR, 1.0, st, 0, mstr));
    hipLaunchKernelGGL(k_take_diag, dim3(4, nbl), dim3(256), 0, st, p->NIre, p->ni, N, NP);
    if (omega) {
      const size_t msz = (size_t)nbl * NP * NP, osz = (size_t)nbl * NP * TP;
      HPX_TRY(tmp.alloc((wb ? 4 : 2) * msz + 4 * osz));
      double *hre = tmp.p, *him = hre + msz, *ore = him + msz, *oim = ore + osz, *ure = oim + osz, *uim = ure + osz;
      if (wb) {     // sqrtm of the column-masked Ni is a general matrix: the product below (conjW = 1) wants its
                    // conjugate transpose stored; one matrix per baseline
        double *gre = uim + osz, *gim = gre + msz;
        hipLaunchKernelGGL(k_dense_planar, dim3(64, nbl), dim3(256), 0, st, nih_dense, 0, gre, gim, N, NP);
        hipLaunchKernelGGL(k_conj_transpose, dim3(64, nbl), dim3(256), 0, st, gre, gim, hre, him, NP);
      } else
      hipLaunchKernelGGL(k_dense_planar, dim3(64, nbl), dim3(256), 0, st, nih_dense, noise_shared, hre, him, N, NP);
      hipLaunchKernelGGL(k_prep_omega_b, dim3(32, nbl), dim3(256), 0, st, omega, ore, oim, T, N, NP, TP, p->omega_mod);
      HPX_HIP(hipGetLastError());
      HPX_TRY(hpx_launch_dft(nbl, NP, TP, hre, him, 1, ore, oim, (long)NP * TP, TP, nullptr, 0, ure, uim,
                             (long)NP * TP, TP, 1.0, st, 0, mstr));
      hipLaunchKernelGGL(k_add_block, dim3(32, nbl), dim3(256), 0, st, p->Zre, p->Zim, zstr, p->ncolR, ure, uim,
                         (long)NP * TP, TP, NP, TP);
      HPX_HIP(hipGetLastError());
    }
    // C = U^H Ninv U = F Ninv F^H / N = F (F Ninv)^H / N  (C is Hermitian): two transforms and a
    // conjugate transpose; CD doubles as scratch for F Ninv
    HPX_TRY(hpx_launch_dft(nbl, NP, NP, p->Fopre, p->Fopim, 0, p->NIre, p->NIim, mstr, NP, nullptr, 0,
                           p->CDre, p->CDim, mstr, NP, 1.0, st, N == NP));
    hpx_devbuf a1h;
    HPX_TRY(a1h.alloc(2 * (size_t)nbl * NP * NP));
    hipLaunchKernelGGL(k_conj_transpose, dim3(64, nbl), dim3(256), 0, st, p->CDre, p->CDim, a1h.p,
                       a1h.p + (size_t)nbl * NP * NP, NP);
    HPX_HIP(hipGetLastError());
    HPX_TRY(hpx_launch_dft(nbl, NP, NP, p->Fopre, p->Fopim, 0, a1h.p, a1h.p + (size_t)nbl * NP * NP, mstr, NP,
                           nullptr, 0, p->CDre, p->CDim, mstr, NP, 1.0 / (double)N, st, N == NP));
    HPX_HIP(hipStreamSynchronize(st));     // a1h is released here
  }
  // R = U^H Z = F Z / sqrt(N)
  HPX_TRY(hpx_launch_dft(nbl, NP, p->ncolR, p->Fopre, p->Fopim, 0, p->Zre, p->Zim,
                         (long)NP * p->ncolR, p->ncolR, nullptr, 0, p->Rre, p->Rim,
                         (long)NP * p->ncolR, p->ncolR, isn, st, N == NP));
  if (!p->dense_noise) {
    hipLaunchKernelGGL(k_circ, dim3(4, nbl), dim3(256), 0, st, p->Rre, p->Rim, p->Cre, p->Cim, N, NP,
                       p->ncolR, TP + MP);
    HPX_HIP(hipGetLastError());
  } else {
    HPX_HIP(hipMemsetAsync(p->Cre, 0, (size_t)nbl * N * sizeof(double), st));
    HPX_HIP(hipMemsetAsync(p->Cim, 0, (size_t)nbl * N * sizeof(double), st));
  }
  if (M > 0) {
    hipLaunchKernelGGL(k_small, dim3(nbl), dim3(256), 0, st, fgp, p->fg_shared, p->Zre, p->Zim,
                       p->Hre, p->Him, p->P4re, p->P4im, N, M, NP, TP, MP, p->ncolR);
    HPX_HIP(hipGetLastError());
  }
  if (omega) {   // P2 = U^H omega_a (shared by all baselines): use G scratch of baseline 0
    hipLaunchKernelGGL(k_prep_omega, dim3(64), dim3(256), 0, st, omega, p->Gre, p->Gim, T, N, NP, TP);
    HPX_HIP(hipGetLastError());
    HPX_TRY(hpx_launch_dft(1, NP, TP, p->Fopre, p->Fopim, 0, p->Gre, p->Gim, 0, TP, nullptr, 0,
                           p->P2re, p->P2im, 0, TP, isn, st, N == NP));
  } else {
    HPX_HIP(hipMemsetAsync(p->P2re, 0, (size_t)NP * TP * sizeof(double), st));
    HPX_HIP(hipMemsetAsync(p->P2im, 0, (size_t)NP * TP * sizeof(double), st));
  }
  // edge tiles for the factor (circulant mode): invariant rows >= rmin of the columns < rmin, P2 by row tile
  p->have_edge = 0;
  {
    const int rmin = 32 * (N / 32);
    if (!p->dense_noise && rmin > 0) {
      p->have_static = 1;                  // (gen_of reads the plan as it stands)
      hpx_gen_batch B = gen_of(p);
      B.rmin = rmin;
      B.e_bstride = (long)((p->ld - rmin) / 16) * rmin * 32;
      hipLaunchKernelGGL(k_build_edge, dim3(32, nbl), dim3(256), 0, st, B, p->E, p->npad, p->ld);
      hipLaunchKernelGGL(k_p2_tiles, dim3(64), dim3(256), 0, st, p->P2re, p->P2im, p->P2Tre, p->P2Tim, NP, TP);
      HPX_HIP(hipGetLastError());
      p->have_edge = 1;
    }
  }
  HPX_HIP(hipStreamSynchronize(st));
  p->have_static = 1;
  if (p->dense_noise) p->solver = HPX_SOLVER_DENSE;
  return HPX_OK;
}

extern "C" int hpx_plan_set_static(hpx_plan* p, const double* vis, const uint8_t* flags,
                                   const double* ninv, const double* fgmodes, int fg_shared,
                                   const int32_t* prior_map, const double* xgrid, int nxrows,
                                   int prior_shared, int ngrid, const double* omega,
                                   const double* fop, int any_flags, void* stream) {
  HPX_REQUIRE(ninv, "hpx_plan_set_static: null argument");
  return set_static_impl(p, vis, flags, ninv, nullptr, nullptr, 0, fgmodes, fg_shared, prior_map, xgrid, nxrows,
                         prior_shared, ngrid, omega, fop, any_flags, stream);
}

extern "C" int hpx_plan_set_static_dense(hpx_plan* p, const double* vis, const uint8_t* flags,
                                         const double* ninv_dense, const double* nih_dense, int noise_shared,
                                         const double* fgmodes, int fg_shared, const int32_t* prior_map,
                                         const double* xgrid, int nxrows, int prior_shared, int ngrid,
                                         const double* omega, const double* fop, int any_flags, void* stream) {
  HPX_REQUIRE(ninv_dense && nih_dense, "hpx_plan_set_static_dense: null noise matrices");
  return set_static_impl(p, vis, flags, nullptr, ninv_dense, nih_dense, noise_shared, fgmodes, fg_shared, prior_map,
                         xgrid, nxrows, prior_shared, ngrid, omega, fop, any_flags, stream);
}

extern "C" int hpx_plan_set_static_dense_flagged(hpx_plan* p, const double* vis, const uint8_t* flags,
                                                 const double* ninv_dense, int noise_shared,
                                                 const double* nih_masked, const double* fgmodes, int fg_shared,
                                                 const int32_t* prior_map, const double* xgrid, int nxrows,
                                                 int prior_shared, int ngrid, const double* omega, const double* fop,
                                                 void* stream) {
  HPX_REQUIRE(ninv_dense && nih_masked, "hpx_plan_set_static_dense_flagged: null noise matrices");
  return set_static_impl(p, vis, flags, nullptr, ninv_dense, nih_masked, noise_shared, fgmodes, fg_shared, prior_map,
                         xgrid, nxrows, prior_shared, ngrid, omega, fop, 1, stream, 1);
}

extern "C" int hpx_plan_set_rng(hpx_plan* p, const double* uniforms, const double* igy, int niter,
                                void* stream) {
  HPX_REQUIRE(p && uniforms && igy && niter > 0, "hpx_plan_set_rng: bad argument");
  hipStream_t st = (hipStream_t)stream;
  const size_t cnt = (size_t)niter * p->N;
  if (niter != p->niter_tab || !p->uni || !p->igy) {   // same length: the tables are refreshed in place
    HPX_TRY(dev_alloc(p, &p->uni, cnt));
    HPX_TRY(dev_alloc(p, &p->igy, cnt));
  }
  // on the caller's stream, and complete on return: the caller may release its tensors at once
  HPX_HIP(hipMemcpyAsync(p->uni, uniforms, cnt * sizeof(double), hipMemcpyDeviceToDevice, st));
  HPX_HIP(hipMemcpyAsync(p->igy, igy, cnt * sizeof(double), hipMemcpyDeviceToDevice, st));
  HPX_HIP(hipStreamSynchronize(st));
  p->niter_tab = niter;
  return HPX_OK;
}

// ---- time-dependent flags / noise (reference docstrings pspec.py:337-340, :398-401,
// FIXMEs :361, :450-451; run-hydra-pspec.py:524-541 reduces them to an any-time mask instead) -------
namespace {
// diag of (nu, N, N) c128 matrices -> (nu, N) f64
__global__ void k_pt_diag(const double* __restrict__ m, double* __restrict__ dg, const int N) {
  const int u = blockIdx.y;
  for (int k = blockIdx.x * blockDim.x + threadIdx.x; k < N; k += gridDim.x * blockDim.x)
    dg[(long)u * N + k] = m[(((long)u * N + k) * N + k) * 2];
}
// lnpart[b][0] = sum_t r_t^H Ninv_{b,t} r_t with the masked residual r [b][NP][TP] and the units' planar Ninv
// [b*T + t][NP][NP] (Hermitian, row-major): one workgroup per baseline, times in order (deterministic)
// sum_t (w r_t)^H Ninv_t (w r_t) with each time's own matrix: one workgroup per (time, baseline) -- it had been one
// per baseline, every thread walking its own matrix row (a stride of NP doubles between neighbouring threads) -- leaves
// the time's term in part[b][t]; the matrices are Hermitian, so row x is read as the conjugate of column x, which
// neighbouring threads read from neighbouring addresses.  k_quadform_pt_sum adds the terms in time order.
__global__ __launch_bounds__(256) void k_quadform_pt(const double* __restrict__ rre, const double* __restrict__ rim,
                                                     const double* __restrict__ nre, const double* __restrict__ nim,
                                                     double* __restrict__ part, const int N, const int T,
                                                     const int NP, const int TP) {
  __shared__ double red[4];
  const int t = blockIdx.x, b = blockIdx.y;
  const double* mr = nre + ((long)b * T + t) * NP * NP;
  const double* mi = nim + ((long)b * T + t) * NP * NP;
  double acc = 0.0;
  for (int x = threadIdx.x; x < N; x += 256) {
    double vr = 0.0, vi = 0.0;                         // v = (Ninv r)[x] = sum_k conj(Ninv[k][x]) r[k]
    for (int k = 0; k < N; ++k) {
      const double ar = mr[(long)k * NP + x], ai = -mi[(long)k * NP + x];
      const double br = rre[((long)b * NP + k) * TP + t], bi = rim[((long)b * NP + k) * TP + t];
      vr += ar * br - ai * bi;
      vi += ar * bi + ai * br;
    }
    acc += rre[((long)b * NP + x) * TP + t] * vr + rim[((long)b * NP + x) * TP + t] * vi;
  }
  const double tot = block_sum(acc, red);
  if (threadIdx.x == 0) part[(long)b * T + t] = tot;
}
__global__ void k_quadform_pt_sum(const double* __restrict__ part, double* __restrict__ lnpart, const int T, const int nbl) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= nbl) return;
  double tot = 0.0;
  for (int t = 0; t < T; ++t) tot += part[(long)b * T + t];
  lnpart[(long)b * HPX_NPART] = tot;
}
// flags_any[b][x] = AND_t flags_t[b][t][x];  ninv_any[b][x] = ninv_t[b][0][x]
__global__ void k_pt_reduce(const uint8_t* __restrict__ ft, const double* __restrict__ nt,
                            uint8_t* __restrict__ fany, double* __restrict__ nany, const int T, const int N) {
  const int b = blockIdx.y;
  for (int x = blockIdx.x * blockDim.x + threadIdx.x; x < N; x += gridDim.x * blockDim.x) {
    uint8_t a = 1;
    for (int t = 0; t < T; ++t) a &= (ft[((long)b * T + t) * N + x] ? 1 : 0);
    fany[(long)b * N + x] = a;
    nany[(long)b * N + x] = nt[(long)b * T * N + x];
  }
}
// D[b][x][t] = w_bt[x] vis[b][t][x]  (the masked data of pspec.py:613, per time)
__global__ void k_pt_data(const double* __restrict__ vis, const uint8_t* __restrict__ ft,
                          double* __restrict__ Dre, double* __restrict__ Dim, const int T, const int N,
                          const int NP, const int TP) {
  const int b = blockIdx.y;
  const long tot = (long)N * T;
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < tot; e += (long)gridDim.x * blockDim.x) {
    const int t = (int)(e / N), x = (int)(e % N);
    const long o = ((long)b * T + t) * N + x;
    const double w = ft[o] ? 1.0 : 0.0;
    Dre[((long)b * NP + x) * TP + t] = w * vis[2 * o];
    Dim[((long)b * NP + x) * TP + t] = w * vis[2 * o + 1];
  }
}
// child's omega_a block: PT[t][x][0] = P2[x][t], other columns zero
__global__ void k_pt_p2(const double* __restrict__ p2re, const double* __restrict__ p2im,
                        double* __restrict__ ptre, double* __restrict__ ptim, const int T, const int NP,
                        const int TP, const int TPc) {
  const long tot = (long)T * NP * TPc;
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < tot; e += (long)gridDim.x * blockDim.x) {
    const int c = (int)(e % TPc), x = (int)((e / TPc) % NP), t = (int)(e / ((long)TPc * NP));
    ptre[e] = (c == 0) ? p2re[(long)x * TP + t] : 0.0;
    ptim[e] = (c == 0) ? p2im[(long)x * TP + t] : 0.0;
  }
}
// child's P2 by row tile: PTT[t][c][0] = P2[c][t]  (rows 1..15 of the unit's only RHS tile stay zero)
__global__ void k_pt_p2t(const double* __restrict__ p2re, const double* __restrict__ p2im,
                         double* __restrict__ tre, double* __restrict__ tim, const int T, const int NP,
                         const int TP) {
  const long tot = (long)T * NP;
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < tot; e += (long)gridDim.x * blockDim.x) {
    const int t = (int)(e / NP), c = (int)(e % NP);
    tre[e << 4] = p2re[(long)c * TP + t];
    tim[e << 4] = p2im[(long)c * TP + t];
  }
}
// fg[u] = fg[u / T]  ((nbl,N,M) c128 -> (nbl*T,N,M))
__global__ void k_pt_expand_fg(const double* __restrict__ src, double* __restrict__ dst, const int T, const long per) {
  const int u = blockIdx.y;
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < per; e += (long)gridDim.x * blockDim.x)
    dst[(long)u * per + e] = src[(long)(u / T) * per + e];
}
// parent X[b][row][t] = child X[b*T + t][row][0]
__global__ void k_pt_gather(const double* __restrict__ cre, const double* __restrict__ cim,
                            double* __restrict__ xre, double* __restrict__ xim, const int T, const int npad,
                            const int TP, const int TPc) {
  const int b = blockIdx.y;
  const long tot = (long)npad * T;
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < tot; e += (long)gridDim.x * blockDim.x) {
    const int row = (int)(e / T), t = (int)(e % T);
    const long oc = ((long)(b * T + t) * npad + row) * TPc;
    xre[((long)b * npad + row) * TP + t] = cre[oc];
    xim[((long)b * npad + row) * TP + t] = cim[oc];
  }
}
}  // namespace

static int set_static_impl(hpx_plan* p, const double* vis, const uint8_t* flags,
                           const double* ninv, const double* ninv_dense, const double* nih_dense,
                           int noise_shared, const double* fgmodes, int fg_shared,
                           const int32_t* prior_map, const double* xgrid, int nxrows,
                           int prior_shared, int ngrid, const double* omega,
                           const double* fop, int any_flags, void* stream, int wb);

// `ninv_td` / `nih_td` non-NULL: full noise matrices per (baseline, time) -- the units of the child are then
// dense-noise systems (with the Woodbury correction when a unit has flagged channels), and `ninv_t` is ignored
// (the diagonals of ninv_td take its place for chi^2 and the parent's surrogate)
static int pertime_impl(hpx_plan* p, const double* vis, const uint8_t* flags_t, const double* ninv_t,
                        const double* ninv_td, const double* nih_td, const double* fgmodes, int fg_shared,
                        const int32_t* prior_map, const double* xgrid, int nxrows, int prior_shared, int ngrid,
                        const double* omega, const double* fop, int any_flags, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  const int nbl = p->nbl, T = p->T, N = p->N, M = p->M, NP = p->NP, TP = p->TP;
  hpx_devbuf dgb;
  int wb = 0, fmax = 0;
  if (ninv_td) {
    HPX_TRY(dgb.alloc((size_t)nbl * T * N));
    hipLaunchKernelGGL(k_pt_diag, dim3(4, nbl * T), dim3(256), 0, st, ninv_td, dgb.p, N);
    HPX_HIP(hipGetLastError());
    ninv_t = dgb.p;
    std::vector<uint8_t> hf((size_t)nbl * T * N);
    HPX_HIP(hipMemcpyAsync(hf.data(), flags_t, hf.size(), hipMemcpyDeviceToHost, st));
    HPX_HIP(hipStreamSynchronize(st));
    for (size_t u = 0; u < (size_t)nbl * T; ++u) {
      int f = 0;
      for (int j = 0; j < N; ++j) f += hf[u * N + j] ? 0 : 1;
      fmax = std::max(fmax, f);
    }
    wb = fmax > 0;
  }
  // 1. the parent's time-independent parts (foreground planes, operator, prior tables, omega_a block)
  //    with the any-time mask -- its own solve operators are never used in this mode
  hpx_devbuf tmp;
  HPX_TRY(tmp.alloc((size_t)nbl * N + ((size_t)nbl * N + 7) / 8 + 8));
  double* nany = tmp.p;
  uint8_t* fany = (uint8_t*)(tmp.p + (size_t)nbl * N);
  hipLaunchKernelGGL(k_pt_reduce, dim3(4, nbl), dim3(256), 0, st, flags_t, ninv_t, fany, nany, T, N);
  HPX_HIP(hipGetLastError());
  HPX_TRY(set_static_impl(p, vis, fany, nany, nullptr, nullptr, 0, fgmodes, fg_shared, prior_map, xgrid, nxrows,
                          prior_shared, ngrid, omega, fop, any_flags, stream));
  // 2. per-time data, flags, noise
  HPX_TRY(dev_alloc(p, &p->flags_t, (size_t)nbl * T * N));
  HPX_TRY(dev_alloc(p, &p->ninv_t, (size_t)nbl * T * N));
  HPX_HIP(hipMemcpyAsync(p->flags_t, flags_t, (size_t)nbl * T * N, hipMemcpyDeviceToDevice, st));
  HPX_HIP(hipMemcpyAsync(p->ninv_t, ninv_t, (size_t)nbl * T * N * sizeof(double), hipMemcpyDeviceToDevice, st));
  hipLaunchKernelGGL(k_pt_data, dim3(64, nbl), dim3(256), 0, st, vis, flags_t, p->Dre, p->Dim, T, N, NP, TP);
  HPX_HIP(hipGetLastError());
  // 3. the child: nbl*T units of one time sample each
  if (p->child) { hpx_plan_destroy(p->child); p->child = nullptr; }
  HPX_TRY(plan_create_impl(&p->child, nbl * T, 1, N, M, wb ? fmax : 0));
  hpx_plan* c = p->child;
  c->omega_mod = T;
  hpx_devbuf fgx;
  const double* fgc = fgmodes;
  if (M > 0 && !fg_shared) {
    const long per = (long)N * M * 2;
    HPX_TRY(fgx.alloc((size_t)nbl * T * per));
    hipLaunchKernelGGL(k_pt_expand_fg, dim3(16, nbl * T), dim3(256), 0, st, fgmodes, fgx.p, T, per);
    HPX_HIP(hipGetLastError());
    fgc = fgx.p;
  }
  // (vis (nbl,T,N) is (nbl*T,1,N); flags_t / ninv_t (nbl,T,N) are (nbl*T,N); the child never draws: no priors)
  if (ninv_td) {
    HPX_TRY(set_static_impl(c, vis, flags_t, nullptr, ninv_td, nih_td, 0, fgc, fg_shared, p->pmap, nullptr, 0, 1,
                            ngrid, omega, fop, wb, stream, wb));
    HPX_TRY(dev_alloc(p, &p->RDre, (size_t)nbl * NP * TP));     // masked residual for the quadratic form
    HPX_TRY(dev_alloc(p, &p->RDim, (size_t)nbl * NP * TP));
  } else
  HPX_TRY(set_static_impl(c, vis, flags_t, ninv_t, nullptr, nullptr, 0, fgc, fg_shared, p->pmap, nullptr, 0, 1,
                          ngrid, omega, fop, any_flags, stream));
  HPX_TRY(dev_alloc(p, &p->PTre, (size_t)T * NP * c->TP));
  HPX_TRY(dev_alloc(p, &p->PTim, (size_t)T * NP * c->TP));
  hipLaunchKernelGGL(k_pt_p2, dim3(64), dim3(256), 0, st, p->P2re, p->P2im, p->PTre, p->PTim, T, NP, TP, c->TP);
  HPX_HIP(hipGetLastError());
  // ... and by row tile for the factor's edge tiles: unit u = (b, t) has one right-hand-side row, time t
  HPX_TRY(dev_alloc(p, &p->PTTre, (size_t)T * NP * 16));
  HPX_TRY(dev_alloc(p, &p->PTTim, (size_t)T * NP * 16));
  HPX_HIP(hipMemsetAsync(p->PTTre, 0, (size_t)T * NP * 16 * sizeof(double), st));
  HPX_HIP(hipMemsetAsync(p->PTTim, 0, (size_t)T * NP * 16 * sizeof(double), st));
  hipLaunchKernelGGL(k_pt_p2t, dim3(64), dim3(256), 0, st, p->P2re, p->P2im, p->PTTre, p->PTTim, T, NP, TP);
  HPX_HIP(hipGetLastError());
  HPX_HIP(hipStreamSynchronize(st));
  p->per_time = ninv_td ? 2 : 1;
  p->solver = HPX_SOLVER_DENSE;
  return HPX_OK;
}

extern "C" int hpx_plan_set_static_pertime(hpx_plan* p, const double* vis, const uint8_t* flags_t,
                                           const double* ninv_t, const double* fgmodes, int fg_shared,
                                           const int32_t* prior_map, const double* xgrid, int nxrows,
                                           int prior_shared, int ngrid, const double* omega,
                                           const double* fop, int any_flags, void* stream) {
  HPX_REQUIRE(p && vis && flags_t && ninv_t && fop && prior_map, "hpx_plan_set_static_pertime: null argument");
  return pertime_impl(p, vis, flags_t, ninv_t, nullptr, nullptr, fgmodes, fg_shared, prior_map, xgrid, nxrows,
                      prior_shared, ngrid, omega, fop, any_flags, stream);
}

extern "C" int hpx_plan_set_static_pertime_dense(hpx_plan* p, const double* vis, const uint8_t* flags_t,
                                                 const double* ninv_t_dense, const double* nih_t,
                                                 const double* fgmodes, int fg_shared, const int32_t* prior_map,
                                                 const double* xgrid, int nxrows, int prior_shared, int ngrid,
                                                 const double* omega, const double* fop, int any_flags,
                                                 void* stream) {
  HPX_REQUIRE(p && vis && flags_t && ninv_t_dense && nih_t && fop && prior_map,
              "hpx_plan_set_static_pertime_dense: null argument");
  return pertime_impl(p, vis, flags_t, nullptr, ninv_t_dense, nih_t, fgmodes, fg_shared, prior_map, xgrid, nxrows,
                      prior_shared, ngrid, omega, fop, any_flags, stream);
}

// the child's generator: 1/a of the unit's baseline, omega_a of the unit's time
static hpx_gen_batch gen_of(const hpx_plan* p);
static hpx_gen_batch gen_of_child(const hpx_plan* p) {
  hpx_gen_batch B = gen_of(p->child);
  B.ia = p->ia;
  B.ia_div = p->T;
  B.p2re = p->PTre;
  B.p2im = p->PTim;
  B.p2_mod = p->T;
  B.p2_stride = (long)p->NP * p->child->TP;
  B.p2tre = p->PTTre;
  B.p2tim = p->PTTim;
  B.p2t_stride = (long)p->NP * 16;
  return B;
}

static hpx_gen_batch gen_of(const hpx_plan* p) {
  hpx_gen_batch B;
  B.ia = p->ia; B.cre = p->Cre; B.cim = p->Cim; B.rre = p->Rre; B.rim = p->Rim;
  B.p2re = p->P2re; B.p2im = p->P2im; B.hre = p->Hre; B.him = p->Him;
  B.p4re = p->P4re; B.p4im = p->P4im;
  B.cdre = p->dense_noise ? p->CDre : nullptr;
  B.cdim = p->dense_noise ? p->CDim : nullptr;
  B.ia_div = 1; B.p2_mod = 1; B.p2_stride = 0;
  {
    const long rmin = 32 * (long)(p->N / 32);
    B.ere = (p->have_edge && rmin > 0) ? p->E : nullptr;
    B.e_bstride = (long)((p->ld - rmin) / 16) * rmin * 32;
    B.p2tre = p->P2Tre; B.p2tim = p->P2Tim; B.p2t_stride = 0;
  }
  B.N = p->N; B.M = p->M; B.NP = p->NP; B.TP = p->TP; B.ncol = p->ncolR;
  B.has_omega = p->has_omega;
  B.rmin = 32 * (p->N / 32);
  return B;
}

static int launch_assemble_edge(hpx_plan* p, hipStream_t st) {
  const hpx_gen_batch B = gen_of(p);
  if (B.ere)      // the factor reads the edge tiles itself: only the last columns are laid out
    hipLaunchKernelGGL(k_assemble_tail, dim3(p->nbl), dim3(256), 0, st, B, p->L, p->npad, p->ld);
  else
    hipLaunchKernelGGL(k_assemble_edge, dim3(p->nbl, 1), dim3(256), 0, st, B, p->L, p->npad, p->ld);
  HPX_HIP(hipGetLastError());
  return HPX_OK;
}

static int launch_assemble(hpx_plan* p, hipStream_t st, int rlo) {
  hipLaunchKernelGGL(k_assemble, dim3(p->npad / 16, p->nbl), dim3(256), 0, st, gen_of(p), p->L,
                     p->npad, p->ld, rlo);
  HPX_HIP(hipGetLastError());
  return HPX_OK;
}

extern "C" int hpx_assemble_K(hpx_plan* p, const double* ps, double* k_out, void* stream) {
  HPX_REQUIRE(p && p->have_static && ps, "hpx_assemble_K: plan not initialised or null ps");
  HPX_REQUIRE(!p->per_time, "hpx_assemble_K: not available with time-dependent flags / noise");
  hipStream_t st = (hipStream_t)stream;
  const long tot = (long)p->nbl * p->N;
  hipLaunchKernelGGL(k_set_a, dim3(256), dim3(256), 0, st, ps, p->ia, p->ps_cur, tot, (double)p->N);
  HPX_HIP(hipGetLastError());
  HPX_TRY(launch_assemble(p, st, 0));
  if (k_out) {
    hipLaunchKernelGGL(k_kaug_out, dim3(128, p->nbl), dim3(256), 0, st, p->L, k_out, p->npad, p->ld);
    HPX_HIP(hipGetLastError());
  }
  HPX_HIP(hipStreamSynchronize(st));
  return HPX_OK;
}

extern "C" int hpx_plan_set_solver(hpx_plan* p, int mode) {
  HPX_REQUIRE(p && p->have_static, "hpx_plan_set_solver: plan has no static inputs");
  HPX_REQUIRE(mode == HPX_SOLVER_DENSE || mode == HPX_SOLVER_FLAT || mode == HPX_SOLVER_LOWRANK ||
              mode == HPX_SOLVER_LOWRANK_DIRECT, "hpx_plan_set_solver: unknown mode");
  HPX_REQUIRE(mode == HPX_SOLVER_DENSE || !p->dense_noise,
              "hpx_plan_set_solver: a dense inverse noise covariance needs the dense solver");
  HPX_REQUIRE(mode == HPX_SOLVER_DENSE || !p->per_time,
              "hpx_plan_set_solver: time-dependent flags / noise need the dense solver");
  if (mode == HPX_SOLVER_FLAT) {
    HPX_REQUIRE(!p->any_flags, "hpx_plan_set_solver: the flat-noise solver needs unflagged data");
    HPX_REQUIRE(p->M <= 16 && p->TP <= 256, "hpx_plan_set_solver: the flat-noise solver needs M <= 16, T <= 256");
    HPX_REQUIRE(hpx_flat_lds_bytes(p) <= 160 * 1024, "hpx_plan_set_solver: too many channels for the flat-noise solver");
    std::vector<double> ni((size_t)p->nbl * p->N);
    HPX_HIP(hipMemcpy(ni.data(), p->ni, ni.size() * sizeof(double), hipMemcpyDeviceToHost));
    for (int b = 0; b < p->nbl; ++b) {
      if (!(ni[(size_t)b * p->N] > 0.0)) {
        hpx_set_error("hpx_plan_set_solver: inverse noise variance of baseline %d is not positive", b);
        return HPX_EINVAL;
      }
      for (int k = 1; k < p->N; ++k)
        if (ni[(size_t)b * p->N + k] != ni[(size_t)b * p->N]) {
          hpx_set_error("hpx_plan_set_solver: inverse noise variance of baseline %d is not flat (channel %d)",
                        b, k);
          return HPX_EINVAL;
        }
    }
  }
  if (mode == HPX_SOLVER_LOWRANK || mode == HPX_SOLVER_LOWRANK_DIRECT) {
    HPX_REQUIRE(p->TP <= 256, "hpx_plan_set_solver: the low-rank solver needs T <= 256");
    const int nbl = p->nbl, N = p->N;
    std::vector<double> ni((size_t)nbl * N);
    std::vector<uint8_t> fl((size_t)nbl * N);
    HPX_HIP(hipMemcpy(ni.data(), p->ni, ni.size() * sizeof(double), hipMemcpyDeviceToHost));
    HPX_HIP(hipMemcpy(fl.data(), p->flags, fl.size(), hipMemcpyDeviceToHost));
    std::vector<int32_t> cnt(nbl, 0);
    std::vector<double> cv(nbl, 0.0);
    int fmax = 0;
    for (int b = 0; b < nbl; ++b) {
      bool have = false;
      for (int k = 0; k < N; ++k) {
        if (!fl[(size_t)b * N + k]) { ++cnt[b]; continue; }
        const double v = ni[(size_t)b * N + k];
        if (!have) { cv[b] = v; have = true; }
        else if (v != cv[b]) {
          hpx_set_error("hpx_plan_set_solver: inverse noise variance of baseline %d is not flat over its "
                        "unflagged channels (channel %d)", b, k);
          return HPX_EINVAL;
        }
      }
      if (!have || !(cv[b] > 0.0)) {
        hpx_set_error("hpx_plan_set_solver: baseline %d has no usable channel", b);
        return HPX_EINVAL;
      }
      fmax = cnt[b] > fmax ? cnt[b] : fmax;
    }
    HPX_REQUIRE(p->M + fmax <= 240, "hpx_plan_set_solver: too many flagged channels for the low-rank solver (M + f <= 240)");
    if (fmax < 1) fmax = 1;
    std::vector<int32_t> list((size_t)nbl * fmax, 0);
    for (int b = 0; b < nbl; ++b) {
      int j = 0;
      for (int k = 0; k < N; ++k)
        if (!fl[(size_t)b * N + k]) list[(size_t)b * fmax + j++] = k;
    }
    // decide the form and check its LDS need BEFORE anything is allocated on the plan
    const int use_fft = (mode == HPX_SOLVER_LOWRANK && hpx_dft_use_fft && p->N == p->NP && (N & (N - 1)) == 0 &&
                         N >= 32 && N <= 4096 && p->M <= 16 && hpx_flat_lds_bytes(p) <= 160 * 1024) ? 1 : 0;
    {
      const int old_fmax = p->lr_fmax, old_npad = p->lr_npad;
      p->lr_fmax = fmax;
      p->lr_npad = ceil16(p->M + fmax);
      if (!use_fft && hpx_lowrank_lds_bytes(p) > 160 * 1024) {
        p->lr_fmax = old_fmax;
        p->lr_npad = old_npad;
        hpx_set_error("hpx_plan_set_solver: Ntimes / flag count too large for the low-rank solver");
        return HPX_EINVAL;
      }
    }
    const size_t nb = nbl, ns = p->lr_npad, lds_ = ns + p->TP, nblkS = (ns + HPX_NB - 1) / HPX_NB;
    HPX_TRY(dev_alloc(p, &p->lr_flist, nb * fmax));
    HPX_TRY(dev_alloc(p, &p->lr_fcount, nb));
    HPX_TRY(dev_alloc(p, &p->lr_c, nb));
    HPX_TRY(dev_alloc(p, &p->lr_L, nb * ns * lds_ * 2));
    HPX_TRY(dev_alloc(p, &p->lr_Wre, nb * nblkS * 1024));
    HPX_TRY(dev_alloc(p, &p->lr_Wim, nb * nblkS * 1024));
    HPX_TRY(dev_alloc(p, &p->lr_Vt, nb * HPX_VT_STRIDE(ns)));
    HPX_HIP(hipMemset(p->lr_Vt, 0, nb * HPX_VT_STRIDE(ns) * sizeof(double)));
    HPX_TRY(dev_alloc(p, &p->lr_Yre, nb * ns * p->TP));
    HPX_TRY(dev_alloc(p, &p->lr_Yim, nb * ns * p->TP));
    HPX_HIP(hipMemcpy(p->lr_flist, list.data(), list.size() * sizeof(int32_t), hipMemcpyHostToDevice));
    HPX_HIP(hipMemcpy(p->lr_fcount, cnt.data(), cnt.size() * sizeof(int32_t), hipMemcpyHostToDevice));
    HPX_HIP(hipMemcpy(p->lr_c, cv.data(), cv.size() * sizeof(double), hipMemcpyHostToDevice));
    HPX_HIP(hipMemset(p->lr_L, 0, nb * ns * lds_ * 2 * sizeof(double)));
    HPX_HIP(hipMemset(p->lr_Yre, 0, nb * ns * p->TP * sizeof(double)));
    HPX_HIP(hipMemset(p->lr_Yim, 0, nb * ns * p->TP * sizeof(double)));
    // FFT form when the channel count has an FFT and the foreground block fits one MFMA tile
    p->lr_fft = use_fft;
    if (p->lr_fft) {
      p->lr_cp = ceil16(1 + p->M);
      const size_t xw = (size_t)p->lr_cp + p->TP;
      std::vector<int32_t> finv((size_t)nbl * N, -1);
      for (int b = 0; b < nbl; ++b)
        for (int j = 0; j < cnt[b]; ++j) finv[(size_t)b * N + list[(size_t)b * fmax + j]] = j;
      HPX_TRY(dev_alloc(p, &p->lr_finv, nb * N));
      HPX_HIP(hipMemcpy(p->lr_finv, finv.data(), finv.size() * sizeof(int32_t), hipMemcpyHostToDevice));
      HPX_TRY(dev_alloc(p, &p->lr_Ire, nb * p->NP * xw));
      HPX_TRY(dev_alloc(p, &p->lr_Iim, nb * p->NP * xw));
      // columns 1 + M .. lr_cp - 1 of the transform input are never written: zero them once
      HPX_HIP(hipMemset(p->lr_Ire, 0, nb * p->NP * xw * sizeof(double)));
      HPX_HIP(hipMemset(p->lr_Iim, 0, nb * p->NP * xw * sizeof(double)));
      HPX_TRY(dev_alloc(p, &p->lr_Ore, nb * p->NP * xw));
      HPX_TRY(dev_alloc(p, &p->lr_Oim, nb * p->NP * xw));
      HPX_TRY(dev_alloc(p, &p->lr_Sre, nb * 16 * (16 + p->TP)));
      HPX_TRY(dev_alloc(p, &p->lr_Sim, nb * 16 * (16 + p->TP)));
    } else {
      HPX_TRY(dev_alloc(p, &p->lr_Bre, nb * p->NP * (ns + p->TP)));     // [Bd | r1]
      HPX_TRY(dev_alloc(p, &p->lr_Bim, nb * p->NP * (ns + p->TP)));
      HPX_TRY(dev_alloc(p, &p->lr_Tre, nb * p->NP * ns));
      HPX_TRY(dev_alloc(p, &p->lr_Tim, nb * p->NP * ns));
    }
    HPX_TRY(hpx_lowrank_prepare(p, 0));
  }
  p->solver = (mode == HPX_SOLVER_LOWRANK_DIRECT) ? HPX_SOLVER_LOWRANK : mode;
  return HPX_OK;
}

// Options: of one plan (p != NULL) or of the library (p == NULL); see include/hpx.h
extern "C" int hpx_set_option(hpx_plan* p, int key, int value) {
  if (p) {
    if (key == HPX_OPT_FACTOR_SPLIT) {
      p->allow_split = value != 0;
      if (p->child) p->child->allow_split = p->allow_split;
      return HPX_OK;
    }
    hpx_set_error("hpx_set_option: key %d is not a plan option", key);
    return HPX_EINVAL;
  }
  int rc = HPX_EINVAL;
  if (key == HPX_OPT_FACTOR_SPLIT || key == HPX_OPT_SPLIT_HEAVY || key == HPX_OPT_SPLIT_SPIN_LIMIT)
    rc = hpx_split_set_option(key, value);
  else if (key == HPX_OPT_EIGH_INNER_SWEEPS || key == HPX_OPT_EIGH_TRACE) rc = hpx_eigh_set_option(key, value);
  if (rc != HPX_OK) hpx_set_error("hpx_set_option: unknown key %d or bad value %d", key, value);
  return rc;
}

extern "C" int hpx_plan_set_profiling(hpx_plan* p, int on) {
  HPX_REQUIRE(p, "null plan");
  p->profiling = on ? 1 : 0;
  return HPX_OK;
}

extern "C" int hpx_plan_stage_ms(hpx_plan* p, float* ms_host) {
  HPX_REQUIRE(p && ms_host, "null argument");
  for (int i = 0; i < HPX_NSTAGE; ++i) ms_host[i] = p->stage_ms[i];
  return HPX_OK;
}

extern "C" int hpx_plan_info(hpx_plan* p, int32_t* info_host) {
  HPX_REQUIRE(p && info_host, "null argument");
  HPX_HIP(hipMemcpy(info_host, p->info, (size_t)p->nbl * sizeof(int32_t), hipMemcpyDeviceToHost));
  return HPX_OK;
}

static int mark(hpx_plan* p, hipStream_t st) {
  if (!p->profiling) return HPX_OK;
  if (p->ev_used == (int)p->events.size()) {
    hipEvent_t ev;
    HPX_HIP(hipEventCreate(&ev));
    p->events.push_back(ev);
  }
  HPX_HIP(hipEventRecord(p->events[p->ev_used++], st));
  return HPX_OK;
}

// Everything after the solve of one iteration: back transform, residual / chi^2 / first
// ---- dense noise with flags: Woodbury correction of the unflagged-noise solution -----------------
// (hpx.h, hpx_plan_set_static_dense_flagged).  W[b] is the f x (f + T) system [I - Q^H Y_P | Q^H Y_r],
// row-major interleaved complex with leading dimension fmax + T, where (Q^H Y)[jf][col] is the model
// (U y + F f)[channel flist[jf]] of solution column col; Y_P are the columns T .. T+f-1 of X.
__global__ __launch_bounds__(256) void k_wb_system(const double* __restrict__ Sre, const double* __restrict__ Sim,
                                                   const double* __restrict__ Xre, const double* __restrict__ Xim,
                                                   const double* __restrict__ Fre, const double* __restrict__ Fim,
                                                   const int fg_shared, const int32_t* __restrict__ flist,
                                                   const int32_t* __restrict__ fcount, double* __restrict__ W_all,
                                                   const int fmax, const int N, const int M, const int T,
                                                   const int NP, const int TP, const int npad) {
  const int b = blockIdx.x, f = fcount[b], ldw = fmax + T;
  double* W = W_all + (long)b * fmax * ldw * 2;
  const double* sre = Sre + (long)b * NP * TP;
  const double* sim = Sim + (long)b * NP * TP;
  const double* xre = Xre + (long)b * npad * TP;
  const double* xim = Xim + (long)b * npad * TP;
  const double* fre = Fre + (fg_shared ? 0 : (long)b * N * M);
  const double* fim = Fim + (fg_shared ? 0 : (long)b * N * M);
  for (int e = threadIdx.x; e < f * (f + T); e += 256) {
    const int jf = e / (f + T), c = e % (f + T);
    const int col = (c < f) ? T + c : c - f;            // solution column: Y_P first, then Y_r
    const int j = flist[(long)b * fmax + jf];
    double mr = sre[(long)j * TP + col], mi = sim[(long)j * TP + col];
    for (int m = 0; m < M; ++m) {
      const double fr = fre[(long)j * M + m], fi = fim[(long)j * M + m];
      const double gr = xre[(long)(N + m) * TP + col], gi = xim[(long)(N + m) * TP + col];
      mr += gr * fr - gi * fi;
      mi += gr * fi + gi * fr;
    }
    double* w = W + ((long)jf * ldw + (c < f ? c : fmax + (c - f))) * 2;
    if (c < f) {
      w[0] = (jf == c ? 1.0 : 0.0) - mr;
      w[1] = -mi;
    } else {
      w[0] = mr;
      w[1] = mi;
    }
  }
}
// Gaussian elimination with partial pivoting on the f x (f + T) system of one baseline (global memory,
// one workgroup), then the back substitution: the coefficients c[kf][t] end up in the right-hand-side
// columns fmax .. fmax+T-1.  A vanishing pivot marks the baseline in info.
__global__ __launch_bounds__(256) void k_wb_solve(double* __restrict__ W_all, const int32_t* __restrict__ fcount,
                                                  const int fmax, const int T, int32_t* __restrict__ info,
                                                  const int iter_tag) {
  __shared__ double redv[4];
  __shared__ int redi[4], piv_s;
  __shared__ double lre[512], lim[512];
  const int b = blockIdx.x, f = fcount[b], ldw = fmax + T, tid = threadIdx.x;
  if (f == 0) return;
  double* W = W_all + (long)b * fmax * ldw * 2;
  const int ncol = fmax + T;                               // columns f .. fmax-1 are unused (never touched)
  for (int k = 0; k < f; ++k) {
    double best = -1.0;
    int at = k;
    for (int r = k + tid; r < f; r += 256) {
      const double a2 = W[((long)r * ldw + k) * 2] * W[((long)r * ldw + k) * 2] +
                        W[((long)r * ldw + k) * 2 + 1] * W[((long)r * ldw + k) * 2 + 1];
      if (a2 > best) { best = a2; at = r; }
    }
    for (int o = 32; o > 0; o >>= 1) {
      const double ob = __shfl_xor(best, o, 64);
      const int oa = __shfl_xor(at, o, 64);
      if (ob > best || (ob == best && oa < at)) { best = ob; at = oa; }
    }
    if ((tid & 63) == 0) { redv[tid >> 6] = best; redi[tid >> 6] = at; }
    __syncthreads();
    if (tid == 0) {
      int w = 0;
      for (int q = 1; q < 4; ++q)
        if (redv[q] > redv[w] || (redv[q] == redv[w] && redi[q] < redi[w])) w = q;
      piv_s = redi[w];
      // (the system is I - Q^H Y_P, entries O(1): a pivot below 1e-12 means the flags leave it singular, e.g. a unit
      // with every channel flagged, whose foreground amplitudes nothing constrains)
      if (!(redv[w] > 1e-24)) atomicCAS(&info[b], 0, iter_tag);
    }
    __syncthreads();
    const int pv = piv_s;
    if (pv != k)
      for (int c = k + tid; c < ncol; c += 256) {
        if (c >= f && c < fmax) continue;
        double* x = W + ((long)k * ldw + c) * 2;
        double* y = W + ((long)pv * ldw + c) * 2;
        const double t0 = x[0], t1 = x[1];
        x[0] = y[0]; x[1] = y[1];
        y[0] = t0; y[1] = t1;
      }
    __syncthreads();
    const double pr = W[((long)k * ldw + k) * 2], pi = W[((long)k * ldw + k) * 2 + 1];
    const double den = 1.0 / (pr * pr + pi * pi);
    for (int r = k + 1 + tid; r < f; r += 256) {           // multipliers l_r = W[r][k] / W[k][k]
      const double ar = W[((long)r * ldw + k) * 2], ai = W[((long)r * ldw + k) * 2 + 1];
      lre[r] = (ar * pr + ai * pi) * den;
      lim[r] = (ai * pr - ar * pi) * den;
    }
    __syncthreads();
    const int nc = (f - k - 1) + T, nr = f - k - 1;
    for (int e = tid; e < nr * nc; e += 256) {
      const int r = k + 1 + e / nc, ci = e % nc;
      const int c = (ci < f - k - 1) ? k + 1 + ci : fmax + (ci - (f - k - 1));
      const double ur = W[((long)k * ldw + c) * 2], ui = W[((long)k * ldw + c) * 2 + 1];
      double* x = W + ((long)r * ldw + c) * 2;
      x[0] -= lre[r] * ur - lim[r] * ui;
      x[1] -= lre[r] * ui + lim[r] * ur;
    }
    __syncthreads();
  }
  // back substitution, one thread per right-hand side
  for (int t = tid; t < T; t += 256) {
    for (int k = f - 1; k >= 0; --k) {
      double sr = W[((long)k * ldw + fmax + t) * 2], si = W[((long)k * ldw + fmax + t) * 2 + 1];
      for (int q = k + 1; q < f; ++q) {
        const double ur = W[((long)k * ldw + q) * 2], ui = W[((long)k * ldw + q) * 2 + 1];
        const double cr = W[((long)q * ldw + fmax + t) * 2], ci = W[((long)q * ldw + fmax + t) * 2 + 1];
        sr -= ur * cr - ui * ci;
        si -= ur * ci + ui * cr;
      }
      const double pr = W[((long)k * ldw + k) * 2], pi = W[((long)k * ldw + k) * 2 + 1];
      const double den = 1.0 / (pr * pr + pi * pi);
      W[((long)k * ldw + fmax + t) * 2] = (sr * pr + si * pi) * den;
      W[((long)k * ldw + fmax + t) * 2 + 1] = (si * pr - sr * pi) * den;
    }
  }
}
// X[:, t] += sum_kf X[:, T + kf] c[kf][t]  for the solution rows (npad) and the signal realisation S (N rows)
// The same solve with the whole system in LDS (f (f + T) complex entries: 134 KB at 77 flagged channels and 32
// times; the launch takes this form when it fits, k_wb_solve otherwise): LU with partial pivoting, right-looking, the
// right-hand sides swept along; the back substitution row-parallel (one barrier per unknown) instead of one thread
// per right-hand side -- with per-time units (T = 1) that was a single lane.  Same pivoting rule and operations as
// k_wb_solve.
__global__ __launch_bounds__(256) void k_wb_solve_lds(double* __restrict__ W_all, const int32_t* __restrict__ fcount,
                                                      const int fmax, const int T, int32_t* __restrict__ info,
                                                      const int iter_tag) {
  extern __shared__ double wl[];             // re [f][ldl] | im [f][ldl], ldl = f + T (+1 when even: bank spread)
  __shared__ double redv[4];
  __shared__ int redi[4], piv_s;
  const int b = blockIdx.x, f = fcount[b], ldw = fmax + T, tid = threadIdx.x;
  if (f == 0) return;
  double* W = W_all + (long)b * fmax * ldw * 2;
  const int nc = f + T, ldl = nc | 1;
  double* wr = wl;
  double* wi = wl + (size_t)f * ldl;
  // compact copy: columns 0 .. f-1 the matrix, f .. f+T-1 the right-hand sides (global columns fmax ..)
  for (int e = tid; e < f * nc; e += 256) {
    const int r = e / nc, c = e % nc;
    const int cg = (c < f) ? c : fmax + (c - f);
    wr[r * ldl + c] = W[((long)r * ldw + cg) * 2];
    wi[r * ldl + c] = W[((long)r * ldw + cg) * 2 + 1];
  }
  __syncthreads();
  for (int k = 0; k < f; ++k) {
    double best = -1.0;
    int at = k;
    for (int r = k + tid; r < f; r += 256) {
      const double a2 = wr[r * ldl + k] * wr[r * ldl + k] + wi[r * ldl + k] * wi[r * ldl + k];
      if (a2 > best) { best = a2; at = r; }
    }
    for (int o = 32; o > 0; o >>= 1) {
      const double ob = __shfl_xor(best, o, 64);
      const int oa = __shfl_xor(at, o, 64);
      if (ob > best || (ob == best && oa < at)) { best = ob; at = oa; }
    }
    if ((tid & 63) == 0) { redv[tid >> 6] = best; redi[tid >> 6] = at; }
    __syncthreads();
    if (tid == 0) {
      int w = 0;
      for (int q = 1; q < 4; ++q)
        if (redv[q] > redv[w] || (redv[q] == redv[w] && redi[q] < redi[w])) w = q;
      piv_s = redi[w];
      if (!(redv[w] > 1e-24)) atomicCAS(&info[b], 0, iter_tag);       // (see k_wb_solve)
    }
    __syncthreads();
    const int pv = piv_s;
    if (pv != k)
      for (int c = k + tid; c < nc; c += 256) {
        const double t0 = wr[k * ldl + c], t1 = wi[k * ldl + c];
        wr[k * ldl + c] = wr[pv * ldl + c]; wi[k * ldl + c] = wi[pv * ldl + c];
        wr[pv * ldl + c] = t0; wi[pv * ldl + c] = t1;
      }
    __syncthreads();
    const double pr = wr[k * ldl + k], pi = wi[k * ldl + k];
    const double den = 1.0 / (pr * pr + pi * pi);
    // rows below: the multiplier l_r = W[r][k] / W[k][k] is formed by every thread of the row for itself (the
    // column k entry is left alone until the step's barrier)
    const int ncu = nc - k - 1, nr = f - k - 1;
    for (int e = tid; e < nr * ncu; e += 256) {
      const int r = k + 1 + e / ncu, c = k + 1 + e % ncu;
      const double ar = wr[r * ldl + k], ai = wi[r * ldl + k];
      const double lre = (ar * pr + ai * pi) * den, lim = (ai * pr - ar * pi) * den;
      const double ur = wr[k * ldl + c], ui = wi[k * ldl + c];
      wr[r * ldl + c] -= lre * ur - lim * ui;
      wi[r * ldl + c] -= lre * ui + lim * ur;
    }
    __syncthreads();
  }
  // back substitution: unknown k of every right-hand side, then its column out of the rows above
  for (int k = f - 1; k >= 0; --k) {
    const double pr = wr[k * ldl + k], pi = wi[k * ldl + k];
    const double den = 1.0 / (pr * pr + pi * pi);
    for (int t = tid; t < T; t += 256) {
      const double sr = wr[k * ldl + f + t], si = wi[k * ldl + f + t];
      wr[k * ldl + f + t] = (sr * pr + si * pi) * den;
      wi[k * ldl + f + t] = (si * pr - sr * pi) * den;
    }
    __syncthreads();
    for (int e = tid; e < k * T; e += 256) {
      const int r = e / T, t = e % T;
      const double ur = wr[r * ldl + k], ui = wi[r * ldl + k];
      const double cr = wr[k * ldl + f + t], ci = wi[k * ldl + f + t];
      wr[r * ldl + f + t] -= ur * cr - ui * ci;
      wi[r * ldl + f + t] -= ur * ci + ui * cr;
    }
    __syncthreads();
  }
  for (int e = tid; e < f * T; e += 256) {
    const int r = e / T, t = e % T;
    W[((long)r * ldw + fmax + t) * 2] = wr[r * ldl + f + t];
    W[((long)r * ldw + fmax + t) * 2 + 1] = wi[r * ldl + f + t];
  }
}
// k_wb_solve in the form that fits: the system in LDS up to 150 KB
static int launch_wb_solve(int nbl, double* W, const int32_t* fcount, int fmax, int T, int32_t* info, int iter_tag,
                           hipStream_t st) {
  const size_t lds = (size_t)2 * fmax * ((fmax + T) | 1) * sizeof(double);
  if (lds <= (size_t)150 * 1024) {
    static hpx_lds_limit limit;
    HPX_TRY(limit.ensure(reinterpret_cast<const void*>(&k_wb_solve_lds), lds));
    hipLaunchKernelGGL(k_wb_solve_lds, dim3(nbl), dim3(256), lds, st, W, fcount, fmax, T, info, iter_tag);
  } else {
    hipLaunchKernelGGL(k_wb_solve, dim3(nbl), dim3(256), 0, st, W, fcount, fmax, T, info, iter_tag);
  }
  HPX_HIP(hipGetLastError());
  return HPX_OK;
}

__global__ __launch_bounds__(256) void k_wb_correct(double* __restrict__ Sre, double* __restrict__ Sim,
                                                    double* __restrict__ Xre, double* __restrict__ Xim,
                                                    const double* __restrict__ W_all,
                                                    const int32_t* __restrict__ fcount, const int fmax, const int T,
                                                    const int NP, const int TP, const int npad) {
  // out[r][t] += sum_k Y_P[r][k] c[k][t] over the rows of X and of S: a (rows x f) by (f x T) product per baseline, on
  // the matrix pipe (it had been a scalar loop per entry: 4.3 ms per iteration at the C3 shape with 77 flagged
  // channels).  One wave per 16-row tile; k beyond the baseline's own f contributes zeros on both sides (those
  // columns of X / S are never written).
  const int b = blockIdx.y, f = fcount[b], ldw = fmax + T;
  if (f == 0) return;
  const double* W = W_all + (long)b * fmax * ldw * 2;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, li = lane & 15, g = lane >> 4;
  const int ntile = (npad + NP) >> 4, nks = (f + 3) >> 2, ntt = (T + 15) >> 4;
  for (int rt = blockIdx.x * 4 + wave; rt < ntile; rt += gridDim.x * 4) {
    const int r0 = rt << 4;
    double* pr = (r0 < npad) ? Xre + ((long)b * npad + r0) * TP : Sre + ((long)b * NP + (r0 - npad)) * TP;
    double* pi = (r0 < npad) ? Xim + ((long)b * npad + r0) * TP : Sim + ((long)b * NP + (r0 - npad)) * TP;
    for (int tt = 0; tt < ntt; ++tt) {
      const int t = (tt << 4) + li;
      const bool tok = t < T;
      d4 ar = {0., 0., 0., 0.}, ai = ar;
      for (int ks = 0; ks < nks; ++ks) {
        const int k = 4 * ks + g;
        const bool kok = k < f;
        // A[m = li][k] = Y_P[r0 + li][k];  B[k][n = li] = c[k][t]
        const double yr = kok ? pr[(long)li * TP + T + k] : 0.0, yi = kok ? pi[(long)li * TP + T + k] : 0.0;
        const long wo = ((long)min(k, f - 1) * ldw + fmax + min(t, T - 1)) * 2;
        const double cr = (kok && tok) ? W[wo] : 0.0, ci = (kok && tok) ? W[wo + 1] : 0.0;
        ar = mfma64(yr, cr, ar);
        ar = mfma64(-yi, ci, ar);
        ai = mfma64(yr, ci, ai);
        ai = mfma64(yi, cr, ai);
      }
      if (tok) {
#pragma unroll
        for (int v = 0; v < 4; ++v) {               // accumulator: row g + 4 v, column li
          const long o = (long)HPX_ACC_ROW(g, v) * TP + t;
          pr[o] += ar[v];
          pi[o] += ai[v];
        }
      }
    }
  }
}

// ln-posterior term / beta, masked transform (flags), bandpower draw.  `rs` = row scaling of
// y' in the back transform (a = sqrt(ps/N), or NULL when X already holds s' = Sh' y').
struct IterOut {
  const double* ps_forced;   // already offset to this iteration, or NULL
  double *ps_out, *lnpost_out, *cr_out, *fg_out, *chisq_out;   // ps/lnpost offset to this iteration
  long ps_bstride, forced_bstride, lnpost_pitch;
  long cr_bstride, fg_bstride, chisq_bstride;
};

#ifndef HPX_DFT_RESID
#define HPX_DFT_RESID 1       // small N without an FFT: dense transform + residual in one kernel
#endif
#ifndef HPX_FUSE_TC
#define HPX_FUSE_TC 8      // fewest time columns per block for which the fused transform + residual kernel is used
#endif
static int post_solve(hpx_plan* p, int it_abs, const IterOut& O, hipStream_t st) {
  const int nbl = p->nbl, N = p->N, M = p->M, T = p->T, NP = p->NP, TP = p->TP;
  const int TPd = (T + 15) & ~15;          // the data's time columns, padded (TP may hold more right-hand sides)
  const double isn = 1.0 / sqrt((double)N);
  ResArgs R;
  R.Xre = p->Xre; R.Xim = p->Xim; R.Sre = p->Sre; R.Sim = p->Sim; R.Dre = p->Dre; R.Dim = p->Dim;
  R.Fre = p->Fre; R.Fim = p->Fim; R.ninv = p->ninv; R.flags = p->flags;
  R.bpart = p->bpart; R.lnpart = p->lnpart; R.Gre = p->Gre; R.Gim = p->Gim;
  R.cr_bstride = O.cr_bstride; R.fg_bstride = O.fg_bstride; R.chisq_bstride = O.chisq_bstride;
  R.cr_out = O.cr_out; R.fg_out = (M > 0) ? O.fg_out : nullptr; R.chisq_out = O.chisq_out;
  R.N = N; R.M = M; R.T = T; R.NP = NP; R.TP = TP; R.npad = p->npad;
  R.fg_shared = p->fg_shared; R.any_flags = p->any_flags;
  R.twre = p->Fopre; R.twim = p->Fopim; R.isn = isn; R.logN = 0; R.tcs = 0; R.nbl = nbl; R.npart = 1;
  // dense noise: the masked residual for the quadratic form; it can share G unless G holds w s (flags)
  R.Rdre = p->dense_noise ? (p->dense_noise == 2 ? p->RDre : p->Gre) : (p->per_time == 2 ? p->RDre : nullptr);
  R.Rdim = p->dense_noise ? (p->dense_noise == 2 ? p->RDim : p->Gim) : (p->per_time == 2 ? p->RDim : nullptr);
  R.flags_t = p->per_time ? p->flags_t : nullptr;
  R.ninv_t = p->per_time ? p->ninv_t : nullptr;
  // time columns per block of the fused kernel: 64 KiB of LDS for the signal, as k_fft
#ifndef HPX_FR_ELEMS
#define HPX_FR_ELEMS 4096      // complex elements of the signal block a workgroup of k_fft_resid holds in LDS
#endif
  int npart = 1, TC = HPX_FR_ELEMS / NP;
  if (TC > 16) TC = 16;
  const bool pow2 = N == NP && (N & (N - 1)) == 0 && N >= 32 && N <= 4096;
  const bool generic_post = p->dense_noise || p->per_time;      // modes only the two-kernel form implements
  if (pow2 && hpx_dft_use_fft && TC >= HPX_FUSE_TC && TP / TC <= HPX_NPART && !generic_post) {   // fewer columns per block: two kernels win
    while ((1 << R.logN) < N) ++R.logN;
    while ((1 << R.tcs) < TC) ++R.tcs;
    npart = TP / TC;
    // s = U z, residual, chi^2, |z|^2 sums in one pass (k_fft_resid); the two event marks
    // book it under "transform"
    const size_t lds = ((size_t)N * TC * 2 + N + (size_t)2 * M * TC + (size_t)2 * M * (256 / TC)) * sizeof(double);
    static hpx_lds_limit limit;
    HPX_TRY(limit.ensure(reinterpret_cast<const void*>(&k_fft_resid), lds));
    R.nbl = nbl; R.npart = npart;
    hipLaunchKernelGGL(k_fft_resid, dim3(((nbl + 7) / 8) * 8 * npart), dim3(256), lds, st, R);
    HPX_HIP(hipGetLastError());
    HPX_TRY(mark(p, st));
  } else if (HPX_DFT_RESID && NP <= 256 && M <= 16 && hpx_dft_use_fft && !generic_post) {
    // small N without an in-LDS FFT: dense transform fused with the residual (k_dft_resid), booked
    // under "transform"
    npart = (NP / 16 + 3) / 4;
    R.nbl = nbl; R.npart = npart;
    hipLaunchKernelGGL(k_dft_resid, dim3(npart, nbl), dim3(256), 0, st, R);
    HPX_HIP(hipGetLastError());
    HPX_TRY(mark(p, st));
  } else {
    // s = U z = conj(F) X / sqrt(N)   (rows >= N of X meet the zero padding of the operator)
    HPX_TRY(hpx_launch_dft(nbl, NP, TP, p->Fopre, p->Fopim, 1, p->Xre, p->Xim, (long)p->npad * TP,
                           TP, nullptr, 0, p->Sre, p->Sim, (long)NP * TP, TP, isn, st, N == NP));
    if (p->dense_noise == 2) {
      // dense noise with flags: X = [Y_r | Y_P] so far (unflagged-noise system); the Woodbury correction
      // x = Y_r + Y_P (I - Q^H Y_P)^-1 Q^H Y_r, where Q^H Y is the model U y + F f at the flagged channels
      const int fm = p->wb_fmax;
      hipLaunchKernelGGL(k_wb_system, dim3(nbl), dim3(256), 0, st, p->Sre, p->Sim, p->Xre, p->Xim, p->Fre, p->Fim,
                         p->fg_shared, p->wb_flist, p->wb_fcount, p->wb_W, fm, N, M, T, NP, TP, p->npad);
      HPX_TRY(launch_wb_solve(nbl, p->wb_W, p->wb_fcount, fm, T, p->info, it_abs + 1, st));
      hipLaunchKernelGGL(k_wb_correct, dim3(8, nbl), dim3(256), 0, st, p->Sre, p->Sim, p->Xre, p->Xim, p->wb_W,
                         p->wb_fcount, fm, T, NP, TP, p->npad);
      HPX_HIP(hipGetLastError());
    }
    HPX_TRY(mark(p, st));
    {
      // slices of the channels: as many as keep 64 channels per workgroup, at most four
      int P = 4;
      while (P > 1 && (N % P != 0 || N / P < 64)) P >>= 1;
      npart = P;
      R.npart = P;
      const size_t lds = (size_t)(2 * M * TP + (N / P) * (TP / 16)) * sizeof(double);
      static hpx_lds_limit limit;
      if (lds > 48 * 1024) HPX_TRY(limit.ensure(reinterpret_cast<const void*>(&k_resid), lds));
      hipLaunchKernelGGL(k_resid, dim3(P, nbl), dim3(256), lds, st, R);
    }
    HPX_HIP(hipGetLastError());
    if (p->dense_noise) {
      // first ln-posterior term with the full matrix over the unflagged channels: sum_t (w r_t)^H Ninv (w r_t)
      // (pspec.py:472-477); k_resid left the masked residual behind, v = Ninv (w r) goes to the Z scratch
      // (the data columns only: with flags TP also counts the Woodbury columns, 112 against 32 at the C3 shape)
      HPX_TRY(hpx_launch_dft(nbl, NP, TPd, p->NIre, p->NIim, 1, R.Rdre, R.Rdim, (long)NP * TP, TP, nullptr, 0,
                             p->Zre, p->Zim, (long)NP * p->ncolR, p->ncolR, 1.0, st, 0, (long)NP * NP));
      hipLaunchKernelGGL(k_quadform, dim3(nbl), dim3(256), 0, st, R.Rdre, R.Rdim, (long)NP * TP, TP, p->Zre, p->Zim,
                         (long)NP * p->ncolR, p->ncolR, p->lnpart, N, T);
      HPX_HIP(hipGetLastError());
    } else if (p->per_time == 2) {      // ... with each time's own matrix (the child's units)
      hipLaunchKernelGGL(k_quadform_pt, dim3(T, nbl), dim3(256), 0, st, R.Rdre, R.Rdim, p->child->NIre, p->child->NIim,
                         p->Zre, N, T, NP, TP);                      // (the Z scratch holds the per-time terms)
      hipLaunchKernelGGL(k_quadform_pt_sum, dim3((nbl + 255) / 256), dim3(256), 0, st, p->Zre, p->lnpart, T, nbl);
      HPX_HIP(hipGetLastError());
    }
  }
  if (p->any_flags) {   // |F (w s)|^2 for the masked S^-1 quadratic form (pspec.py:479-483)
    HPX_TRY(hpx_launch_dft(nbl, NP, TPd, p->Fopre, p->Fopim, 0, p->Gre, p->Gim, (long)NP * TP, TP,
                           nullptr, 0, p->Zre, p->Zim, (long)NP * p->ncolR, p->ncolR, 1.0, st,
                           N == NP));
    hipLaunchKernelGGL(k_betam, dim3(16, nbl), dim3(256), 0, st, p->Zre, p->Zim, p->betam, N, T, NP,
                       p->ncolR);
    HPX_HIP(hipGetLastError());
  }
  HPX_TRY(mark(p, st));
  DrawArgs D;
  D.beta = p->beta; D.betam = p->betam; D.lnp1 = p->lnp1;
  D.bpart = p->bpart; D.lnpart = p->lnpart; D.npart = npart;
  D.uni = p->uni + (long)it_abs * N; D.igy = p->igy + (long)it_abs * N;
  D.xgrid = p->xgrid; D.pmap = p->pmap;
  D.ps_forced = O.ps_forced; D.forced_bstride = O.forced_bstride;
  D.ia = p->ia; D.ps_cur = p->ps_cur;
  D.ps_out = O.ps_out; D.ps_bstride = O.ps_bstride;
  D.N = N; D.T = T; D.ngrid = p->ngrid; D.prior_shared = p->prior_shared;
  D.any_flags = p->any_flags; D.lgam_T = p->lgam_T;
  D.lnblk = p->lnblk; D.dcount = p->dcount;
  D.lnpost_out = O.lnpost_out; D.lnpost_pitch = O.lnpost_pitch;      // (written by the kernel: no copy afterwards)
  // slices per baseline: as many as leave no CU without work, in blocks of 64 channels
  int nslice = 1;
  {
    int dev = 0, cus = 0;
    static int cu_of[32] = {};
    if (hipGetDevice(&dev) == hipSuccess && dev >= 0 && dev < 32) {
      if (!cu_of[dev]) (void)hipDeviceGetAttribute(&cu_of[dev], hipDeviceAttributeMultiprocessorCount, dev);
      cus = cu_of[dev];
    }
    while (nslice < 8 && 2 * nslice * nbl <= cus && 2 * nslice <= (N + 63) / 64) nslice *= 2;
    while (((N + 63) / 64 + nslice - 1) / nslice > 64) nslice *= 2;      // (a slice keeps at most 64 block sums)
  }
  hipLaunchKernelGGL(k_draw, dim3(nslice, nbl), dim3(256), (size_t)N * sizeof(int) + 8, st, D);
  HPX_HIP(hipGetLastError());
  HPX_TRY(mark(p, st));
  return HPX_OK;
}

static int finish_run(hpx_plan* p, int niter, double* ps_last, hipStream_t st) {
  const int nbl = p->nbl;
  p->have_ps = 1;
  if (ps_last)
    HPX_HIP(hipMemcpyAsync(ps_last, p->ps_cur, (size_t)nbl * p->N * sizeof(double),
                           hipMemcpyDeviceToDevice, st));
  HPX_HIP(hipStreamSynchronize(st));
  if (p->profiling) {
    for (int s = 0; s < HPX_NSTAGE; ++s) p->stage_ms[s] = 0.f;
    const int per = HPX_NSTAGE + 1;
    for (int it = 0; it < niter; ++it)
      for (int s = 0; s < HPX_NSTAGE; ++s) {
        float ms = 0.f;
        HPX_HIP(hipEventElapsedTime(&ms, p->events[it * per + s], p->events[it * per + s + 1]));
        p->stage_ms[s] += ms;
      }
  }
  std::vector<int32_t> info(nbl);   // report the first non-positive pivot, if any
  HPX_HIP(hipMemcpy(info.data(), p->info, (size_t)nbl * sizeof(int32_t), hipMemcpyDeviceToHost));
  // a hand-off time-out of the split factor first: the factor of such a system is incomplete, whatever else is flagged
  for (int b = 0; b < nbl; ++b)
    if (info[b] & HPX_INFO_TIMEOUT) {
      hpx_set_error("split factor: hand-off between the workgroups of baseline %d timed out at iteration %d (another "
                    "process on this GPU? switch the form off: HPX_OPT_FACTOR_SPLIT = 0)", b,
                    (info[b] & ~HPX_INFO_TIMEOUT) - 1);
      return HPX_ETIMEOUT;
    }
  for (int b = 0; b < nbl; ++b)
    if (info[b] != 0) {
      hpx_set_error("non-positive pivot: baseline %d, iteration %d", b, info[b] - 1);
      return HPX_ENOTPD;
    }
  if (p->child) {
    std::vector<int32_t> ci(p->child->nbl);
    HPX_HIP(hipMemcpy(ci.data(), p->child->info, ci.size() * sizeof(int32_t), hipMemcpyDeviceToHost));
    for (size_t u = 0; u < ci.size(); ++u)
      if (ci[u] & HPX_INFO_TIMEOUT) {
        hpx_set_error("split factor: hand-off between the workgroups of baseline %d, time %d timed out at iteration %d",
                      (int)(u / p->T), (int)(u % p->T), (ci[u] & ~HPX_INFO_TIMEOUT) - 1);
        return HPX_ETIMEOUT;
      }
    for (size_t u = 0; u < ci.size(); ++u)
      if (ci[u] != 0) {
        hpx_set_error("non-positive pivot: baseline %d, time %d, iteration %d", (int)(u / p->T), (int)(u % p->T),
                      ci[u] - 1);
        return HPX_ENOTPD;
      }
  }
  return HPX_OK;
}

// what one run of `niter` iterations reads and writes (hpx_gibbs_run's arguments)
struct RunArgs {
  const double *ps0, *ps_forced;
  double *ps_out, *lnpost_out, *cr_out, *fg_out, *chisq_out, *ps_last;
  int iter0, niter, thin;
  hipStream_t st;
};

static int run_check(const hpx_plan* p, const RunArgs& A, const char* who) {
  const char* why = nullptr;
  if (!(p && p->have_static)) why = "plan has no static inputs";
  else if (!(A.ps_out && A.lnpost_out && A.niter > 0 && A.iter0 >= 0)) why = "bad argument";
  else if (!(p->uni && A.iter0 + A.niter <= p->niter_tab)) why = "random tables too short";
  else if (!(A.ps0 || p->have_ps)) why = "no starting bandpowers (ps0 is NULL on a plan that has not run yet)";
  if (why) {
    hpx_set_error("%s: %s", who, why);
    return HPX_EINVAL;
  }
  return HPX_OK;
}

static int run_begin(hpx_plan* p, const RunArgs& A) {
  hipStream_t st = A.st;
  if (A.ps0) {
    hipLaunchKernelGGL(k_set_a, dim3(256), dim3(256), 0, st, A.ps0, p->ia, p->ps_cur, (long)p->nbl * p->N,
                       (double)p->N);
    HPX_HIP(hipGetLastError());
  }
  HPX_HIP(hipMemsetAsync(p->info, 0, (size_t)p->nbl * sizeof(int32_t), st));
  if (p->child) HPX_HIP(hipMemsetAsync(p->child->info, 0, (size_t)p->child->nbl * sizeof(int32_t), st));
  p->ev_used = 0;
  return HPX_OK;
}

// Per-time units with a full noise matrix AND flagged channels: the Woodbury correction per unit (as post_solve's,
// with one data column), on the unit solutions X = [Y_r | Y_P] of the unflagged-noise systems
static int child_woodbury(hpx_plan* c, int iter_tag, hipStream_t st) {
  const int N = c->N, M = c->M;
  HPX_TRY(hpx_launch_dft(c->nbl, c->NP, c->TP, c->Fopre, c->Fopim, 1, c->Xre, c->Xim, (long)c->npad * c->TP,
                         c->TP, nullptr, 0, c->Sre, c->Sim, (long)c->NP * c->TP, c->TP,
                         1.0 / sqrt((double)N), st, N == c->NP));
  const int fm = c->wb_fmax;
  hipLaunchKernelGGL(k_wb_system, dim3(c->nbl), dim3(256), 0, st, c->Sre, c->Sim, c->Xre, c->Xim, c->Fre,
                     c->Fim, c->fg_shared, c->wb_flist, c->wb_fcount, c->wb_W, fm, N, M, 1, c->NP, c->TP,
                     c->npad);
  HPX_TRY(launch_wb_solve(c->nbl, c->wb_W, c->wb_fcount, fm, 1, c->info, iter_tag, st));
  hipLaunchKernelGGL(k_wb_correct, dim3(8, c->nbl), dim3(256), 0, st, c->Sre, c->Sim, c->Xre, c->Xim, c->wb_W,
                     c->wb_fcount, fm, 1, c->NP, c->TP, c->npad);
  HPX_HIP(hipGetLastError());
  return HPX_OK;
}

// iteration `it` (0-based within the run) of plan p: everything is enqueued on A.st, nothing waits
static int run_iteration(hpx_plan* p, const RunArgs& A, int it) {
  hipStream_t st = A.st;
  const int nbl = p->nbl, N = p->N, M = p->M, T = p->T, TP = p->TP;
  const int iter0 = A.iter0, niter = A.niter, thin = A.thin;
  const int nkeep = (niter + thin - 1) / thin;
  {
    HPX_TRY(mark(p, st));
    // Only the edge rows (foreground modes, padding, right-hand sides: rows >= rmin) are
    // assembled; the signal x signal part of the matrix is generated inside the factor kernel.
    if (p->solver == HPX_SOLVER_FLAT || p->solver == HPX_SOLVER_LOWRANK) {
      // flat noise: diagonal + border system solved through its Schur complement (hpx_flat.hip
      // without flags, hpx_lowrank.hip with flags); booked under the "factor" stage
      HPX_TRY(mark(p, st));
      if (p->solver == HPX_SOLVER_FLAT) HPX_TRY(hpx_launch_solve_flat(p, iter0 + it + 1, st));
      else HPX_TRY(hpx_launch_solve_lowrank(p, iter0 + it + 1, st));
      HPX_TRY(mark(p, st));
      HPX_TRY(mark(p, st));
    } else {
      const hpx_gen_batch gen = gen_of(p);
      if (p->per_time) {
        // one system per (baseline, time): assemble / factor / solve over the child's nbl*T units,
        // then the solutions go to their time column of this plan's X
        hpx_plan* c = p->child;
        const hpx_gen_batch gc = gen_of_child(p);
        if (c->dense_noise)       // full noise matrix per unit: the whole matrix is laid out
          hipLaunchKernelGGL(k_assemble, dim3(c->npad / 16, c->nbl), dim3(256), 0, st, gc, c->L, c->npad, c->ld, 0);
        else if (gc.ere) hipLaunchKernelGGL(k_assemble_tail, dim3(c->nbl), dim3(256), 0, st, gc, c->L, c->npad, c->ld);
        else hipLaunchKernelGGL(k_assemble_edge, dim3(c->nbl, 1), dim3(256), 0, st, gc, c->L, c->npad, c->ld);
        HPX_HIP(hipGetLastError());
        HPX_TRY(mark(p, st));
        HPX_TRY(hpx_launch_factor(c->nbl, c->npad, c->ld, c->L, c->Wre, c->Wim, c->Vt, c->info, iter0 + it + 1,
                                  c->dense_noise ? nullptr : &gc, st, p->allow_split));
        HPX_TRY(mark(p, st));
        HPX_TRY(hpx_launch_backsolve(c->nbl, c->npad, c->TP, c->ld, c->L, c->Wre, c->Wim, c->Xre, c->Xim, st));
        if (c->dense_noise == 2) HPX_TRY(child_woodbury(c, iter0 + it + 1, st));
        hipLaunchKernelGGL(k_pt_gather, dim3(32, nbl), dim3(256), 0, st, c->Xre, c->Xim, p->Xre, p->Xim, T, p->npad,
                           TP, c->TP);
        HPX_HIP(hipGetLastError());
        HPX_TRY(mark(p, st));
      } else if (p->dense_noise) {
        // general Hermitian C: the whole augmented matrix is laid out, then factored in place
        HPX_TRY(launch_assemble(p, st, 0));
        HPX_TRY(mark(p, st));
        HPX_TRY(hpx_launch_factor(nbl, p->npad, p->ld, p->L, p->Wre, p->Wim, p->Vt, p->info, iter0 + it + 1,
                                  nullptr, st, p->allow_split));
      } else {
        HPX_TRY(launch_assemble_edge(p, st));
        HPX_TRY(mark(p, st));
        HPX_TRY(hpx_launch_factor(nbl, p->npad, p->ld, p->L, p->Wre, p->Wim, p->Vt, p->info, iter0 + it + 1,
                                  &gen, st, p->allow_split));
      }
      if (!p->per_time) {
        HPX_TRY(mark(p, st));
        HPX_TRY(hpx_launch_backsolve(nbl, p->npad, TP, p->ld, p->L, p->Wre, p->Wim, p->Xre, p->Xim, st));
        HPX_TRY(mark(p, st));
      }
    }
    const bool keep = (it % thin) == 0;
    const long slot = it / thin;
    IterOut O;
    O.ps_forced = A.ps_forced ? A.ps_forced + (long)it * N : nullptr;
    O.forced_bstride = (long)niter * N;
    O.ps_out = A.ps_out + (long)it * N; O.ps_bstride = (long)niter * N;
    O.lnpost_out = A.lnpost_out + it; O.lnpost_pitch = niter;
    O.cr_bstride = (long)nkeep * T * N * 2;
    O.fg_bstride = (long)nkeep * T * M * 2;
    O.chisq_bstride = (long)nkeep * T * N;
    O.cr_out = (A.cr_out && keep) ? A.cr_out + slot * T * N * 2 : nullptr;
    O.fg_out = (A.fg_out && keep) ? A.fg_out + slot * T * M * 2 : nullptr;
    O.chisq_out = (A.chisq_out && keep) ? A.chisq_out + slot * T * N : nullptr;
    HPX_TRY(post_solve(p, iter0 + it, O, st));
  }
  return HPX_OK;
}

extern "C" int hpx_gibbs_run(hpx_plan* p, const double* ps0, int iter0, int niter,
                             const double* ps_forced, double* ps_out, double* lnpost_out,
                             double* cr_out, double* fg_out, double* chisq_out, int thin,
                             double* ps_last, void* stream) {
  const RunArgs A = {ps0, ps_forced, ps_out, lnpost_out, cr_out, fg_out, chisq_out, ps_last,
                     iter0, niter, thin < 1 ? 1 : thin, (hipStream_t)stream};
  HPX_TRY(run_check(p, A, "hpx_gibbs_run"));
  HPX_TRY(run_begin(p, A));
  for (int it = 0; it < niter; ++it) HPX_TRY(run_iteration(p, A, it));
  return finish_run(p, niter, ps_last, A.st);
}

// ---- general first iteration ---------------------------------------------------------------
namespace {
// (nbl,N,N) c128 row-major -> planar [b][NP][NP] (zero padded); entry [k][x] = M[k][x]
__global__ void k_mat_planar(const double* __restrict__ m, double* __restrict__ re,
                             double* __restrict__ im, const int N, const int NP) {
  const int b = blockIdx.y;
  const long tot = (long)NP * NP;
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < tot;
       e += (long)gridDim.x * blockDim.x) {
    const int k = (int)(e / NP), x = (int)(e % NP);
    double vr = 0.0, vi = 0.0;
    if (k < N && x < N) {
      vr = m[(((long)b * N + k) * N + x) * 2];
      vi = m[(((long)b * N + k) * N + x) * 2 + 1];
    }
    re[(long)b * tot + e] = vr;
    im[(long)b * tot + e] = vi;
  }
}
// explicit circulant C[k][x] = circ[(k - x) mod N]
__global__ void k_circ_matrix(const double* __restrict__ cre, const double* __restrict__ cim,
                              double* __restrict__ re, double* __restrict__ im, const int N,
                              const int NP) {
  const int b = blockIdx.y;
  const long tot = (long)NP * NP;
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < tot;
       e += (long)gridDim.x * blockDim.x) {
    const int k = (int)(e / NP), x = (int)(e % NP);
    double vr = 0.0, vi = 0.0;
    if (k < N && x < N) {
      const int m = (k - x + N) % N;
      vr = cre[(long)b * N + m];
      vi = cim[(long)b * N + m];
    }
    re[(long)b * tot + e] = vr;
    im[(long)b * tot + e] = vi;
  }
}
// K'_aug for a general Sh': top-left I + XT, fg rows conj(GS), right-hand sides conj(QS + P2)
__global__ __launch_bounds__(256) void k_assemble_general(
    const hpx_gen_batch B, const double* __restrict__ XTre, const double* __restrict__ XTim,
    const double* __restrict__ RSre, const double* __restrict__ RSim, double* __restrict__ L_all,
    const int npad, const int ld) {
  const int b = blockIdx.y, cb = blockIdx.x;
  const hpx_gen G = hpx_gen_for(B, b);
  const int N = B.N, M = B.M, NP = B.NP, TP = B.TP;
  double* L = L_all + (long)b * npad * ld * 2;
  const double* xr = XTre + (long)b * NP * NP;
  const double* xi = XTim + (long)b * NP * NP;
  const double* rr = RSre + (long)b * NP * B.ncol;
  const double* ri = RSim + (long)b * NP * B.ncol;
  const int rbeg = cb * 16, nrow = ld - rbeg;
  for (int e = threadIdx.x; e < 16 * nrow; e += 256) {
    const int c = rbeg + e / nrow, r = rbeg + e % nrow;
    double vr = 0.0, vi = 0.0;
    if (r >= c && c < N && (r < N + M || r >= npad)) {
      if (r < N) {                       // out[x=r][col=c] of the GEMM: stored [x][c]
        vr = xr[(long)r * NP + c] + (r == c ? 1.0 : 0.0);
        vi = (r == c) ? 0.0 : xi[(long)r * NP + c];
      } else if (r < N + M) {            // conj((Sh' G)[c][m])
        vr = rr[(long)c * B.ncol + TP + (r - N)];
        vi = -ri[(long)c * B.ncol + TP + (r - N)];
      } else {                           // conj((Sh' Q)[c][t] + P2[c][t])
        const int t = r - npad;
        vr = rr[(long)c * B.ncol + t];
        vi = ri[(long)c * B.ncol + t];
        if (B.has_omega) { vr += G.p2re[(long)c * TP + t]; vi += G.p2im[(long)c * TP + t]; }      // (the unit's own block)
        vi = -vi;
      }
    } else {
      hpx_gen_entry(G, r, c, npad, vr, vi);   // H, P4, padding: independent of S
      if (c < N) { vr = 0.0; vi = 0.0; }      // (c < N cases are all handled above)
    }
    const long o = HPX_LIDX(r, c, npad);
    L[o] = vr;
    L[o + 16] = vi;
  }
}
// X rows [0,N) <- s' (from scratch G), a <- 1
__global__ void k_take_sprime(const double* __restrict__ Gre, const double* __restrict__ Gim,
                              double* __restrict__ Xre, double* __restrict__ Xim,
                              const int N, const int NP, const int TP, const int npad) {
  const int b = blockIdx.y;
  const long tot = (long)N * TP;
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < tot;
       e += (long)gridDim.x * blockDim.x) {
    Xre[(long)b * npad * TP + e] = Gre[(long)b * NP * TP + e];
    Xim[(long)b * npad * TP + e] = Gim[(long)b * NP * TP + e];
  }
}
}  // namespace

extern "C" int hpx_gibbs_step_general(hpx_plan* p, const double* shp, int iter0, double* ps_out,
                                      double* lnpost_out, double* cr_out, double* fg_out,
                                      double* chisq_out, double* ps_last, void* stream) {
  HPX_REQUIRE(p && p->have_static && shp && ps_out && lnpost_out, "hpx_gibbs_step_general: bad argument");
  HPX_REQUIRE(!p->per_time || p->child, "hpx_gibbs_step_general: per-time plan without its units");
  HPX_REQUIRE(p->uni && iter0 >= 0 && iter0 < p->niter_tab, "hpx_gibbs_step_general: random tables too short");
  hipStream_t st = (hipStream_t)stream;
  const int nbl = p->nbl, N = p->N, M = p->M, T = p->T, NP = p->NP, TP = p->TP;
  if (p->per_time) {
    // One system per (baseline, time) as in run_iteration's per-time branch, each assembled from the explicit
    // K' = [[I + Sh' C_t Sh', Sh' G_t], [., H_t]] of its own circulant C_t (flags and noise of that time):
    // Sh' is the baseline's, copied to its Ntimes units for the batched products.
    hpx_plan* c = p->child;
    const int units = c->nbl;
    const size_t um = (size_t)units * NP * NP, ur = (size_t)units * NP * c->ncolR;
    const long mstr = (long)NP * NP;
    if (!p->SHre) { HPX_TRY(dev_alloc(p, &p->SHre, (size_t)nbl * NP * NP)); HPX_TRY(dev_alloc(p, &p->SHim, (size_t)nbl * NP * NP)); }
    if (!c->SHre) {
      HPX_TRY(dev_alloc(c, &c->SHre, um)); HPX_TRY(dev_alloc(c, &c->SHim, um));
      HPX_TRY(dev_alloc(c, &c->CMre, um)); HPX_TRY(dev_alloc(c, &c->CMim, um));
      HPX_TRY(dev_alloc(c, &c->Y1re, um)); HPX_TRY(dev_alloc(c, &c->Y1im, um));
      HPX_TRY(dev_alloc(c, &c->XTre, um)); HPX_TRY(dev_alloc(c, &c->XTim, um));
      HPX_TRY(dev_alloc(c, &c->RSre, ur)); HPX_TRY(dev_alloc(c, &c->RSim, ur));
    }
    hipLaunchKernelGGL(k_mat_planar, dim3(64, nbl), dim3(256), 0, st, shp, p->SHre, p->SHim, N, NP);
    hipLaunchKernelGGL(k_pt_expand_fg, dim3(64, units), dim3(256), 0, st, p->SHre, c->SHre, T, mstr);
    hipLaunchKernelGGL(k_pt_expand_fg, dim3(64, units), dim3(256), 0, st, p->SHim, c->SHim, T, mstr);
    if (c->dense_noise) {       // a full noise matrix per unit: C_t = U^H Ninv_t U as laid out at set-up
      HPX_HIP(hipMemcpyAsync(c->CMre, c->CDre, um * sizeof(double), hipMemcpyDeviceToDevice, st));
      HPX_HIP(hipMemcpyAsync(c->CMim, c->CDim, um * sizeof(double), hipMemcpyDeviceToDevice, st));
    } else {
      hipLaunchKernelGGL(k_circ_matrix, dim3(64, units), dim3(256), 0, st, c->Cre, c->Cim, c->CMre, c->CMim, N, NP);
    }
    HPX_HIP(hipGetLastError());
    HPX_TRY(hpx_launch_dft(units, NP, NP, c->CMre, c->CMim, 1, c->SHre, c->SHim, mstr, NP, nullptr, 0, c->Y1re, c->Y1im,
                           mstr, NP, 1.0, st, 0, mstr));
    HPX_TRY(hpx_launch_dft(units, NP, NP, c->SHre, c->SHim, 1, c->Y1re, c->Y1im, mstr, NP, nullptr, 0, c->XTre, c->XTim,
                           mstr, NP, 1.0, st, 0, mstr));
    HPX_TRY(hpx_launch_dft(units, NP, c->ncolR, c->SHre, c->SHim, 1, c->Rre, c->Rim, (long)NP * c->ncolR, c->ncolR,
                           nullptr, 0, c->RSre, c->RSim, (long)NP * c->ncolR, c->ncolR, 1.0, st, 0, mstr));
    HPX_HIP(hipMemsetAsync(p->info, 0, (size_t)nbl * sizeof(int32_t), st));
    HPX_HIP(hipMemsetAsync(c->info, 0, (size_t)units * sizeof(int32_t), st));
    p->ev_used = 0;
    HPX_TRY(mark(p, st));
    hipLaunchKernelGGL(k_assemble_general, dim3(c->npad / 16, units), dim3(256), 0, st, gen_of_child(p), c->XTre,
                       c->XTim, c->RSre, c->RSim, c->L, c->npad, c->ld);
    HPX_HIP(hipGetLastError());
    HPX_TRY(mark(p, st));
    HPX_TRY(hpx_launch_factor(units, c->npad, c->ld, c->L, c->Wre, c->Wim, c->Vt, c->info, iter0 + 1, nullptr, st,
                              p->allow_split));
    HPX_TRY(mark(p, st));
    HPX_TRY(hpx_launch_backsolve(units, c->npad, c->TP, c->ld, c->L, c->Wre, c->Wim, c->Xre, c->Xim, st));
    // s' = Sh' y' per unit, then the units' solutions to their time column of this plan's X
    HPX_TRY(hpx_launch_dft(units, NP, c->TP, c->SHre, c->SHim, 1, c->Xre, c->Xim, (long)c->npad * c->TP, c->TP, nullptr,
                           0, c->Gre, c->Gim, (long)NP * c->TP, c->TP, 1.0, st, 0, mstr));
    hipLaunchKernelGGL(k_take_sprime, dim3(32, units), dim3(256), 0, st, c->Gre, c->Gim, c->Xre, c->Xim, N, NP, c->TP,
                       c->npad);
    HPX_HIP(hipGetLastError());
    if (c->dense_noise == 2) HPX_TRY(child_woodbury(c, iter0 + 1, st));      // (every column went through Sh' alike)
    hipLaunchKernelGGL(k_pt_gather, dim3(32, nbl), dim3(256), 0, st, c->Xre, c->Xim, p->Xre, p->Xim, T, p->npad, TP,
                       c->TP);
    HPX_HIP(hipGetLastError());
    HPX_TRY(mark(p, st));
    IterOut O;
    O.ps_forced = nullptr; O.forced_bstride = 0;
    O.ps_out = ps_out; O.ps_bstride = N;
    O.lnpost_out = lnpost_out; O.lnpost_pitch = 1;
    O.cr_bstride = (long)T * N * 2; O.fg_bstride = (long)T * M * 2; O.chisq_bstride = (long)T * N;
    O.cr_out = cr_out; O.fg_out = fg_out; O.chisq_out = chisq_out;
    HPX_TRY(post_solve(p, iter0, O, st));
    return finish_run(p, 1, ps_last, st);
  }
  const size_t msz = (size_t)nbl * NP * NP, rsz = (size_t)nbl * NP * p->ncolR;
  if (!p->SHre) {
    HPX_TRY(dev_alloc(p, &p->SHre, msz)); HPX_TRY(dev_alloc(p, &p->SHim, msz));
    HPX_TRY(dev_alloc(p, &p->CMre, msz)); HPX_TRY(dev_alloc(p, &p->CMim, msz));
    HPX_TRY(dev_alloc(p, &p->Y1re, msz)); HPX_TRY(dev_alloc(p, &p->Y1im, msz));
    HPX_TRY(dev_alloc(p, &p->XTre, msz)); HPX_TRY(dev_alloc(p, &p->XTim, msz));
    HPX_TRY(dev_alloc(p, &p->RSre, rsz)); HPX_TRY(dev_alloc(p, &p->RSim, rsz));
  }
  const long mstr = (long)NP * NP;
  hipLaunchKernelGGL(k_mat_planar, dim3(64, nbl), dim3(256), 0, st, shp, p->SHre, p->SHim, N, NP);
  if (p->dense_noise) {
    HPX_HIP(hipMemcpyAsync(p->CMre, p->CDre, msz * sizeof(double), hipMemcpyDeviceToDevice, st));
    HPX_HIP(hipMemcpyAsync(p->CMim, p->CDim, msz * sizeof(double), hipMemcpyDeviceToDevice, st));
  } else {
    hipLaunchKernelGGL(k_circ_matrix, dim3(64, nbl), dim3(256), 0, st, p->Cre, p->Cim, p->CMre, p->CMim,
                       N, NP);
  }
  HPX_HIP(hipGetLastError());
  // The dense kernel computes out = W in with W[x][k] read from the planar buffer at [k][x];
  // both C and Sh' are Hermitian, so the stored row-major matrix is conj(W^T): conjW = 1.
  HPX_TRY(hpx_launch_dft(nbl, NP, NP, p->CMre, p->CMim, 1, p->SHre, p->SHim, mstr, NP, nullptr, 0,
                         p->Y1re, p->Y1im, mstr, NP, 1.0, st, 0, mstr));        // Y1 = C Sh'
  HPX_TRY(hpx_launch_dft(nbl, NP, NP, p->SHre, p->SHim, 1, p->Y1re, p->Y1im, mstr, NP, nullptr, 0,
                         p->XTre, p->XTim, mstr, NP, 1.0, st, 0, mstr));        // XT = Sh' C Sh'
  HPX_TRY(hpx_launch_dft(nbl, NP, p->ncolR, p->SHre, p->SHim, 1, p->Rre, p->Rim,
                         (long)NP * p->ncolR, p->ncolR, nullptr, 0, p->RSre, p->RSim,
                         (long)NP * p->ncolR, p->ncolR, 1.0, st, 0, mstr));     // RS = Sh' [Q | G | .]
  HPX_HIP(hipMemsetAsync(p->info, 0, (size_t)nbl * sizeof(int32_t), st));
  p->ev_used = 0;
  HPX_TRY(mark(p, st));
  hipLaunchKernelGGL(k_assemble_general, dim3(p->npad / 16, nbl), dim3(256), 0, st, gen_of(p),
                     p->XTre, p->XTim, p->RSre, p->RSim, p->L, p->npad, p->ld);
  HPX_HIP(hipGetLastError());
  HPX_TRY(mark(p, st));
  HPX_TRY(hpx_launch_factor(nbl, p->npad, p->ld, p->L, p->Wre, p->Wim, p->Vt, p->info, iter0 + 1, nullptr, st,
                            p->allow_split));
  HPX_TRY(mark(p, st));
  HPX_TRY(hpx_launch_backsolve(nbl, p->npad, TP, p->ld, p->L, p->Wre, p->Wim, p->Xre, p->Xim, st));
  HPX_TRY(mark(p, st));
  // s' = Sh' y'  -> X rows [0,N): beta = N sum |s'|^2 and s = U s' as in the scaled system
  HPX_TRY(hpx_launch_dft(nbl, NP, TP, p->SHre, p->SHim, 1, p->Xre, p->Xim, (long)p->npad * TP, TP,
                         nullptr, 0, p->Gre, p->Gim, (long)NP * TP, TP, 1.0, st, 0, mstr));
  hipLaunchKernelGGL(k_take_sprime, dim3(32, nbl), dim3(256), 0, st, p->Gre, p->Gim, p->Xre, p->Xim,
                     N, NP, TP, p->npad);
  HPX_HIP(hipGetLastError());
  IterOut O;
  O.ps_forced = nullptr; O.forced_bstride = 0;
  O.ps_out = ps_out; O.ps_bstride = N;
  O.lnpost_out = lnpost_out; O.lnpost_pitch = 1;
  O.cr_bstride = (long)T * N * 2; O.fg_bstride = (long)T * M * 2; O.chisq_bstride = (long)T * N;
  O.cr_out = cr_out; O.fg_out = fg_out; O.chisq_out = chisq_out;
  HPX_TRY(post_solve(p, iter0, O, st));
  return finish_run(p, 1, ps_last, st);
}

extern "C" int hpx_invgamma_inversion(int n, int alpha, const double* beta, const double* u,
                                      const double* xgrid, int ngrid, double* out, void* stream) {
  HPX_REQUIRE(n > 0 && alpha >= 1 && beta && u && xgrid && out && ngrid >= 2 && ngrid <= 8192,
              "hpx_invgamma_inversion: bad argument");
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(k_inv_test, dim3(n), dim3(256), 0, st, alpha,
                     lgamma((double)alpha), beta, u, xgrid, ngrid, out);
  HPX_HIP(hipGetLastError());
  HPX_HIP(hipStreamSynchronize(st));
  return HPX_OK;
}
