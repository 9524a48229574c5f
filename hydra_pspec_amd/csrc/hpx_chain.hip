// The Gibbs iteration: assembly of the augmented system, the run loop, the general first iteration.
#include "hpx_chain.h"

namespace {

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

// ln posterior of iterations iter0 .. iter0 + niter - 1 from the block sums of a sliced k_draw:
// -(chi^2 total) - (the blocks in block order), the arithmetic of the one-slice kernel
__global__ void k_lnpost_combine(const double* __restrict__ hist, const int nbl, const int nsub, const int iter0,
                                 const int niter, double* __restrict__ lnpost_out, const long pitch,
                                 double* __restrict__ lnp1) {
  const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= (long)nbl * niter) return;
  const int b = (int)(e / niter), it = (int)(e % niter);
  const double* h = hist + ((long)(iter0 + it) * nbl + b) * (nsub + 1);
  const double lnp = -h[nsub] - lnpost_blocks(h, nsub);
  lnpost_out[(long)b * pitch + it] = lnp;
  if (it == niter - 1) lnp1[b] = lnp;
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


static int launch_assemble_edge(hpx_plan* p, hipStream_t st) {
  const hpx_gen_batch B = hpx_gen_of(p);
  if (B.ere)      // the factor reads the edge tiles itself: only the last columns are laid out
    hipLaunchKernelGGL(k_assemble_tail, dim3(p->nbl), dim3(256), 0, st, B, p->L, p->npad, p->ld);
  else
    hipLaunchKernelGGL(k_assemble_edge, dim3(p->nbl, 1), dim3(256), 0, st, B, p->L, p->npad, p->ld);
  HPX_HIP(hipGetLastError());
  return HPX_OK;
}

static int launch_assemble(hpx_plan* p, hipStream_t st, int rlo) {
  hipLaunchKernelGGL(k_assemble, dim3(p->npad / 16, p->nbl), dim3(256), 0, st, hpx_gen_of(p), p->L,
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

static int finish_run(hpx_plan* p, int iter0, int niter, double* lnpost_out, long lnpost_pitch, double* ps_last,
                      hipStream_t st) {
  const int nbl = p->nbl;
  p->have_ps = 1;
  if (p->draw_slices > 1) {        // the sliced draw left block sums: the ln posterior of the run's iterations from them
    const long tot = (long)nbl * niter;
    hipLaunchKernelGGL(k_lnpost_combine, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, st, p->lnhist, nbl,
                       (p->N + 15) / 16, iter0, niter, lnpost_out, lnpost_pitch, p->lnp1);
    HPX_HIP(hipGetLastError());
  }
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

// iteration `it` (0-based within the run) of plan p: everything is enqueued on A.st, nothing waits
static int run_iteration(hpx_plan* p, const RunArgs& A, int it) {
  hipStream_t st = A.st;
  const int nbl = p->nbl, N = p->N, M = p->M, T = p->T, TP = p->TP;
  const int iter0 = A.iter0, niter = A.niter, thin = A.thin;
  const int nkeep = (niter + thin - 1) / thin;
  {
    HPX_TRY(hpx_mark(p, st));
    // Only the edge rows (foreground modes, padding, right-hand sides: rows >= rmin) are
    // assembled; the signal x signal part of the matrix is generated inside the factor kernel.
    if (p->solver == HPX_SOLVER_FLAT || p->solver == HPX_SOLVER_LOWRANK) {
      // flat noise: diagonal + border system solved through its Schur complement (hpx_flat.hip
      // without flags, hpx_lowrank.hip with flags); booked under the "factor" stage
      HPX_TRY(hpx_mark(p, st));
      if (p->solver == HPX_SOLVER_FLAT) HPX_TRY(hpx_launch_solve_flat(p, iter0 + it + 1, st));
      else HPX_TRY(hpx_launch_solve_lowrank(p, iter0 + it + 1, st));
      HPX_TRY(hpx_mark(p, st));
      HPX_TRY(hpx_mark(p, st));
    } else {
      const hpx_gen_batch gen = hpx_gen_of(p);
      if (p->per_time) {
        // one system per (baseline, time): assemble / factor / solve over the child's nbl*T units,
        // then the solutions go to their time column of this plan's X
        hpx_plan* c = p->child;
        const hpx_gen_batch gc = hpx_gen_of_child(p);
        if (c->dense_noise)       // full noise matrix per unit: the whole matrix is laid out
          hipLaunchKernelGGL(k_assemble, dim3(c->npad / 16, c->nbl), dim3(256), 0, st, gc, c->L, c->npad, c->ld, 0);
        else if (gc.ere) hipLaunchKernelGGL(k_assemble_tail, dim3(c->nbl), dim3(256), 0, st, gc, c->L, c->npad, c->ld);
        else hipLaunchKernelGGL(k_assemble_edge, dim3(c->nbl, 1), dim3(256), 0, st, gc, c->L, c->npad, c->ld);
        HPX_HIP(hipGetLastError());
        HPX_TRY(hpx_mark(p, st));
        HPX_TRY(hpx_launch_factor(c->nbl, c->npad, c->ld, c->L, c->Wre, c->Wim, c->Vt, c->info, iter0 + it + 1,
                                  c->dense_noise ? nullptr : &gc, st, p->allow_split));
        HPX_TRY(hpx_mark(p, st));
        HPX_TRY(hpx_launch_backsolve(c->nbl, c->npad, c->TP, c->ld, c->L, c->Wre, c->Wim, c->Xre, c->Xim, st));
        if (c->dense_noise == 2) HPX_TRY(hpx_child_woodbury(c, iter0 + it + 1, st));
        hipLaunchKernelGGL(k_pt_gather, dim3(32, nbl), dim3(256), 0, st, c->Xre, c->Xim, p->Xre, p->Xim, T, p->npad,
                           TP, c->TP);
        HPX_HIP(hipGetLastError());
        HPX_TRY(hpx_mark(p, st));
      } else if (p->dense_noise) {
        // general Hermitian C: the whole augmented matrix is laid out, then factored in place
        HPX_TRY(launch_assemble(p, st, 0));
        HPX_TRY(hpx_mark(p, st));
        HPX_TRY(hpx_launch_factor(nbl, p->npad, p->ld, p->L, p->Wre, p->Wim, p->Vt, p->info, iter0 + it + 1,
                                  nullptr, st, p->allow_split));
      } else {
        // a small batch: the split form first -- it generates the last columns itself when the plan has edge
        // tiles, no assembly launch in front of it
        int took = 0;
        const bool try_split = p->allow_split && gen.ere && hpx_factor_split_parts(nbl, p->npad, p->ld) > 0;
        if (try_split) {
          HPX_TRY(hpx_mark(p, st));
          HPX_TRY(hpx_launch_factor_split(nbl, p->npad, p->ld, p->L, p->Wre, p->Wim, p->Vt, p->info, iter0 + it + 1,
                                          &gen, st, &took));
        }
        if (!took) {      // (after a refused split -- no room beside other streams' split launches -- booked under "factor")
          HPX_TRY(launch_assemble_edge(p, st));
          if (!try_split) HPX_TRY(hpx_mark(p, st));
          HPX_TRY(hpx_launch_factor(nbl, p->npad, p->ld, p->L, p->Wre, p->Wim, p->Vt, p->info, iter0 + it + 1,
                                    &gen, st, try_split ? 0 : p->allow_split));
        }
      }
      if (!p->per_time) {
        HPX_TRY(hpx_mark(p, st));
        HPX_TRY(hpx_launch_backsolve(nbl, p->npad, TP, p->ld, p->L, p->Wre, p->Wim, p->Xre, p->Xim, st));
        HPX_TRY(hpx_mark(p, st));
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
    HPX_TRY(hpx_post_solve(p, iter0 + it, O, st));
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
  // a run that continues from the plan's own bandpowers keeps a copy of them while the split factor may be taken: the
  // run is repeated from there should the form's hand-off time out
  const bool may_split = p->allow_split && p->split_retry && !p->per_time && p->solver == HPX_SOLVER_DENSE &&
                         hpx_factor_split_parts(p->nbl, p->npad, p->ld) > 0;
  if (may_split && !ps0) {
    if (!p->ps_start) HPX_TRY(dev_alloc(p, &p->ps_start, (size_t)p->nbl * p->N));
    HPX_HIP(hipMemcpyAsync(p->ps_start, p->ps_cur, (size_t)p->nbl * p->N * sizeof(double), hipMemcpyDeviceToDevice, A.st));
  }
  HPX_TRY(run_begin(p, A));
  for (int it = 0; it < niter; ++it) HPX_TRY(run_iteration(p, A, it));
  int rc = finish_run(p, iter0, niter, lnpost_out, niter, ps_last, A.st);
  if (rc == HPX_ETIMEOUT && may_split) {
    // The parts of a system did not run side by side (the launcher's books cover this process only: another process
    // on the GPU).  Nothing of the run is kept: the plan leaves the form for good and the run is repeated on the
    // one-workgroup kernel -- the chain a batch too large to split would have given (tests/test_gpu_split.py).
    p->allow_split = 0;
    if (p->child) p->child->allow_split = 0;
    p->split_fallbacks += 1;
    RunArgs B = A;
    if (!ps0) B.ps0 = p->ps_start;
    HPX_TRY(run_begin(p, B));
    for (int it = 0; it < niter; ++it) HPX_TRY(run_iteration(p, B, it));
    rc = finish_run(p, iter0, niter, lnpost_out, niter, ps_last, A.st);
  }
  return rc;
}

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
    HPX_TRY(hpx_mark(p, st));
    hipLaunchKernelGGL(k_assemble_general, dim3(c->npad / 16, units), dim3(256), 0, st, hpx_gen_of_child(p), c->XTre,
                       c->XTim, c->RSre, c->RSim, c->L, c->npad, c->ld);
    HPX_HIP(hipGetLastError());
    HPX_TRY(hpx_mark(p, st));
    HPX_TRY(hpx_launch_factor(units, c->npad, c->ld, c->L, c->Wre, c->Wim, c->Vt, c->info, iter0 + 1, nullptr, st,
                              p->allow_split));
    HPX_TRY(hpx_mark(p, st));
    HPX_TRY(hpx_launch_backsolve(units, c->npad, c->TP, c->ld, c->L, c->Wre, c->Wim, c->Xre, c->Xim, st));
    // s' = Sh' y' per unit, then the units' solutions to their time column of this plan's X
    HPX_TRY(hpx_launch_dft(units, NP, c->TP, c->SHre, c->SHim, 1, c->Xre, c->Xim, (long)c->npad * c->TP, c->TP, nullptr,
                           0, c->Gre, c->Gim, (long)NP * c->TP, c->TP, 1.0, st, 0, mstr));
    hipLaunchKernelGGL(k_take_sprime, dim3(32, units), dim3(256), 0, st, c->Gre, c->Gim, c->Xre, c->Xim, N, NP, c->TP,
                       c->npad);
    HPX_HIP(hipGetLastError());
    if (c->dense_noise == 2) HPX_TRY(hpx_child_woodbury(c, iter0 + 1, st));      // (every column went through Sh' alike)
    hipLaunchKernelGGL(k_pt_gather, dim3(32, nbl), dim3(256), 0, st, c->Xre, c->Xim, p->Xre, p->Xim, T, p->npad, TP,
                       c->TP);
    HPX_HIP(hipGetLastError());
    HPX_TRY(hpx_mark(p, st));
    IterOut O;
    O.ps_forced = nullptr; O.forced_bstride = 0;
    O.ps_out = ps_out; O.ps_bstride = N;
    O.lnpost_out = lnpost_out; O.lnpost_pitch = 1;
    O.cr_bstride = (long)T * N * 2; O.fg_bstride = (long)T * M * 2; O.chisq_bstride = (long)T * N;
    O.cr_out = cr_out; O.fg_out = fg_out; O.chisq_out = chisq_out;
    HPX_TRY(hpx_post_solve(p, iter0, O, st));
    return finish_run(p, iter0, 1, lnpost_out, 1, ps_last, st);
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
  HPX_TRY(hpx_mark(p, st));
  hipLaunchKernelGGL(k_assemble_general, dim3(p->npad / 16, nbl), dim3(256), 0, st, hpx_gen_of(p),
                     p->XTre, p->XTim, p->RSre, p->RSim, p->L, p->npad, p->ld);
  HPX_HIP(hipGetLastError());
  HPX_TRY(hpx_mark(p, st));
  HPX_TRY(hpx_launch_factor(nbl, p->npad, p->ld, p->L, p->Wre, p->Wim, p->Vt, p->info, iter0 + 1, nullptr, st,
                            p->allow_split));
  HPX_TRY(hpx_mark(p, st));
  HPX_TRY(hpx_launch_backsolve(nbl, p->npad, TP, p->ld, p->L, p->Wre, p->Wim, p->Xre, p->Xim, st));
  HPX_TRY(hpx_mark(p, st));
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
  HPX_TRY(hpx_post_solve(p, iter0, O, st));
  return finish_run(p, iter0, 1, lnpost_out, 1, ps_last, st);
}
