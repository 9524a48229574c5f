// Plan management: creation / destruction, the random tables, solver and option switches, profiling marks,
// the generator descriptors of a plan; error plumbing of the library.
#include "hpx_chain.h"

namespace {



}  // namespace

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

// ---------------------------------------------------------------------------
int hpx_plan_create_impl(hpx_plan** out, int nbl, int T, int N, int M, int extra_rhs);
extern "C" int hpx_plan_create(hpx_plan** out, int nbl, int T, int N, int M) {
  HPX_REQUIRE(out, "hpx_plan_create: null out");
  HPX_REQUIRE(nbl > 0 && T > 1 && N > 0 && M >= 0, "hpx_plan_create: need nbl>0, T>1, N>0, M>=0");
  return hpx_plan_create_impl(out, nbl, T, N, M, 0);
}
extern "C" int hpx_plan_create_ex(hpx_plan** out, int nbl, int T, int N, int M, int extra_rhs) {
  HPX_REQUIRE(out, "hpx_plan_create_ex: null out");
  HPX_REQUIRE(nbl > 0 && T > 1 && N > 0 && M >= 0 && extra_rhs >= 0 && extra_rhs <= N,
              "hpx_plan_create_ex: need nbl>0, T>1, N>0, M>=0, 0 <= extra_rhs <= N");
  return hpx_plan_create_impl(out, nbl, T, N, M, extra_rhs);
}
int hpx_plan_create_impl(hpx_plan** out, int nbl, int T, int N, int M, int extra_rhs) {
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
  p->split_retry = 1;
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
  // k_draw's slices per baseline: as many as keep the launch within four (256-thread) workgroups per CU, in groups of 16
  // channels (at most 256 per slice); with more than one the group sums of every iteration are kept until
  // the end of a run
  {
    int dev = 0, cus = 0, nslice = 1;
    const int nsub = (p->N + 15) / 16;
    if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    while (nslice < 16 && 2 * nslice * p->nbl <= 4 * cus && 2 * nslice <= nsub) nslice *= 2;
    while ((nsub + nslice - 1) / nslice > 256) nslice *= 2;
    if (nslice > 1 && (niter != p->niter_tab || nslice != p->draw_slices || !p->lnhist))
      HPX_TRY(dev_alloc(p, &p->lnhist, (size_t)niter * p->nbl * (nsub + 1)));
    p->draw_slices = nslice;
  }
  p->niter_tab = niter;
  return HPX_OK;
}
hpx_gen_batch hpx_gen_of_child(const hpx_plan* p) {
  hpx_gen_batch B = hpx_gen_of(p->child);
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

hpx_gen_batch hpx_gen_of(const hpx_plan* p) {
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
    if (key == HPX_OPT_SPLIT_RETRY) {
      p->split_retry = value != 0;
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

extern "C" int hpx_get_option(const hpx_plan* p, int key, int* value) {
  HPX_REQUIRE(p && value, "hpx_get_option: null argument");
  if (key == HPX_OPT_FACTOR_SPLIT) *value = p->allow_split;
  else if (key == HPX_OPT_SPLIT_RETRY) *value = p->split_retry;
  else if (key == HPX_OPT_SPLIT_FALLBACKS) *value = p->split_fallbacks;
  else {
    hpx_set_error("hpx_get_option: key %d is not a plan option", key);
    return HPX_EINVAL;
  }
  return HPX_OK;
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

int hpx_mark(hpx_plan* p, hipStream_t st) {
  if (!p->profiling) return HPX_OK;
  if (p->ev_used == (int)p->events.size()) {
    hipEvent_t ev;
    HPX_HIP(hipEventCreate(&ev));
    p->events.push_back(ev);
  }
  HPX_HIP(hipEventRecord(p->events[p->ev_used++], st));
  return HPX_OK;
}
