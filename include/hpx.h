/* hpx.h -- C-ABI of the MI355X-native hydra-pspec Gibbs hot path (libhpx.so).
 *
 * The reference (HydraRadio/hydra-pspec) is pure Python and has no FFI of its
 * own; this header is the boundary a binding for it would target (ctypes stub
 * in INTEGRATION.md; the in-tree host side is hydra_pspec_amd/hpx.py).  Each
 * entry point names the reference code it replaces (paths under the reference
 * repo root).
 *
 * Conventions
 *  - All array arguments are DEVICE pointers (HBM) unless marked "host".
 *  - Complex arrays are interleaved (re,im) IEEE fp64 = numpy complex128,
 *    C-contiguous, exactly the arrays the reference passes around.
 *  - `stream` is a hipStream_t passed as void* (NULL = default stream).
 *  - Return value: 0 = ok, HPX_EINVAL bad argument, HPX_EHIP HIP runtime
 *    error, HPX_ENOTPD non-positive pivot in some baseline's factorisation
 *    (reported after the run; see hpx_plan_info), HPX_ETIMEOUT a hand-off
 *    between the workgroups of a split factorisation timed out (a statement
 *    about the device being shared, not about the matrix).  Nothing throws.
 *    hpx_last_error() returns a thread-local message for the last failure.
 *  - State lives in the plan.  Process-wide: the library options of hpx_set_option(NULL, ..) and, per
 *    device, the books of the split factorisations in flight (thread-safe).  One host thread per plan;
 *    different plans may be driven from different threads / streams.
 */
#ifndef HPX_H
#define HPX_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define HPX_OK      0
#define HPX_EINVAL (-1)
#define HPX_EHIP   (-2)
#define HPX_ENOTPD (-3)
#define HPX_ETIMEOUT (-4)   /* the split factor's hand-off between workgroups timed out (see HPX_OPT_FACTOR_SPLIT) */

#define HPX_VERSION 100

typedef struct hpx_plan hpx_plan;

int hpx_version(void);
const char* hpx_last_error(void);

/* Number of visible HIP devices / select one for the calling thread
 * (one process per GPU: the launcher calls this once with LOCAL_RANK). */
int hpx_device_count(void);
int hpx_set_device(int dev);

/* ---- plan: one batch of `nbl` independent baselines of shape (T,N), M fg modes.
 * Replaces the per-baseline argument tuple of
 * hydra_pspec/pspec.py:493-507 (gibbs_sample_with_fg).  Allocates every
 * workspace up front; hpx_gibbs_run allocates nothing. */
int hpx_plan_create(hpx_plan** out, int nbl, int T, int N, int M);
/* The same with room for `extra_rhs` more right-hand-side columns per system besides the Ntimes
 * data columns (hpx_plan_set_static_dense_flagged carries one per flagged channel). */
int hpx_plan_create_ex(hpx_plan** out, int nbl, int T, int N, int M, int extra_rhs);
int hpx_plan_destroy(hpx_plan* p);
/* bytes of device memory held by the plan */
int64_t hpx_plan_bytes(const hpx_plan* p);

/* Static inputs (everything that does not change along the chain) and the
 * iteration-invariant operators derived from them
 * (C = U^H Ni U as its circulant generator, G = U^H Ni F, H = F^H Ni F, the
 * data/noise parts of the right-hand side; pspec.py:359-369, :220-222).
 *   vis      (nbl,T,N) c128   visibilities (NOT yet multiplied by flags; the
 *                             library applies vis*flags as pspec.py:613 does)
 *   flags    (nbl,N)   u8     1 = use channel, 0 = flagged (pspec.py:520-522)
 *   ninv     (nbl,N)   f64    diagonal of the inverse noise covariance
 *   fgmodes  (nbl|1,N,M) c128 foreground modes; fg_shared!=0 => one set for all
 *   prior_map (nbl|1,N) i32   for each channel the row of `xgrid` holding its
 *                             prior grid, or -1 for "no prior" (the reference's
 *                             test any(prior[:,i] > 0), pspec.py:114)
 *   xgrid    (nxrows,ngrid) f64 logspace(log10 lo, log10 hi, ngrid) per distinct
 *                             prior box (pspec.py:50); ngrid is 1000 in the
 *                             reference.  nxrows may be 0 (no priors).
 *   omega    (T,4,N)   f64    the reference's per-time normal draws
 *                             omi,omj,omk,oml (pspec.py:196-217), identical for
 *                             every baseline and iteration; NULL = map estimate
 *                             (pspec.py:210-212)
 *   fop      (N,N)     c128   utils.fourier_operator(N) (utils.py:15-41)
 *   any_flags                 0 if no baseline has a flagged channel (skips the
 *                             masked transform of the log-posterior) */
int hpx_plan_set_static(hpx_plan* p, const double* vis, const uint8_t* flags,
                        const double* ninv, const double* fgmodes, int fg_shared,
                        const int32_t* prior_map, const double* xgrid, int nxrows,
                        int prior_shared, int ngrid, const double* omega,
                        const double* fop, int any_flags, void* stream);

/* The same for a dense (non-diagonal) Hermitian inverse noise covariance, what the reference driver
 * passes for a correlated --noise_cov (run-hydra-pspec.py:427-438, pspec.py:361-369):
 *   ninv_dense (nbl|1,N,N) c128   Ninv = inv(noise_cov), Hermitian
 *   nih_dense  (nbl|1,N,N) c128   its principal square root sqrtm(Ninv) (pspec.py:362)
 *   noise_shared                  != 0: one pair of matrices for all baselines
 * C = U^H Ninv U is then a general Hermitian matrix (not a circulant): every iteration lays the
 * whole augmented system out and factors it in place; the first ln-posterior term is the full
 * quadratic form r^H Ninv r (pspec.py:472-477), chi^2 uses Ninv.diagonal() (pspec.py:452).
 * Needs unflagged data (any_flags == 0; with flags: hpx_plan_set_static_dense_flagged below): the reference's column-masked Ni = Ninv diag(w) is not
 * Hermitian (pspec.py:361 FIXME) and is refused with HPX_EINVAL. */
int hpx_plan_set_static_dense(hpx_plan* p, const double* vis, const uint8_t* flags,
                              const double* ninv_dense, const double* nih_dense, int noise_shared,
                              const double* fgmodes, int fg_shared, const int32_t* prior_map,
                              const double* xgrid, int nxrows, int prior_shared, int ngrid,
                              const double* omega, const double* fop, int any_flags, void* stream);

/* A dense Hermitian inverse noise covariance TOGETHER WITH flagged channels.  The reference masks the
 * COLUMNS of Ninv, Ni = flags.T * Ninv * flags = Ninv diag(w) (pspec.py:361), so its system
 * A = [[1 + S Ni, S Ni F], [F^H Ni, F^H Ni F]] (pspec.py:365-369) is not Hermitian; it is, however, a rank-f
 * update of the Hermitian one (f = number of flagged channels): with B = [U F], E the unit vectors of the
 * flagged channels,  B^H Ni B = B^H Ninv B - (B^H Ninv E)(E^T B), both factors iteration-invariant.  The
 * unflagged-noise system is factored as in hpx_plan_set_static_dense with f more right-hand sides (one
 * per flagged channel; the plan must come from hpx_plan_create_ex(.., extra_rhs >= the largest flag count))
 * and the solution follows from the Woodbury identity: an f x f system solved with partial pivoting.
 *   ninv_dense (nbl|1,N,N) c128   Ninv = inv(noise_cov), Hermitian, NOT masked
 *   nih_masked (nbl,N,N)   c128   sqrtm(Ni) of the column-masked Ni, as the reference takes it
 *                                 (pspec.py:362; a general matrix: scipy.linalg.sqrtm on the host)
 * The first ln-posterior term is the quadratic form over the unflagged channels with
 * Ninv[flags][:, flags] (pspec.py:472-477), chi^2 uses the unmasked Ninv.diagonal() (pspec.py:452). */
int hpx_plan_set_static_dense_flagged(hpx_plan* p, const double* vis, const uint8_t* flags,
                                      const double* ninv_dense, int noise_shared, const double* nih_masked,
                                      const double* fgmodes, int fg_shared, const int32_t* prior_map,
                                      const double* xgrid, int nxrows, int prior_shared, int ngrid,
                                      const double* omega, const double* fop, void* stream);

/* Time-dependent flags and noise: flags_t (nbl,T,N) u8 and ninv_t (nbl,T,N) f64 (diagonal inverse
 * noise variances per time sample).  The mode the reference documents but does not implement
 * (docstrings pspec.py:337-340, :398-401; FIXMEs :361, :450-451; its driver reduces per-time flags to
 * an any-time mask instead, run-hydra-pspec.py:524-541): every time sample then has its own noise
 * matrix Ni_t = diag(ninv_t w_t), hence its own system -- nbl*T factorisations per iteration, the
 * "(baseline x time, Nfreq, Nfreq)" batch.  Per time: data w_t d_t, chi^2 with ninv_t, the
 * ln-posterior's noise term over the channels unflagged at t and its signal term through the
 * flags_t x flags_t sub-block of inv(S).  With flags_t[b][t] = flags[b] and ninv_t[b][t] = ninv[b] for
 * every t the chains are those of hpx_plan_set_static.  Dense solver only. */
int hpx_plan_set_static_pertime(hpx_plan* p, const double* vis, const uint8_t* flags_t,
                                const double* ninv_t, const double* fgmodes, int fg_shared,
                                const int32_t* prior_map, const double* xgrid, int nxrows,
                                int prior_shared, int ngrid, const double* omega,
                                const double* fop, int any_flags, void* stream);

/* The same with a full (non-diagonal) inverse noise covariance per time sample: the Ninv of shape
 * (Ntimes, Nfreqs, Nfreqs) of the reference's docstrings (pspec.py:337-340, :398-401).  Every
 * (baseline, time) unit is then a dense-noise system as in hpx_plan_set_static_dense; units with flagged
 * channels take the Woodbury correction of hpx_plan_set_static_dense_flagged (one extra right-hand side
 * per flagged channel of the unit, found here from flags_t).
 *   ninv_t_dense (nbl,T,N,N) c128   Ninv_{b,t}, Hermitian, NOT masked
 *   nih_t        (nbl,T,N,N) c128   sqrtm of the column-masked Ni_{b,t} = Ninv_{b,t} diag(w_{b,t}) (pspec.py:361-362)
 * chi^2 uses Ninv_{b,t}.diagonal(), the first ln-posterior term is sum_t r_t^H Ninv_{b,t}[w_t][:, w_t] r_t.
 * With ninv_t_dense[b][t] = Ninv[b] and flags_t[b][t] = flags[b] for every t the chains are those of
 * hpx_plan_set_static_dense(_flagged).  Dense solver only. */
int hpx_plan_set_static_pertime_dense(hpx_plan* p, const double* vis, const uint8_t* flags_t,
                                      const double* ninv_t_dense, const double* nih_t, const double* fgmodes,
                                      int fg_shared, const int32_t* prior_map, const double* xgrid, int nxrows,
                                      int prior_shared, int ngrid, const double* omega, const double* fop,
                                      int any_flags, void* stream);

/* Random tables of the bandpower draw (pspec.py:113-125): one uniform per
 * channel per iteration from the chain's global stream.
 *   uniforms (niter,N) f64   U
 *   igy      (niter,N) f64   1/gammainccinv(T-1, U)  (= invgamma.ppf(U, a=T-1))
 * Both shared by all baselines of the plan (the reference driver passes the same
 * seed for every baseline, run-hydra-pspec.py:547).  Copied on `stream`; complete on return. */
int hpx_plan_set_rng(hpx_plan* p, const double* uniforms, const double* igy, int niter,
                     void* stream);

/* Run `niter` Gibbs iterations for all baselines, starting at table row
 * `iter0` (pspec.py:606-623; one iteration = pspec.py:377-490).
 *   ps0       (nbl,N)       bandpowers defining the initial covariance
 *                           S = F^H diag(ps0/N^2) F; NULL continues from the
 *                           state the plan's previous run left (-1 if there
 *                           is none)
 *   ps_forced (nbl,niter,N) optional (NULL): teacher forcing -- iteration i+1
 *                           uses ps_forced[:,i] instead of its own draw
 *   ps_out    (nbl,niter,N) f64   always
 *   lnpost_out(nbl,niter)   f64   always
 *   cr_out    (nbl,nkeep,T,N) c128, fg_out (nbl,nkeep,T,M) c128,
 *   chisq_out (nbl,nkeep,T,N) f64: optional (NULL); iteration i is stored at
 *                           slot i/thin when i % thin == 0 (nkeep=ceil(niter/thin))
 *   ps_last   (nbl,N)       optional: the state to resume from (= last draw or
 *                           last forced value) */
int hpx_gibbs_run(hpx_plan* p, const double* ps0, int iter0, int niter,
                  const double* ps_forced, double* ps_out, double* lnpost_out,
                  double* cr_out, double* fg_out, double* chisq_out, int thin,
                  double* ps_last, void* stream);

/* Iteration 0 for an initial covariance that is NOT of the form
 * F^H diag(.) F (pspec.py:599 accepts any matrix): the caller supplies
 * Sh' = U^H sqrtm(S_initial) U, (nbl,N,N) c128.  Runs exactly one iteration
 * (table row iter0) with outputs as hpx_gibbs_run(niter=1).  Also for plans with time-dependent flags /
 * noise, diagonal or a full matrix per time (one explicit system per baseline and time; pspec.py:442 accepts any
 * matrix there too). */
int hpx_gibbs_step_general(hpx_plan* p, const double* shp, int iter0,
                           double* ps_out, double* lnpost_out, double* cr_out,
                           double* fg_out, double* chisq_out, double* ps_last,
                           void* stream);

/* per-baseline factorisation status of the last run: (nbl,) int32 host array,
 * 0 = ok, k>0 = non-positive pivot first seen at iteration k-1+iter0; with bit 30
 * (0x40000000) set: the split factor's hand-off timed out at iteration
 * (k & ~0x40000000)-1+iter0 (hpx_gibbs_run then returns HPX_ETIMEOUT).  The `info`
 * arrays of hpx_zpotrf_batched / hpx_zpotrs_batched use the same encoding with k = 1. */
int hpx_plan_info(hpx_plan* p, int32_t* info_host);

/* Options of one plan (p != NULL) or of the library (p == NULL: process-wide, set before the first use
 * of what they steer).  Returns HPX_EINVAL for an unknown key.
 *   HPX_OPT_FACTOR_SPLIT (plan or library; default 1): batches of fewer systems than half the CUs may take the
 *     SPLIT factorisation (several co-operating workgroups per system).  Its workgroups wait for each other, so
 *     all of them must be resident together: the library keeps count of its own split launches per device and
 *     stream, but cannot see another PROCESS on the same GPU -- a caller that shares a GPU between processes
 *     (several ranks per device) sets 0.  Also: the split form orders its additions differently, so a baseline's
 *     chain inside a small batch and inside a large one agree to rounding, not bit for bit; 0 gives one order of
 *     operations at every batch size (what a driver that re-shards between runs wants).
 *   HPX_OPT_SPLIT_HEAVY (library, testing; default 0): agent-scope fences in every hand-off of the split form
 *     (the protocol it falls back to when the parts of a system do not share an XCD).
 *   HPX_OPT_SPLIT_SPIN_LIMIT (library, testing; default 1<<22): polls before a hand-off gives up.
 *   HPX_OPT_EIGH_INNER_SWEEPS (library; default 1), HPX_OPT_EIGH_TRACE (library; default 0): hpx_zheev_psd_batched. */
#define HPX_OPT_FACTOR_SPLIT 1
#define HPX_OPT_SPLIT_HEAVY 2
#define HPX_OPT_SPLIT_SPIN_LIMIT 3
#define HPX_OPT_EIGH_INNER_SWEEPS 4
#define HPX_OPT_EIGH_TRACE 5
#define HPX_OPT_SPLIT_RETRY 6       /* plan; default 1: hpx_gibbs_run repeats a run whose split factor timed out, once, on the
                                     * one-workgroup kernel (the plan's HPX_OPT_FACTOR_SPLIT is 0 from then on) instead of
                                     * returning HPX_ETIMEOUT */
#define HPX_OPT_SPLIT_FALLBACKS 7   /* plan, read-only (hpx_get_option): how often that happened */
int hpx_set_option(hpx_plan* p, int key, int value);
/* The current value of a plan option (HPX_OPT_FACTOR_SPLIT, HPX_OPT_SPLIT_RETRY, HPX_OPT_SPLIT_FALLBACKS). */
int hpx_get_option(const hpx_plan* plan, int key, int* value);

/* Solver of the per-iteration linear system.  HPX_SOLVER_DENSE (default): batched Cholesky of
 * the (N+M) system (k_factor / k_backsolve), any diagonal Ninv and flags.  HPX_SOLVER_FLAT: for
 * unflagged baselines whose inverse noise variance is the same in every channel the system is
 * diagonal plus a rank-M border and is solved exactly through the M x M Schur complement
 * (hpx_flat.hip), O(N M (M+T)) instead of O(N^3); -1 if the plan's inputs do not qualify
 * (flags present, non-flat Ninv, M > 16, T > 256).  Both replace build_matrices + the
 * preconditioned CG of gcr_fgmodes_1d (pspec.py:325-374, :228) and agree to rounding. */
#define HPX_SOLVER_DENSE 0
#define HPX_SOLVER_FLAT 1
/* HPX_SOLVER_LOWRANK: flagged baselines whose UNFLAGGED channels share one inverse noise
 * variance (the reference driver's default noise model, run-hydra-pspec.py:436-438, with data
 * flags): C = c I - c Vf Vf^H with one column of Vf per flagged channel, so the system is
 * diagonal plus a border of width M + f and is solved exactly through the (M+f) x (M+f) Schur
 * complement (hpx_lowrank.hip; the small dense system reuses the batched Cholesky);
 * -1 unless every baseline qualifies and M + max f <= 240, T <= 256.  For power-of-two N and
 * M <= 16 the products with Vf are taken as length-N transforms evaluated at the flagged
 * channels (the border is never formed, O(N log N (M + T)) per iteration); otherwise the border is
 * laid out once per plan and contracted on the MFMA. */
#define HPX_SOLVER_LOWRANK 2
#define HPX_SOLVER_LOWRANK_DIRECT 3   /* as 2, but always the explicit-border (MFMA) form, never the FFT form */
int hpx_plan_set_solver(hpx_plan* p, int mode);

/* time (ms) spent in each stage of the last hpx_gibbs_run, measured with HIP
 * events on the run's stream; host array of HPX_NSTAGE floats
 * [assemble, factor, backsolve, transform, residual, draw]; needs
 * hpx_plan_set_profiling(p,1) before the run (adds event records only).
 * For power-of-two N <= 512, and for N <= 256 without an FFT, the back transform and
 * the residual are one kernel, booked under "transform" ("residual" then only holds
 * the masked transform of flagged data); the structured solvers book their whole
 * solve under "factor". */
#define HPX_NSTAGE 6
int hpx_plan_set_profiling(hpx_plan* p, int on);
int hpx_plan_stage_ms(hpx_plan* p, float* ms_host);

/* ---- unit-testable stages (same kernels the run uses) ------------------- */

/* Augmented system for the current bandpowers `ps` (nbl,N): lower triangle of
 * M = A^-1 K' A^-1 = [[C + diag(N/ps), G],[G^H, H]]  (K' = [[I + D^1/2 C D^1/2, D^1/2 G],
 * [G^H D^1/2, H]], A = diag(D^1/2, I): the symmetric scaling the device solves in, DESIGN.md
 * section 2) plus the T right-hand-side rows A^-1 r', written to the plan's factor buffer and
 * copied to `k_out`
 * (nbl, npad+Tpad, npad) c128 row-major (upper triangle zero) if non-NULL.
 * Replaces build_matrices + the RHS of gcr_fgmodes_1d (pspec.py:325-374,
 * :220-222) in Hermitian form. */
int hpx_assemble_K(hpx_plan* p, const double* ps, double* k_out, void* stream);
int hpx_plan_dims(const hpx_plan* p, int* npad, int* tpad, int* ld);

/* Batched complex Hermitian positive-definite factor / solve, stand-alone:
 *   a (nb,n,n) c128 row-major, lower triangle referenced; l_out (nb,n,n) c128
 *   lower-triangular L with A = L L^H (upper zero).  info (nb,) int32 device.
 * hpx_zpotrs_batched: solves A X = B for b (nb,n,nrhs) c128 given the same A
 * (factors internally).  Replaces the pinv-preconditioned CG of
 * pspec.py:228 / :372. */
int hpx_zpotrf_batched(int nb, int n, const double* a, double* l_out, int32_t* info,
                       void* stream);
int hpx_zpotrs_batched(int nb, int n, int nrhs, const double* a, const double* b,
                       double* x_out, int32_t* info, void* stream);

/* Which of the three forms of the batched factorisation a batch of nb systems of order n with nrhs right-hand
 * sides takes on the current device when nothing else runs there (diagnostic; bench.py names the kernel it prices
 * with it): 0 = the 32-wide kernel (k_factor), 1 = the wide kernel (k_factor_wide, 128-column super-blocks, from
 * order 400 on), 2 = the split kernel (k_factor_split: *parts co-operating workgroups per system; small batches). */
int hpx_factor_form(int nb, int n, int nrhs, int* parts);

/* Batched centred DFT along the channel axis: out[b,t,:] = F in[b,t,:]
 * (inverse!=0: F^H in / N), (nb,T,N) c128.  fop as in hpx_plan_set_static.
 * Replaces sample_S's fftshift(fft(ifftshift())) (pspec.py:92-95) and
 * fourier_operator products (pspec.py:321, :464). */
int hpx_dft_batched(int nb, int T, int N, const double* fop, const double* in,
                    double* out, int inverse, void* stream);

/* Truncated inverse-gamma draw by CDF inversion (pspec.py:11-64) for `n`
 * independent (alpha, beta, u, xgrid row) tuples; alpha must be a positive
 * integer (the path always calls it with alpha = Ntimes). */
int hpx_invgamma_inversion(int n, int alpha, const double* beta, const double* u,
                           const double* xgrid, int ngrid, double* out, void* stream);

/* DPSS weighted fit in closed form (hydra_pspec/dpss.py:7-94): for each of nb
 * spectra d (nb,N) c128 with weights tw (nb,N) f64 (= taper*w), basis
 * modes (nm,N) f64 and inverse covariance icov (N,N) c128 (Hermitian part is
 * used), returns amps (nb, 2*nm) f64 interleaved (re,im).  nm <= 32. */
int hpx_dpss_project(int nb, int N, int nm, const double* d, const double* tw,
                     const double* modes, const double* icov, double* amps,
                     void* stream);
/* The (baseline x time) batch: `ngroups` groups of `per` spectra that share their weights
 * (d (ngroups*per, N) c128, tw (ngroups, N)).  Per group the weighted normal matrix
 * A = Mw^H icov Mw, its inverse and the projector P = Mw^H icov diag(tw) are formed once (dense
 * product on the f64 MFMA); per spectrum the tall-skinny projection P d and the nm x nm multiply run
 * in one MFMA kernel that reads the visibility cube once.  `work`: device workspace of at least
 * hpx_dpss_workspace_bytes(...) bytes -- the call is then fully asynchronous on `stream` and
 * allocates nothing; NULL: allocated and released inside the call (which then synchronises and
 * reports a non-positive-definite normal matrix as HPX_ENOTPD).  reuse_projector != 0: the per-group
 * stage is skipped and the projector the previous call left in the SAME caller workspace is used
 * (new data, unchanged weights: tw / modes / icov may then be NULL).
 * A group whose weighted normal matrix is not positive definite (a fully flagged baseline, fewer
 * unflagged channels than modes) gets ZERO amplitudes (what the reference's L-BFGS fit from a zero start
 * returns for a fully flagged spectrum, dpss.py:81-92) and is marked in the workspace;
 * hpx_dpss_group_info copies the marks of the last per-group stage to the host (info_host (ngroups,) i32,
 * 1 = singular; synchronises `stream`) -- the way to learn of it with a caller workspace. */
int64_t hpx_dpss_workspace_bytes(int ngroups, int per, int N, int nm);
int hpx_dpss_group_info(const void* work, int64_t work_bytes, int ngroups, int per, int N, int nm,
                        int32_t* info_host, void* stream);
int hpx_dpss_project_grouped(int ngroups, int per, int N, int nm, const double* d,
                             const double* tw, const double* modes, const double* icov,
                             double* amps, void* work, int64_t work_bytes, int reuse_projector,
                             void* stream);

/* OQE (hydra_pspec/oqe.py).  Every O(s^3) piece is a dense product with the DFT matrix
 * M[a][j] = exp(-2 pi i a j / s) (oqe.py:7-10) or the weighting R on the f64 MFMA; Q_tau is rank
 * one, so the trace formulas collapse (hpx_extra.hip).  `work` / `work_bytes`: device workspace of at
 * least hpx_oqe_workspace_bytes(nb, s, nvis) bytes -- the call then allocates nothing and does not
 * synchronise; NULL: allocated and released inside the call (which then synchronises).
 *   hpx_oqe_fisher: F_ab = 1/2 tr(R* Q_a R Q_b) (oqe.py:43-50, variant 0) or Ft (oqe.py:53-66,
 *     variant 1) for nb weightings R (nb,s,s) c128 -> (nb,s,s) c128.
 *   hpx_oqe_qh: q_h (oqe.py:104-114) for V (nb,2P,s) -> (nb,P,s). */
int64_t hpx_oqe_workspace_bytes(int nb, int s, int nvis);
int hpx_oqe_fisher(int nb, int s, const double* R, double* F_out, int variant,
                   void* work, int64_t work_bytes, void* stream);
int hpx_oqe_qh(int nb, int npair, int s, const double* R, const double* V,
               double* q_out, void* work, int64_t work_bytes, void* stream);
/* diag(M (R C R') M^H) (nb,s) c128 with R' = conj(R) (conj_right != 0) or R: the quantity behind
 * oqe.bias (R C conj R; oqe.py:23-24) and Sig_QEN / Sig_QESN (R C R; oqe.py:161-186). */
int hpx_oqe_sandwich_diag(int nb, int s, const double* R, const double* C, int conj_right,
                          double* out, void* work, int64_t work_bytes, void* stream);
/* M_opt (oqe.py:77-84): diag(1/F_aa) with row a divided by sum_b (M F)_ab; (nb,s,s) c128 -> same. */
int hpx_oqe_mopt(int nb, int s, const double* F, double* M_out, void* stream);
/* beta[b][k] = sum_t |sk[b][t][k]|^2: the statistic of the bandpower draw (sample_S, pspec.py:96-100);
 * sk (nb,T,N) c128 -> out (nb,N) f64. */
int hpx_power_sum(int nb, int T, int N, const double* sk, double* out, void* stream);
/* out = a x + b y over n doubles (a utility for callers that iterate on device matrices; oqe.M_Fhalf itself
 * is one call of hpx_sqrtm_hpd_batched since round 5). */
int hpx_lincomb(int64_t n, double a, const double* x, double b, const double* y, double* out,
                void* stream);

/* Foreground modes from the data: for each baseline the `nmodes` leading
 * eigenvectors (unit norm, largest component real positive, eigenvalues descending) of the
 * frequency-frequency covariance over time np.cov(vis_b.T) -- what
 * scripts/calc-vis-cov-matrices.py:235-249 writes and run-hydra-pspec.py:453 truncates to
 * Nfgmodes columns.  vis (nb,T,N) c128; modes (nb,N,nmodes) c128; evals (nb,nmodes) f64.
 * min(T,N) <= 1024 (T <= N goes through the T x T Gram matrix; orders from 128 on through
 * hpx_zheev_psd_batched's solver). */
int hpx_fgmodes_eig(int nb, int T, int N, int nmodes, const double* vis, double* modes,
                    double* evals, void* stream);

/* Auto-correlation estimator without the noise bias: q[v][tau] = 1/2 x_v^H (R^* Q_tau R) x_v
 * = 1/2 conj(FFT(R^T x_v))[tau] FFT(R x_v)[tau] for every visibility x_v of V (nb,nvis,s) c128;
 * R (nb,s,s) c128, q_out (nb,nvis,s) c128.  Replaces the double loop of oqe.q / oqe.qhat
 * (oqe.py:27-30, :88-101); the caller subtracts bias[tau]. */
int hpx_oqe_qauto(int nb, int nvis, int s, const double* R, const double* V, double* q_out,
                  void* work, int64_t work_bytes, void* stream);

/* Eigendecomposition of nb Hermitian positive semi-definite matrices, orders up to 2048 (csrc/hpx_eigh.hip: blocked
 * one-sided Jacobi on the Cholesky factor, 16 x 16 rotations on the FP64 MFMA) -- the solver behind hpx_fgmodes_eig
 * from order 128 on (the covariance the reference diagonalises with numpy.linalg.eigh,
 * scripts/calc-vis-cov-matrices.py:239-247).  a (nb,n0,n0) c128; with n = hpx_zheev_psd_order(n0)
 * (n0 rounded up to a multiple of 16, of 32 from 241 on): w (nb,n) f64 eigenvalues and v (nb,n0,n) c128 unit
 * eigenvectors as columns, unsorted; the n - n0 pairs of the zero padding have eigenvalue 0.
 * sweeps_out (host int, optional).  HPX_EINVAL (and *sweeps_out = -1) if the sweeps do not converge within
 * the limit (30): no unconverged pair is handed out. */
int hpx_zheev_psd_batched(int nb, int n0, const double* a, double* w, double* v, int* sweeps_out, void* stream);
int hpx_zheev_psd_order(int n0);

/* Principal square root and inverse square root of nb Hermitian positive-definite matrices on the device
 * (coupled Newton-Schulz iteration on the batched FP64-MFMA product; csrc/hpx_sqrtm.hip).  Set-up of the
 * correlated-noise paths: replaces the per-baseline host calls scipy.linalg.sqrtm(Ni) / eigh of reference
 * pspec.py:361-362 (for flagged channels through Ni = P [[A, 0], [B, 0]] P^T -> P [[A^1/2, 0], [B A^-1/2, 0]] P^T).
 * a (nb,n,n) c128 row-major, n a multiple of 16 (pad with an identity block); sq, isq (nb,n,n) c128, either may be
 * NULL; stops when ||I - Z Y||_F < tol * n (tol ~ 1e-7: the step taken after that squares it, and one more
 * polishing step follows), HPX_EINVAL if that
 * takes more than max_iter steps or a matrix is not positive definite; iters_out (host int, optional). */
int hpx_sqrtm_hpd_batched(int nb, int n, const double* a, double* sq, double* isq, double tol,
                          int max_iter, int* iters_out, void* stream);
/* The reference's sqrtm(Ni) of the column-masked Ni = Ninv diag(w) (pspec.py:361-362), for nb Hermitian
 * positive-definite Ninv and channel masks, entirely on the device: with the unflagged channels u first
 * Ni = P [[A, 0], [B, 0]] P^T has the principal root P [[A^1/2, 0], [B A^-1/2, 0]] P^T.
 *   a (nb | 1, n, n) c128   Ninv (shared != 0: one matrix for all systems)
 *   w (nb, n) u8            1 = use the channel, 0 = flagged
 *   out (nb, n, n) c128     the root, in channel order
 * Works in chunks of at most 256 systems (workspace 100 n^2 bytes per system of a chunk); tol / max_iter /
 * iters_out as in hpx_sqrtm_hpd_batched. */
int hpx_sqrtm_masked_batched(int nb, int n, const double* a, int shared, const uint8_t* w, double* out,
                             double tol, int max_iter, int* iters_out, void* stream);

/* Empirical lane map of v_mfma_f64_16x16x4_f64 (diagnostic used by the tests):
 * computes D = A(16x4) * B(4x16) and writes, for lane l and register v, the
 * value D holds; host (64*4) doubles. */
int hpx_mfma_probe(const double* a_host, const double* b_host, double* d_host);

/* Measured FP64-MFMA issue peak of the current device: every CU runs `iters`
 * x 4 back-to-back independent v_mfma_f64_16x16x4_f64 per wave (2 waves per
 * SIMD); returns TFLOP/s (host double).  Used by bench.py for the roofline. */
int hpx_mfma_f64_peak(int iters, double* tflops_host);

#ifdef __cplusplus
}
#endif
#endif
