// Batched complex-Hermitian Cholesky, split form: SEVERAL workgroups per system, for batches too small to give
// every CU a system of its own (BASELINE config 2: 64 systems of order 272 on 256 CUs).
//
// With one workgroup per system such a batch runs on a quarter of the chip and each system's 17 tile columns go
// through one CU's latency chain (elimination -> tiles below -> trailing update) one after the other.  Here a
// system's 16 x 16 tiles are dealt over `parts` workgroups and stay in REGISTERS for the whole factorisation
// (right-looking): tile (r, c) belongs to workgroup r mod parts; there the diagonal tiles belong to wave 0 (the
// ELIMINATOR, a code path of its own) and the others to the seven WORKER waves, wave 1 + (r / parts + c) mod 7 -- a
// column's tiles of one workgroup sit in different waves.  Per tile column j
//   E  the eliminator that owns the diagonal tile runs the fused 16 x 16 Cholesky + inverse on its own (elim16w:
//      one wave, the tile in registers, no barrier), stores L_jj and
//      inv(L_jj) (Vt) and releases flag A[j];
//   B  every workgroup takes inv(L_jj) (from LDS if it is the owner, else from Vt once A[j] is up), forms its
//      tiles of the column, X(r, j) = D(r, j) inv(L_jj)^H, stores them into the
//      factor and counts itself in on B[j];
//   D  once B[j] shows all parts, the column (rows j+1 .. nrt-1) is copied into LDS once per workgroup (LDS-DMA; the
//      workers meet on a counter in LDS for this, the eliminator is not held up;
//      the workgroup's own tiles were put there by B) and every wave updates its trailing tiles
//      D(r, c) -= X(r, j) X(c, j)^H.
// Look-ahead: the owner of diagonal tile j+1 also owns X(j+1, j), the only operand that tile's last update
// needs, so it runs E(j+1) BEFORE its own D(j) -- the chain from one elimination to the next is one flag
// hand-off plus one tile product, and the trailing updates of all parts run beside it.
//
// Hand-off: release / acquire at agent scope on counters in the system's Vt block (HPX_VT_SYNC), no grid
// sync; the parts of a system are placed on one XCD (block index mod 8) so the data they exchange stays in one
// L2.  The LAST part to finish zeroes the system's counters again.
// Residency: a part spins on the other parts of its system, so every part of every split launch in flight on the
// device must be resident together (one workgroup of this kernel fills a CU's register file).  The launcher keeps
// the books per device (SplitGuard below): the workgroups of the split launches still in flight on OTHER streams plus
// those of the new launch must not exceed the CUs, else the new launch takes fewer parts or the one-workgroup kernel
// (hpx_launch_factor).  What the books cannot see -- another PROCESS running split launches on the same GPU -- is
// covered by the time-out: a spin gives up after `spin_limit` polls and flags the system in `info` with
// HPX_INFO_TIMEOUT, which finish_run reports as HPX_ETIMEOUT (not as a non-positive pivot); callers that share a
// GPU between processes switch the form off (HPX_OPT_FACTOR_SPLIT).
#define HPX_TILES_STAGE0_D 512            // (only the tile hand-over area of the first staging buffer is used here)
#include "hpx_factor_tiles.h"
#include <mutex>

namespace {

constexpr int SPLIT_NW = 8;             // waves per workgroup: the eliminator and seven workers
constexpr int SPLIT_NSD = 5;            // diagonal tiles the eliminator can hold
constexpr int SPLIT_NSW = 9;            // tiles a worker wave can hold
constexpr int SPLIT_MAX_CT = 40;        // tile columns the counter block holds
constexpr unsigned SPIN_LIMIT_DEFAULT = 1u << 22;
static_assert(2 * SPLIT_MAX_CT + 2 <= 2 * HPX_VT_SYNC, "counters must fit the Vt block's sync area");

// ---- hand-off between workgroups.  `heavy`: release / acquire fences at agent scope -- on this part a write-back
// and an invalidate of the XCD's L2 (the L2s of different XCDs are not coherent with each other), microseconds
// each and serialised between the CUs of an XCD.  Light: the parts of a system share one L2 (they sit on one XCD:
// checked at run time, see xmask below), so all that is needed is that the producer's stores have reached L2
// (vmcnt 0: the vector L1 is write-through) and that the consumer's L1 holds no stale line (buffer_inv sc0) --
// the protocol the memory model uses between the CUs of a workgroup in threadgroup-split mode.
__device__ HPX_INL void handoff_release(const bool heavy) {
  if (heavy) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
  else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}
__device__ HPX_INL void handoff_acquire(const bool heavy) {
  if (heavy) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  else asm volatile("buffer_inv sc0" ::: "memory");
}
// wave-uniform wait for *flag >= target: relaxed polls (they go to L2 and invalidate nothing), the acquire follows
__device__ HPX_INL bool spin_until(int* flag, const int target, const bool heavy, const unsigned spin_limit) {
  unsigned n = 0;
  bool ok = true;
  while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
    __builtin_amdgcn_s_sleep(1);
    if (++n > spin_limit) { ok = false; break; }
  }
  handoff_acquire(heavy);
  return ok;
}
// every wave's stores and LDS writes are complete, then the barrier
__device__ HPX_INL void full_barrier() {
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// The seven worker waves meet without the eliminator (s_barrier counts every wave of the workgroup, and the
// eliminator is busy with the next diagonal tile just then): a counter in LDS, one increment per wave and step.
__device__ HPX_INL void worker_barrier(int* cnt, const int target, const int lane) {
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");       // this wave's LDS-DMA and LDS writes have landed
  if (lane == 0) __hip_atomic_fetch_add(cnt, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  while (__hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < target) __builtin_amdgcn_s_sleep(0);
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

#ifdef HPX_SPLIT_TRACE
// timing of system 0's steps (a debugging build): [part][column + 1][stamp] in 10 ns ticks
__device__ long long hpx_split_trace[8 * (SPLIT_MAX_CT + 1) * 8];
#define HPX_STAMP(k_, t_) if (b == 0 && tid == (t_)) hpx_split_trace[(w * (SPLIT_MAX_CT + 1) + j + 1) * 8 + (k_)] = wall_clock64()
#else
#define HPX_STAMP(k_, t_)
#endif

// one trailing update D(r, c) -= X(r, j) X(c, j)^H, the column's tiles in LDS
__device__ HPX_INL void tile_update(d4& re, d4& im, const lds_f64* pa /* X(c, j) */, const lds_f64* pb /* X(r, j) */,
                                    const int rd_re, const int rd_im) {
#pragma unroll
  for (int v = 0; v < 4; ++v) {
    const double pr = pa[v * 128 + rd_re], pi = pa[v * 128 + rd_im];
    const double br = pb[v * 128 + rd_re], bm = pb[v * 128 + rd_im];
    re = mfma64(-pr, br, re);
    re = mfma64(-pi, bm, re);
    im = mfma64(-pr, bm, im);
    im = mfma64(pi, br, im);
  }
}

template <bool GEN>
__global__ __launch_bounds__(64 * SPLIT_NW, 2) void k_factor_split(double* __restrict__ L_all, double* __restrict__ Wre_all,
                                                                   double* __restrict__ Wim_all, double* __restrict__ Vt_all,
                                                                   int32_t* __restrict__ info, const int npad, const int ld,
                                                                   const int iter_tag, const hpx_gen_batch GB, const int nbl,
                                                                   const int parts, const int force_heavy,
                                                                   const unsigned spin_limit, int* __restrict__ xcd_miss) {
  extern __shared__ double lds_panel[];        // [nrt][512]: the current column's tiles (odd-column swizzle)
  __shared__ int wcount;                       // worker_barrier's counter
  const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
  const int b = (idx / parts) * 8 + xcd, w = idx % parts;
  if (b >= nbl) return;
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
  int* const flagA = (int*)(X.Vt + (long)npad * 32);
  int* const cntB = flagA + SPLIT_MAX_CT;
  int* const done = cntB + SPLIT_MAX_CT;
  int* const arrive = done + 1;                // parts that have started (low byte) and the XCDs they run on (one bit each)
  GenVec<false> V = {};
  if constexpr (GEN) {
    V.ia = (const glb_f64*)G.ia;
    V.cre = (const glb_f64*)G.cre;
    V.cim = (const glb_f64*)G.cim;
  }
  lds_f64* const Pn = (lds_f64*)lds_panel;
  lds_f64* const Vs = (lds_f64*)(hpx_stage1 + FV_OFF);
  // Which protocol: every part enters its XCD and counts itself in (`arrive`); when all have (no data is exchanged
  // yet, so no fence), a system whose parts share one XCD uses the light hand-off from the first column on.
  if (tid == 0) wcount = 0;
  if (tid == 0) {
    const unsigned xcc = __builtin_amdgcn_s_getreg((3 << 11) | 20) & 15u;        // HW_REG_XCC_ID, bits 3:0
    // (one word: XCD bits above the count -- the same thread's OR and ADD on one location stay in order)
    __hip_atomic_fetch_or(arrive, 256 << xcc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_fetch_add(arrive, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  bool heavy = true, bad = false, timed_out = false;
  // the lane's indices, re-derived through an empty asm in every step: otherwise each slot's LDS addresses become
  // loop invariants that outlive the registers (spilled, and reloaded in the middle of the hand-off chain)
  int lane = X.lane, li = lane & 15, g = lane >> 4;
  unsigned src_lane = 0;
  int rd_re = 0, rd_im = 0;
#define HPX_LANE_INDICES()                                   \
  lane = opaque(X.lane);                                     \
  li = lane & 15;                                            \
  g = lane >> 4;                                             \
  src_lane = g * 32 + 2 * ((li + 8 * (g & 1)) & 15);         \
  rd_re = g * 32 + li + 16 * (g & 1);                        \
  rd_im = g * 32 + li + 16 * (1 - (g & 1))

  auto agree = [&] {                           // (every wave polls for itself: no barrier, the tiles are loaded meanwhile)
    unsigned n = 0, word;
    while (((word = (unsigned)__hip_atomic_load(arrive, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) & 255u) < (unsigned)parts) {
      __builtin_amdgcn_s_sleep(1);
      if (++n > spin_limit) { timed_out = true; break; }
    }
    const unsigned m = word >> 8;
    const bool spread = (m & (m - 1)) != 0;
    heavy = force_heavy || spread || timed_out;
    // the parts do NOT share an XCD: this launch goes on with the agent-scope fences (correct, slow), and the host
    // is told -- the launcher stops choosing the split form on this device (a word in pinned host memory)
    if (spread && xcd_miss && tid == 0) __hip_atomic_store(xcd_miss, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  };
  if (X.wave == 0) {
    // =================== the eliminator: the workgroup's diagonal tiles (r = w + s parts) =====================
    d4 e1[SPLIT_NSD], e2[SPLIT_NSD];           // re, im of D^T[c][r]: lane li <-> row, register v <-> column g + 4 v
#pragma unroll
    for (int s = 0; s < SPLIT_NSD; ++s) {
      const int r = w + s * parts;
      e1[s] = (d4){0., 0., 0., 0.};
      e2[s] = e1[s];
      if (r < X.nct) tile_init<GEN, false, true>(G, V, X.Lb, r * 16, r * 16, npad, li, g, e1[s], e2[s]);
    }
    agree();
    // E: diagonal tile t, fully updated -> L_tt, inv(L_tt) (Vt, Vs), W; then flag A[t]
    auto eliminate = [&](const int t, const d4 er, const d4 ei) {
      // flag A[t] goes up as soon as the inverse tile is stored (this wave's own stores: nobody else wrote for it)
      bad |= elim16w(X, t, t + 1 == X.nct, er, ei, [&] {
        handoff_release(heavy);
        if (lane == 0) __hip_atomic_store(&flagA[t], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#ifdef HPX_SPLIT_TRACE
        if (b == 0 && tid == 0) hpx_split_trace[(w * (SPLIT_MAX_CT + 1) + t) * 8 + 4] = wall_clock64();   // (slot 4 of step t-1: published)
#endif
      });
      if (t & 1) {
        // odd tile of a pair: W10 = -inv(L11) L10 inv(L00); L10 = X(t, t-1) was stored by this workgroup in B(t-1),
        // inv(L00) by the owner of tile t-1 (flag A[t-1] acquired in step t-1); inv(L11) is in Vs (this wave's own
        // LDS writes)
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        d4 tre = {0., 0., 0., 0.}, tim = tre;
        const double* l10 = X.Lb + HPX_LIDX(t * 16 + li, t * 16 - 16 + g, X.npad);   // L10[r = li][k = 4 v + g]
        const double* v00 = X.Vt + (long)(t - 1) * 512 + li * 32 + g;                // inv(L00)[k = 4 v + g][c' = li]
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
    };
    // step j = -1 is the first elimination alone: E(0) is "the look-ahead of the column before the first"
    for (int j = -1; j < X.nct; ++j) {
      HPX_LANE_INDICES();
      HPX_STAMP(0, 0);
      if (j >= 0) {
        if (w != j % parts) {                  // the inverse of the diagonal tile -> Vs
          if (!spin_until(&flagA[j], 1, heavy, spin_limit)) timed_out = true;
          glds_tile(X.Vt + (long)j * 512, 8 * src_lane, lds_addr(hpx_stage1 + FV_OFF));
          wait_vm<0>();
        }
        HPX_STAMP(1, 0);
        lds_barrier();                         // (1) Vs is ready
        full_barrier();                        // (2) the workers' tiles of column j are stored, and in LDS
        HPX_STAMP(2, 0);
        if (j + 1 == X.nct) break;
      }
      // this workgroup's diagonal tiles right of column j take their update from its own X(r, j); the next
      // diagonal tile first, eliminated at once (look-ahead: the workers wait for the other parts meanwhile)
      if ((j + 1) % parts == w) {
        d4 er = {0., 0., 0., 0.}, ei = er;
#pragma unroll
        for (int s = 0; s < SPLIT_NSD; ++s)
          if (w + s * parts == j + 1) { er = e1[s]; ei = e2[s]; }
        if (j >= 0) tile_update(er, ei, Pn + (j + 1) * 512, Pn + (j + 1) * 512, rd_re, rd_im);
        eliminate(j + 1, er, ei);
      }
      HPX_STAMP(3, 0);
      if (j < 0) continue;
#pragma unroll
      for (int s = 0; s < SPLIT_NSD; ++s) {
        const int r = w + s * parts;
        if (r > j + 1 && r < X.nct) tile_update(e1[s], e2[s], Pn + r * 512, Pn + r * 512, rd_re, rd_im);
      }
      lds_barrier();                           // (3) the workers are through with the column and Vs
    }
  } else {
    // =================== the workers: tiles (r, c) below the diagonal and of the right-hand-side rows ==========
    // rows r = w + k parts; of row k the columns c = (wv - k) mod 7, + 7, ... (< min(r, nct)): a column's tiles of one
    // workgroup sit in different waves
    constexpr int NWK = SPLIT_NW - 1;
    const int wv = X.wave - 1;
    int tr[SPLIT_NSW], tc[SPLIT_NSW];
    d4 a1[SPLIT_NSW], a2[SPLIT_NSW];
    {
      int r = w, k = 0, c = wv;
#pragma unroll
      for (int s = 0; s < SPLIT_NSW; ++s) {
        while (r < X.nrt && c >= min(r, X.nct)) { r += parts; ++k; c = (wv + NWK * SPLIT_MAX_CT - k) % NWK; }
        if (r < X.nrt) {
          tr[s] = r;
          tc[s] = c;
          tile_init<GEN, false, true>(G, V, X.Lb, r * 16, c * 16, npad, li, g, a1[s], a2[s]);
          c += NWK;
          __builtin_amdgcn_sched_barrier(0);
        } else {
          tr[s] = 0;
          tc[s] = -1;
          a1[s] = (d4){0., 0., 0., 0.};
          a2[s] = a1[s];
        }
      }
    }
    agree();
    for (int j = 0; j < X.nct; ++j) {
      HPX_LANE_INDICES();
      lds_barrier();                           // (1) Vs is ready
      // ---- B: the wave's tiles of column j, X = D inv(L_jj)^H, into the factor and into the LDS column
#pragma unroll
      for (int s = 0; s < SPLIT_NSW; ++s)
        if (tc[s] == j) {
          d4 xr = {0., 0., 0., 0.}, xi = {0., 0., 0., 0.};
#pragma unroll
          for (int v = 0; v < 4; ++v) {
            const double pr = Vs[v * 128 + rd_re], pi = Vs[v * 128 + rd_im];
            xr = mfma64(pr, a1[s][v], xr);
            xr = mfma64(pi, a2[s][v], xr);
            xi = mfma64(pr, a2[s][v], xi);
            xi = mfma64(-pi, a1[s][v], xi);
          }
          double* o_ = X.Lb + HPX_LIDX(tr[s] * 16 + li, j * 16 + g, X.npad);
          lds_f64* xs = Pn + tr[s] * 512;
#pragma unroll
          for (int v = 0; v < 4; ++v) {
            o_[(4 * v) * 32] = xr[v];
            o_[(4 * v) * 32 + 16] = xi[v];
            const int k = g + 4 * v;                         // column of the tile; row li
            xs[k * 32 + li + 16 * (k & 1)] = xr[v];
            xs[k * 32 + li + 16 * (1 - (k & 1))] = xi[v];
          }
        }
      full_barrier();                          // (2) the workgroup's tiles of the column are stored
      if (tid == 64) {
        handoff_release(heavy);
        __hip_atomic_fetch_add(&cntB[j], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      if (j + 1 == X.nct) break;               // last column: nothing to its right
      // ---- D: the other parts' tiles of the column -> LDS, then the trailing updates
      if (!spin_until(&cntB[j], parts, heavy, spin_limit)) timed_out = true;
      HPX_STAMP(5, 64);
      {
        int n = 0;
        for (int r = j + 1; r < X.nrt; ++r) {
          if (r % parts == w) continue;
          if (n++ % NWK == wv)
            glds_tile(X.Lb + HPX_LIDX(r * 16, j * 16, X.npad), 8 * src_lane, lds_addr(lds_panel + r * 512));
        }
        wait_vm<0>();
      }
      HPX_STAMP(6, 64);
      worker_barrier(&wcount, (SPLIT_NW - 1) * (j + 1), lane);       // the column is staged (workers only)
#pragma unroll
      for (int s = 0; s < SPLIT_NSW; ++s) {
        if (tc[s] > j) tile_update(a1[s], a2[s], Pn + tc[s] * 512, Pn + tr[s] * 512, rd_re, rd_im);
        __builtin_amdgcn_sched_barrier(0);     // one tile's operands at a time: the accumulators need the registers
      }
      HPX_STAMP(7, 64);
      lds_barrier();                           // (3) the column and Vs are free again
    }
  }
#undef HPX_LANE_INDICES
  __syncthreads();
  if (tid == 0) {
    const int old = __hip_atomic_fetch_add(done, 1, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
    if (old == parts - 1) {                    // every part is past its last wait: leave the counters zero
      for (int i = 0; i < 2 * SPLIT_MAX_CT + 2; ++i) flagA[i] = 0;
    }
  }
  // a time-out outranks a pivot report: what the other parts computed from tiles that never arrived is meaningless
  if (timed_out && info) {
    atomicOr(&info[b], HPX_INFO_TIMEOUT);
    atomicCAS(&info[b], HPX_INFO_TIMEOUT, HPX_INFO_TIMEOUT | iter_tag);      // (the iteration, unless an earlier report holds one)
  } else if (bad && info) atomicCAS(&info[b], 0, iter_tag);
}

// do the tiles of a system fit the waves' registers with `parts` workgroups per system?
bool split_fits(const int parts, const int nct, const int nrt) {
  if ((nct + parts - 1) / parts > SPLIT_NSD) return false;
  for (int w = 0; w < parts; ++w)
    for (int v = 0; v < SPLIT_NW - 1; ++v) {
      int n = 0, k = 0;
      for (int r = w; r < nrt; r += parts, ++k)
        for (int c = (v + (SPLIT_NW - 1) * SPLIT_MAX_CT - k) % (SPLIT_NW - 1); c < (r < nct ? r : nct); c += SPLIT_NW - 1) ++n;
      if (n > SPLIT_NSW) return false;
    }
  return true;
}

// ---- the books of the split launches in flight, per device (see "Residency" at the top) -----------------------
// One stream is the common case and costs nothing: launches of one stream run one after the other, so only the
// newest counts and no event is needed.  From the moment a second stream takes the form, every split launch is
// followed by an event on its stream, and a launch counts as in flight until its event has completed.
struct SplitGuard {
  std::mutex mu;
  int cus = 0;
  int* xcd_miss = nullptr;          // pinned host word the kernel sets when the parts of a system sat on different XCDs
  bool multi = false;               // more than one stream has used the form
  hipStream_t solo = nullptr;       // (!multi) the one stream so far, and the workgroups of its newest launch
  bool have_solo = false;
  int solo_wgs = 0;
  struct Flight { hipStream_t st; hipEvent_t ev; int wgs; bool live; };
  std::vector<Flight> flights;      // (multi) one slot per stream seen
  int init(int dev) {
    if (cus) return HPX_OK;
    int n = 0;
    HPX_HIP(hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev));
    HPX_HIP(hipHostMalloc((void**)&xcd_miss, sizeof(int), hipHostMallocMapped));
    *xcd_miss = 0;
    cus = n;
    return HPX_OK;
  }
  Flight* slot(hipStream_t st) {
    for (auto& f : flights) if (f.st == st) return &f;
    Flight f = {st, nullptr, 0, false};
    if (hipEventCreateWithFlags(&f.ev, hipEventDisableTiming) != hipSuccess) return nullptr;
    flights.push_back(f);
    return &flights.back();
  }
  // workgroups of split launches that may still be running on streams other than `st`
  int busy_elsewhere(hipStream_t st) {
    int n = 0;
    for (auto& f : flights) {
      if (!f.live || f.st == st) continue;
      if (hipEventQuery(f.ev) == hipErrorNotReady) n += f.wgs;
      else f.live = false;
    }
    return n;
  }
};
SplitGuard g_guard[32];
int g_split_enabled = 1;                       // HPX_OPT_FACTOR_SPLIT at library level
int g_force_heavy = 0;                         // HPX_OPT_SPLIT_HEAVY (testing)
unsigned g_spin_limit = SPIN_LIMIT_DEFAULT;    // HPX_OPT_SPLIT_SPIN_LIMIT (testing)

// Workgroups per system for a batch of nbl systems with `free_cus` CUs to run on, or 0 when the split form does not
// apply: the batch must leave at least half of those CUs without a system, a wave's tiles must fit its registers and
// the column its LDS.
int split_parts_for(int nbl, int npad, int ld, int free_cus) {
  const int nct = npad >> 4, nrt = ld >> 4;
  if (free_cus <= 0 || nbl <= 0 || nct < 2 || nct > SPLIT_MAX_CT) return 0;
  if ((size_t)nrt * 4096 + (BUF_D + HPX_TILES_STAGE0_D) * sizeof(double) > (size_t)156 * 1024) return 0;
  const int live = 8 * ((nbl + 7) / 8);        // block indices are dealt in eights (one system's parts on one XCD)
  for (int parts = 8; parts >= 2; parts >>= 1)
    if (live * parts <= free_cus && split_fits(parts, nct, nrt)) return parts;
  return 0;
}

template <bool GEN>
int launch_split_t(int nbl, int parts, int npad, int ld, double* L, double* Wre, double* Wim, double* Vt, int32_t* info,
                   int iter_tag, const hpx_gen_batch& gen, int* xcd_miss, hipStream_t st) {
  static hpx_lds_limit limit;
  const size_t lds = (size_t)(ld >> 4) * 512 * sizeof(double);
  HPX_TRY(limit.ensure(reinterpret_cast<const void*>(&k_factor_split<GEN>), lds));
  const int grid = 8 * ((nbl + 7) / 8) * parts;
  hipLaunchKernelGGL((k_factor_split<GEN>), dim3(grid), dim3(64 * SPLIT_NW), lds, st, L, Wre, Wim, Vt, info, npad, ld, iter_tag,
                     gen, nbl, parts, g_force_heavy, g_spin_limit, xcd_miss);
  HPX_HIP(hipGetLastError());
  return HPX_OK;
}

}  // namespace

int hpx_split_set_option(int key, int value) {
  if (key == HPX_OPT_FACTOR_SPLIT) g_split_enabled = value != 0;
  else if (key == HPX_OPT_SPLIT_HEAVY) g_force_heavy = value != 0;
  else if (key == HPX_OPT_SPLIT_SPIN_LIMIT) g_spin_limit = value > 0 ? (unsigned)value : SPIN_LIMIT_DEFAULT;
  else return HPX_EINVAL;
  return HPX_OK;
}

// what hpx_launch_factor asks: may this batch take the split form now?  (no books touched: a dry query for tests)
int hpx_factor_split_parts(int nbl, int npad, int ld) {
#ifdef HPX_NO_SPLIT
  return 0;
#endif
  if (!g_split_enabled) return 0;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 32) return 0;
  SplitGuard& G = g_guard[dev];
  std::lock_guard<std::mutex> lk(G.mu);
  if (G.init(dev) != HPX_OK || *(volatile int*)G.xcd_miss) return 0;
  return split_parts_for(nbl, npad, ld, G.cus);
}

#ifdef HPX_SPLIT_TRACE
extern "C" int hpx_debug_split_trace(long long* out) {
  HPX_HIP(hipMemcpyFromSymbol(out, HIP_SYMBOL(hpx_split_trace), sizeof(long long) * 8 * (SPLIT_MAX_CT + 1) * 8));
  return HPX_OK;
}
#endif

// The split launch, or *took = 0 when the form does not apply to this batch or the device has no room for its
// parts beside the split launches in flight on other streams (the caller then takes a one-workgroup kernel).
int hpx_launch_factor_split(int nbl, int npad, int ld, double* L, double* Wre, double* Wim, double* Vt,
                            int32_t* info, int iter_tag, const hpx_gen_batch* gen, hipStream_t st, int* took) {
  *took = 0;
#ifdef HPX_NO_SPLIT
  return HPX_OK;
#endif
  if (!g_split_enabled) return HPX_OK;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 32) return HPX_OK;
  SplitGuard& G = g_guard[dev];
  std::lock_guard<std::mutex> lk(G.mu);           // held over the launch: the books and the queue stay in step
  HPX_TRY(G.init(dev));
  if (*(volatile int*)G.xcd_miss) return HPX_OK;  // the parts of a system did not share an XCD on this device: not this form
  if (!G.multi && G.have_solo && G.solo != st) {
    // a second stream: from here on launches are followed by events.  The first stream's newest launch has none
    // yet -- one is put behind everything that stream has queued so far.
    G.multi = true;
    SplitGuard::Flight* f = G.slot(G.solo);
    if (f && hipEventRecord(f->ev, G.solo) == hipSuccess) { f->live = true; f->wgs = G.solo_wgs; }
  }
  const int busy = G.multi ? G.busy_elsewhere(st) : 0;
  const int parts = split_parts_for(nbl, npad, ld, G.cus - busy);
  if (!parts) return HPX_OK;
  int rc;
  if (!gen) {
    hpx_gen_batch none = {};
    rc = launch_split_t<false>(nbl, parts, npad, ld, L, Wre, Wim, Vt, info, iter_tag, none, G.xcd_miss, st);
  } else {
    rc = launch_split_t<true>(nbl, parts, npad, ld, L, Wre, Wim, Vt, info, iter_tag, *gen, G.xcd_miss, st);
  }
  if (rc != HPX_OK) return rc;
  *took = parts;
  if (!G.multi) {
    G.solo = st;
    G.have_solo = true;
    G.solo_wgs = 8 * ((nbl + 7) / 8) * parts;
  } else {
    SplitGuard::Flight* f = G.slot(st);
    if (!f || hipEventRecord(f->ev, st) != hipSuccess) {
      // no event: this launch cannot be tracked -- wait for it (never seen; keeps the books truthful)
      HPX_HIP(hipStreamSynchronize(st));
    } else {
      f->live = true;
      f->wgs = 8 * ((nbl + 7) / 8) * parts;
    }
  }
  return HPX_OK;
}
