// Everything after the solve: back transform, model, residual, chi^2, ln posterior, the inverse-gamma bandpower
// draw (k_resid, k_fft_resid, k_dft_resid, k_betam, k_draw) and their launcher.
#include "hpx_chain.h"
#include <type_traits>
#include "hpx_fft.h"

namespace {

// chi^2 of one element with the operation order WRITTEN OUT (explicit fma): the same bits whether or not the samples /
// chi^2 of the iteration are kept or the batch has flags -- left to the compiler's contraction, instantiations that store
// the scaled signal formed the residual from a rounded product, the others from a fused one (ln-posterior differing in
// the last bit between a thinned and an unthinned run of the same chain)
__device__ __forceinline__ double chi2_term(const double rr, const double ri, const double nv) {
  return fma(ri, ri, rr * rr) * nv;
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
    const double c2 = chi2_term(rr, ri, A.ninv_t ? A.ninv_t[ot] : ninv[x]);
    acc = fma(w, c2, acc);
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
// SPLIT8 (blocks of 8 time columns when that is the block size of the channel count, N >= 512): the model term's
// MFMA tile is 16 channels x 16 columns and eight of them were padding -- half the matrix-pipe time, and half the
// lanes idle in the element-wise part behind it.  Here the columns 8 .. 15 carry the IMAGINARY parts of the amplitudes:
// with B = [Re f | Im f] one MFMA chain gives F_re [Re f | Im f] and a second F_im [Re f | Im f] (two MFMAs per k-step
// instead of four); a swap of the lane halves (DPP row_ror:8) brings the partner's second product over, and lane
// (t, g) then holds Re of the model term, lane (t + 8, g) Im.  The residual is finished one COMPONENT per lane: every
// lane loads one of (Re d, Im d), reads one of (Re s, Im s) and adds its square to chi^2.
typedef __attribute__((address_space(3))) double lds_f64_t;
#ifdef HPX_FR_TRACE
__device__ unsigned long long hpx_fr_trace[8 * 4 * 16];
#define FR_TR(id_)                                                                               \
  if (blockIdx.x < 8) {                                                                          \
    unsigned long long t_;                                                                       \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory");                \
    if ((threadIdx.x & 63) == 0) hpx_fr_trace[(blockIdx.x * 4 + (threadIdx.x >> 6)) * 16 + (id_)] = t_; \
  }
#else
#define FR_TR(id_)
#endif
// OPT (the SPLIT8 form): bit 0 the batch has flags (the masked signal is written), bit 1 samples / chi^2 are kept this
// iteration -- compile-time, so that the tile loop of the common iteration carries neither the stores nor their tests
// ELEMS: complex elements of the signal block in LDS: 4096 (64 KB, two workgroups per CU) or, for 1024 channels, 8192 (128 KB:
// eight time columns still fit ONE workgroup per CU, which then has 512 threads -- the same eight waves per CU)
template <int NTH, bool SPLIT8, int OPT, int ELEMS = 4096>
__global__ __launch_bounds__(NTH, (ELEMS > 4096) ? 1 : NTH / 128) void k_fft_resid(const ResArgs A) {
  constexpr int NW = NTH / 64;                       // waves
  extern __shared__ double fl[];
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
  double* bp = A.bpart + ((long)b * HPX_NPART + cg) * N;
  FR_TR(0)
  // The block of the solution first: 16 loads per thread and array, all in flight before anything waits (the block is
  // at most 4096 elements: one batch); the twiddles and the amplitudes are requested behind them (in front of them
  // measures the same: 0.217 ms either way at config 3).
  constexpr int UB = ELEMS / NTH;
  const int total = N << tcs;
  double zr[UB], zi[UB];
#pragma unroll
  for (int u = 0; u < UB; ++u) {
    const int e = min(tid + NTH * u, total - 1), k = e >> tcs, tc = e & (TC - 1);
#if HPX_FR_NT & 1
    zr[u] = __builtin_nontemporal_load(&xre[(long)k * TP + c0 + tc]);
    zi[u] = __builtin_nontemporal_load(&xim[(long)k * TP + c0 + tc]);
#else
    zr[u] = xre[(long)k * TP + c0 + tc];
    zi[u] = xim[(long)k * TP + c0 + tc];
#endif
  }
  for (int j = tid; j < h; j += NTH) {
    tw[j] = A.twre[(long)(h + 1) * N + h + j];
    tw[h + j] = A.twim[(long)(h + 1) * N + h + j];
  }
  for (int e = tid; e < (M << tcs); e += NTH) {
    const int m = e >> tcs, tc = e & (TC - 1);
    lfr[e] = xre[(long)(N + m) * TP + c0 + tc];
    lfi[e] = xim[(long)(N + m) * TP + c0 + tc];
  }
#pragma unroll
  for (int u = 0; u < UB; ++u) {
    const int e = tid + NTH * u;
    if (e < total) {                                      // uniform over the workgroup
      const int k = e >> tcs, tc = e & (TC - 1);
      const double sg = (k & 1) ? -1.0 : 1.0;
      fre[e] = zr[u] * sg;
      fim[e] = zi[u] * sg;
      // sum over this block's time columns: groups of eight first, then the groups -- a block of 16 columns leaves
      // what two blocks of 8 leave once k_draw has added their slots in pairs (the launcher takes 8 instead of 16
      // columns per block for batches that would not fill the CUs: a baseline's chain must not depend on that)
      double v = zr[u] * zr[u] + zi[u] * zi[u];
      if (tcs == 3) {                                     // (constant distances: no trip through the LDS crossbar's queue per step)
        v += __shfl_xor(v, 1, 64);
        v += __shfl_xor(v, 2, 64);
        v += __shfl_xor(v, 4, 64);
      } else {
        for (int o = 1; o < TC; o <<= 1) v += __shfl_xor(v, o, 64);
      }
      if (tc == 0) bp[k] = v;
    }
  }
  FR_TR(1)
  int s = 0;
  for (; s + 3 <= logN; s += 3) {
    __syncthreads();
    FR_TR(2 + s / 3)
    fft_pass<3, 1, NTH>(fre, fim, tw, N, h, logN, s, tcs, tid);
  }
  if (logN - s == 2) {
    __syncthreads();
    fft_pass<2, 1, NTH>(fre, fim, tw, N, h, logN, s, tcs, tid);
  } else if (logN - s == 1) {
    __syncthreads();
    fft_pass<1, 1, NTH>(fre, fim, tw, N, h, logN, s, tcs, tid);
  }
  __syncthreads();
  FR_TR(6)
  const double* dre = A.Dre + (long)b * A.NP * TP;
  const double* dim_ = A.Dim + (long)b * A.NP * TP;
  const double* fmr = A.Fre + (A.fg_shared ? 0 : (long)b * N * M);
  const double* fmi = A.Fim + (A.fg_shared ? 0 : (long)b * N * M);
  const double* ninv = A.ninv + (long)b * N;
  const uint8_t* fl8 = A.flags + (long)b * N;
  double acc = 0.0;
  if (SPLIT8 && M <= 16) {
    constexpr bool FL = (OPT & 1) != 0, KEEP = (OPT & 2) != 0;
    const int wave = tid >> 6, lane = tid & 63, li = lane & 15, g = lane >> 4;
    const bool hi = li >= 8;                          // this lane's component: Re (li < 8) or Im
    const int tc = li & 7, t = c0 + tc;
    const bool tvalid = t < T;
    double bb[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {                  // B[m = 4 ks + g][col = li] = Re f[m][tc] | Im f[m][tc]
      const int m = 4 * ks + g;
      bb[ks] = (m < M) ? (hi ? lfi : lfr)[(m << tcs) + tc] : 0.0;
    }
    const int ntile = N >> 4;
    const int tlast = wave + NW * ((ntile - 1 - wave) / NW);
    // Everything a tile reads, as (wave-uniform base of the tile) + (32-bit lane offset in bytes, fixed over the
    // tiles): one offset register per stream instead of a 64-bit address per load and tile.
    const char* fr_b = reinterpret_cast<const char*>(fmr);
    const char* fi_b = reinterpret_cast<const char*>(fmi);
    const char* d_b = reinterpret_cast<const char*>(hi ? dim_ : dre);      // the lane's component of the data
    const char* nv_b = reinterpret_cast<const char*>(ninv);
    const lds_f64_t* ssel = (const lds_f64_t*)(hi ? fim : fre);            // ... and of the signal (LDS)
    unsigned fo[4], dof[4], sof[4];
    bool mok[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const int m = 4 * ks + g;
      mok[ks] = m < M;
      fo[ks] = 8u * (unsigned)(li * M + min(m, M - 1));                    // (clamped load, then the select: no branch)
    }
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      const int r = HPX_ACC_ROW(g, v);
      dof[v] = 8u * (unsigned)(r * TP + (tvalid ? t : 0));
      sof[v] = (unsigned)r;                                                // channel within the tile
    }
    const double sc = (g & 1) ? -A.isn : A.isn;       // (x & 1 = g & 1: the tile starts at a multiple of 16, rows g + 4 v)
    double nfr[4], nfi[4], nd[4], nnv[4], nw[4];
#define HPX_FR_LOAD8(xt_)                                                             \
  {                                                                                   \
    const int x0_ = (xt_) << 4;                                                       \
    const char* fa_ = fr_b + (long)x0_ * M * 8;                                       \
    const char* fb_ = fi_b + (long)x0_ * M * 8;                                       \
    const char* da_ = d_b + (long)x0_ * TP * 8;                                       \
    _Pragma("unroll") for (int ks = 0; ks < 4; ++ks) {                                \
      const double fr_ = *reinterpret_cast<const double*>(fa_ + fo[ks]);              \
      const double fi_ = *reinterpret_cast<const double*>(fb_ + fo[ks]);              \
      nfr[ks] = mok[ks] ? fr_ : 0.0;                                                  \
      nfi[ks] = mok[ks] ? fi_ : 0.0;                                                  \
    }                                                                                 \
    _Pragma("unroll") for (int v = 0; v < 4; ++v) {                                   \
      nd[v] = *reinterpret_cast<const double*>(da_ + dof[v]);                         \
      nnv[v] = *reinterpret_cast<const double*>(nv_b + 8u * (unsigned)(x0_ + sof[v])); \
      nw[v] = fl8[x0_ + sof[v]] ? 1.0 : 0.0;                                          \
    }                                                                                 \
  }
    if (wave < ntile) HPX_FR_LOAD8(wave)
    for (int xt = wave; xt < ntile; xt += NW) {
      const int x0 = xt << 4;
      double cfr[4], cfi[4], cd[4], cnv[4], cw[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        cfr[q] = nfr[q]; cfi[q] = nfi[q]; cd[q] = nd[q]; cnv[q] = nnv[q]; cw[q] = nw[q];
      }
      HPX_FR_LOAD8(min(xt + NW, tlast))               // branch-free: re-read at the end
      d4 d1 = {0., 0., 0., 0.}, d2 = {0., 0., 0., 0.};
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {                // A[channel x0 + li][m = 4 ks + g] = F[x][m]
        if (4 * ks >= M) break;
        d1 = mfma64(cfr[ks], bb[ks], d1);             // F_re [Re f | Im f]
        d2 = mfma64(cfi[ks], bb[ks], d2);             // F_im [Re f | Im f]
      }
      double mdl[4];
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        // the partner lane's F_im product: Re (F f) = F_re Re f - F_im Im f,  Im (F f) = F_re Im f + F_im Re f
        const long long raw = __double_as_longlong(d2[v]);
        const int lo_ = __builtin_amdgcn_mov_dpp((int)(raw & 0xffffffffll), 0x128, 0xf, 0xf, false);     // row_ror:8
        const int hi_ = __builtin_amdgcn_mov_dpp((int)(raw >> 32), 0x128, 0xf, 0xf, false);
        const double other = __longlong_as_double(((long long)hi_ << 32) | (unsigned int)lo_);
        mdl[v] = hi ? d1[v] + other : d1[v] - other;
      }
      double* gsel = (hi ? A.Gim : A.Gre) + (long)b * A.NP * TP;
      if (tvalid) {
#pragma unroll
        for (int v = 0; v < 4; ++v) {
          const int x = x0 + HPX_ACC_ROW(g, v);
          const int pidx = (int)(__brev((unsigned)x) >> (32 - logN));
          const double sraw = ssel[(pidx << tcs) + tc];
          const double sv = sraw * sc;                // (for the outputs; the residual takes the fused form)
          const double r = cd[v] - fma(sraw, sc, mdl[v]);
          const double w = cw[v];
          const double c2 = (r * r) * cnv[v];         // this component's share of the channel's chi^2 term
          acc = fma(w, c2, acc);
          if (FL) gsel[(long)x * TP + t] = w * sv;
          if (KEEP) {
            if (A.cr_out) A.cr_out[(long)b * A.cr_bstride + ((long)t * N + x) * 2 + (hi ? 1 : 0)] = sv;
            if (A.chisq_out) {                        // (the two components meet for it)
              const long long rc = __double_as_longlong(c2);
              const int cl = __builtin_amdgcn_mov_dpp((int)(rc & 0xffffffffll), 0x128, 0xf, 0xf, false);
              const int ch = __builtin_amdgcn_mov_dpp((int)(rc >> 32), 0x128, 0xf, 0xf, false);
              const double c2o = __longlong_as_double(((long long)ch << 32) | (unsigned int)cl);
              if (!hi) A.chisq_out[(long)b * A.chisq_bstride + (long)t * N + x] = c2 + c2o;
            }
          }
        }
      } else if (FL) {                                // a padding column: the masked signal is zero there
#pragma unroll
        for (int v = 0; v < 4; ++v) gsel[(long)(x0 + HPX_ACC_ROW(g, v)) * TP + t] = 0.0;
      }
    }
#undef HPX_FR_LOAD8
  } else if (M <= 16) {
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
    constexpr bool FL = (OPT & 1) != 0, KEEP = (OPT & 2) != 0;      // (as in the form above)
    const int t = c0 + li;
    // Tiles run over channels in NATURAL order, so that everything in global memory (data, mode
    // rows, noise, outputs) is touched with unit stride; the bit reversal of the FFT output is
    // undone by the LDS read of s instead.  The global operands of the next tile are requested
    // before the current one is worked on (two workgroups per CU: nothing else hides them).
    const int ntile = N >> 4;
    const int tlast = wave + NW * ((ntile - 1 - wave) / NW);      // this wave's last tile
    const bool tvalid = (li < TC) && (t < T);
    // (wave-uniform base of the tile) + (32-bit lane offset in bytes, fixed over the tiles)
    const char* fr_b = reinterpret_cast<const char*>(fmr);
    const char* fi_b = reinterpret_cast<const char*>(fmi);
    const char* dr_b = reinterpret_cast<const char*>(dre);
    const char* di_b = reinterpret_cast<const char*>(dim_);
    const char* nv_b = reinterpret_cast<const char*>(ninv);
    unsigned fo[4], dof[4], sof[4];
    bool mok[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const int m = 4 * ks + g;
      mok[ks] = m < M;
      fo[ks] = 8u * (unsigned)(li * M + min(m, M - 1));
    }
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      const int r = HPX_ACC_ROW(g, v);
      dof[v] = 8u * (unsigned)(r * TP + (tvalid ? t : 0));
      sof[v] = (unsigned)r;
    }
    const double sc = (g & 1) ? -A.isn : A.isn;       // (x & 1 = g & 1: rows g + 4 v of a tile that starts at a multiple of 16)
    double nfr[4], nfi[4], ndr[4], ndi[4], nnv[4], nw[4];
#define HPX_FR_LOAD(xt_)                                                              \
  {                                                                                   \
    const int x0_ = (xt_) << 4;                                                       \
    const char* fa_ = fr_b + (long)x0_ * M * 8;                                       \
    const char* fb_ = fi_b + (long)x0_ * M * 8;                                       \
    const char* da_ = dr_b + (long)x0_ * TP * 8;                                      \
    const char* db_ = di_b + (long)x0_ * TP * 8;                                      \
    _Pragma("unroll") for (int ks = 0; ks < 4; ++ks) {                                \
      const double fr_ = *reinterpret_cast<const double*>(fa_ + fo[ks]);              \
      const double fi_ = *reinterpret_cast<const double*>(fb_ + fo[ks]);              \
      nfr[ks] = mok[ks] ? fr_ : 0.0;                                                  \
      nfi[ks] = mok[ks] ? fi_ : 0.0;                                                  \
    }                                                                                 \
    _Pragma("unroll") for (int v = 0; v < 4; ++v) {                                   \
      ndr[v] = *reinterpret_cast<const double*>(da_ + dof[v]);                        \
      ndi[v] = *reinterpret_cast<const double*>(db_ + dof[v]);                        \
      nnv[v] = *reinterpret_cast<const double*>(nv_b + 8u * (unsigned)(x0_ + sof[v])); \
      nw[v] = fl8[x0_ + sof[v]] ? 1.0 : 0.0;                                          \
    }                                                                                 \
  }
    if (wave < ntile) HPX_FR_LOAD(wave)
    for (int xt = wave; xt < ntile; xt += NW) {
      const int x0 = xt << 4;
      double cfr[4], cfi[4], cdr[4], cdi[4], cnv[4], cw[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        cfr[q] = nfr[q]; cfi[q] = nfi[q]; cdr[q] = ndr[q]; cdi[q] = ndi[q]; cnv[q] = nnv[q]; cw[q] = nw[q];
      }
      HPX_FR_LOAD(min(xt + NW, tlast))                // branch-free: re-read at the end
      d4 mr = {0., 0., 0., 0.}, mi = {0., 0., 0., 0.};
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {                // A[channel x0 + li][m = 4 ks + g] = F[x][m]
        if (4 * ks >= M) break;                       // k-steps made of padding only (M <= 12: one in four)
        mr = mfma64(cfr[ks], bfr[ks], mr);
        mr = mfma64(-cfi[ks], bfi[ks], mr);
        mi = mfma64(cfr[ks], bfi[ks], mi);
        mi = mfma64(cfi[ks], bfr[ks], mi);
      }
      if (tvalid) {
#pragma unroll
        for (int v = 0; v < 4; ++v) {
          const int x = x0 + HPX_ACC_ROW(g, v);
          const int pidx = (int)(__brev((unsigned)x) >> (32 - logN));
          const double rawr = fre[(pidx << tcs) + li], rawi = fim[(pidx << tcs) + li];
          const double sr = rawr * sc, si = rawi * sc;                 // (for the outputs)
          const double rr = cdr[v] - fma(rawr, sc, mr[v]), ri = cdi[v] - fma(rawi, sc, mi[v]);
          const double w = cw[v];
          const double c2 = chi2_term(rr, ri, cnv[v]);
          acc = fma(w, c2, acc);
          if (FL) {
            const long o = (long)b * A.NP * TP + (long)x * TP + t;
            A.Gre[o] = w * sr;
            A.Gim[o] = w * si;
          }
          if (KEEP) {
            if (A.cr_out) {
              double* q = A.cr_out + (long)b * A.cr_bstride + ((long)t * N + x) * 2;
              q[0] = sr;
              q[1] = si;
            }
            if (A.chisq_out) A.chisq_out[(long)b * A.chisq_bstride + (long)t * N + x] = c2;
          }
        }
      } else if (FL && li < TC) {                     // a padding column: the masked signal is zero there
#pragma unroll
        for (int v = 0; v < 4; ++v) {
          const long o = (long)b * A.NP * TP + (long)(x0 + HPX_ACC_ROW(g, v)) * TP + t;
          A.Gre[o] = 0.0;
          A.Gim[o] = 0.0;
        }
      }
    }
#undef HPX_FR_LOAD
  } else {
    // 256 elements (256 / TC channels x TC times) per round; the foreground-mode rows F[x][:] of
    // the round's channels are staged in LDS first (TC threads share a channel: from global
    // memory each of them would fetch the same 2 M values again)
    double* sfr = lfi + (M << tcs);                   // [256 / TC][M]
    double* sfi = sfr + (NTH >> tcs) * M;
    const int xper = NTH >> tcs;
    for (int e0 = 0; e0 < (N << tcs); e0 += NTH) {
      __syncthreads();
      for (int q = tid; q < xper * M; q += NTH) {
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
      const double c2 = chi2_term(rr, ri, ninv[x]);
      acc = fma(w, c2, acc);
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
    for (int e = tid; e < (M << tcs); e += NTH) {
      const int m = e >> tcs, t = c0 + (e & (TC - 1));
      if (t >= T) continue;
      double* q = A.fg_out + (long)b * A.fg_bstride + ((long)t * M + m) * 2;
      q[0] = lfr[e];
      q[1] = lfi[e];
    }
  }
  FR_TR(7)
  // chi^2 total of the block, again per group of eight time columns first (a lane's column is tid & (TC - 1) in both
  // branches above; with TC = 16 bit 3 of the lane tells the group): the lanes of a group of eight, the groups of a
  // wave that belong to the same eight columns, the waves ((w0 + w1) + (w2 + w3)), then columns 0-7 + columns 8-15
  __shared__ double red2[2 * NW];
  double v = acc;
  v += __shfl_xor(v, 1, 64);
  v += __shfl_xor(v, 2, 64);
  v += __shfl_xor(v, 4, 64);
  const bool second = (TC == 16) && (tid & 8);
  double w0 = second ? 0.0 : v, w1 = second ? v : 0.0;
#pragma unroll
  for (int o = 8; o < 64; o <<= 1) {
    w0 += __shfl_xor(w0, o, 64);
    w1 += __shfl_xor(w1, o, 64);
  }
  if ((tid & 63) == 0) {
    red2[tid >> 6] = w0;
    red2[NW + (tid >> 6)] = w1;
  }
  __syncthreads();
  if (tid == 0) {                    // (the waves in pairs, the pairs in order)
    double t0 = 0.0, t1 = 0.0;
    for (int w = 0; w < NW; w += 2) {
      t0 += red2[w] + red2[w + 1];
      t1 += red2[NW + w] + red2[NW + w + 1];
    }
    A.lnpart[(long)b * HPX_NPART + cg] = t0 + t1;
  }
  FR_TR(8)
}
#ifdef HPX_FR_TRACE
extern "C" int hpx_debug_fr_trace(unsigned long long* host) {
  return hipMemcpyFromSymbol(host, HIP_SYMBOL(hpx_fr_trace), sizeof(unsigned long long) * 8 * 4 * 16) == hipSuccess ? 0 : -2;
}
#endif

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
        const double s_r = sr[v] * A.isn, s_i = si[v] * A.isn;       // (for the outputs)
        const double rr = dre[o] - fma(sr[v], A.isn, mr[v]), ri = dim_[o] - fma(si[v], A.isn, mi[v]);
        const double w = cw[v];
        const double c2 = chi2_term(rr, ri, cnv[v]);
        acc = fma(w, c2, acc);
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
__device__ __forceinline__ double igamc_int(const int a, const double z, const double lgam_a, const double* rk) {
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

// One inversion draw by a group of W = 16, 32 or 64 lanes of a wave; pspec.py:50-62.  The reference tabulates the CDF
// on the 1000-point grid, normalises it (cdf -= min; cdf /= max), de-duplicates and interpolates linearly
// at u.  The table is monotone, so min and max are its end points and the bracket
// [first occurrence of the previous distinct value, first value >= u] is found by W-ary searches
// (one evaluation per lane and round) instead of evaluating all 1000 points, and
// the prior channels of a baseline are sampled side by side (256 / W groups per workgroup).
// All lanes of the group return the sample.
// Every round is a dependent chain of a grid read, two divisions and a 31-term sum (about 1.5 us at T = 32), so the
// rounds are what a draw costs: two for the end points + three per search x (up to) three searches + two for the
// bracket's values it had been, 22 of k_draw's 30 us at config 2.  Now, at W = 64 and the reference's 1000 points, TWO:
// the end points ride on the first level of the search (63 probes + the first point), and the last round evaluates the
// table at hi, hi - 1, ..., hi - (W - 1) for the upper bound hi of the remaining bracket: its top lanes finish the search,
// and the previous distinct value, its first occurrence (where the run of equal values below the answer ends) and both
// bracket values come out of the same round by shuffles.  (W = 32: three rounds, W = 16: four.)  Only a run of equal
// values longer than the group looked at (the flat ends of the table) and the u-outside-the-table cases take the
// searches of the general form.
template <int W, class Pred>
__device__ __forceinline__ int first_true_w(const int n, Pred pred) {
  // smallest i in [0, n) with pred(i), n if none; pred is monotone (false ... false true ... true)
  static_assert(W == 16 || W == 32 || W == 64, "group width");
  constexpr int LW = (W == 16) ? 4 : (W == 32 ? 5 : 6);
  const int j = threadIdx.x & (W - 1), gsh = (threadIdx.x & 63) & ~(W - 1);
  const unsigned long long gmask = (W == 64) ? ~0ull : ((1ull << (W & 63)) - 1ull);
  int lo = 0, hi = n;                   // the answer is in [lo, hi]; hi < n is known to be true
  while (hi > lo) {
    const int step = (hi - lo + W - 1) >> LW;
    const int idx = min(lo + (j + 1) * step - 1, hi - 1);
    const bool p = pred(idx);
    const int m = __popcll((__ballot(!p) >> gsh) & gmask);     // leading false probes
    if (m == W) break;                  // every probe up to hi - 1 is false: the answer is hi
    const int nlo = lo + m * step;
    hi = min(lo + (m + 1) * step - 1, hi - 1);
    lo = nlo;
  }
  return hi;
}

// the general form: every case of the reference, by searches alone
template <int W, class XG>
__device__ __forceinline__ double inversion_draw_general(const int alpha, const double lgam, const double beta, const double u,
                                         const XG xg, const int ngrid, const double* rk,
                                         const double mn, const double mx, int hi) {
  auto cval = [&](const int i) { return (igamc_int(alpha, beta / xg[i], lgam, rk) - mn) / mx; };
  if (hi >= ngrid) {                    // u above the table: last two distinct values
    const double top = cval(ngrid - 1);
    hi = first_true_w<W>(ngrid, [&](const int i) { return !(cval(i) < top); });
  }
  int lo;
  if (hi == 0) {                        // u at/below the first value: first two distinct values
    const double bot = cval(0);
    hi = first_true_w<W>(ngrid, [&](const int i) { return !(cval(i) <= bot); });
    lo = 0;
    if (hi >= ngrid) return xg[0];      // degenerate table (all equal): the reference would give NaN
  } else {
    const double below = cval(hi - 1);  // previous distinct value; its first occurrence:
    lo = first_true_w<W>(ngrid, [&](const int i) { return !(cval(i) < below); });
  }
  const double clo = cval(lo), chi = cval(hi), xlo = xg[lo], xhi = xg[hi];
  const double slope = (xhi - xlo) / (chi - clo);
  return slope * (u - clo) + xlo;
}

// XG: pointer type of the grid row (staging the slice's rows in LDS first was tried: the copy and its barrier cost more
// than the searches' reads, which hit in L2 -- config 2: 30.3 against 24.9 us)
template <int W, class XG>
__device__ __forceinline__ double inversion_draw(const int alpha, const double lgam, const double beta, const double u,
                                 const XG xg, const int ngrid, const double* rk) {
  constexpr int LW = (W == 16) ? 4 : (W == 32 ? 5 : 6);
  const int j = threadIdx.x & (W - 1), gsh = (threadIdx.x & 63) & ~(W - 1);
  const unsigned long long gmask = (W == 64) ? ~0ull : ((1ull << (W & 63)) - 1ull);
  auto raw = [&](const int i) { return igamc_int(alpha, beta / xg[i], lgam, rk); };
  // ---- round 1: the end points AND the first level of the search.  Lanes 0 .. W-2 probe every step1-th point (the
  // last of them is the table's last point: cdf[-1]), lane W-1 the first point (cdf.min()).
  const int step1 = (ngrid + W - 2) / (W - 1);
  const int idx1 = (j == W - 1) ? 0 : min((j + 1) * step1 - 1, ngrid - 1);
  const double e1 = raw(idx1);
  const double mn = __shfl(e1, gsh + W - 1, 64);
  const double mx = __shfl(e1, gsh + W - 2, 64) - mn;                            // (cdf - min).max()
  auto cval = [&](const int i) { return (raw(i) - mn) / mx; };
  // searchsorted(unique, u, 'left') in original indexing = number of table values < u
  const bool f1 = (j < W - 1) && (((e1 - mn) / mx) < u);
  const int m1 = __popcll((__ballot(f1) >> gsh) & gmask);                        // leading false probes
  if (m1 == W - 1) return inversion_draw_general<W, XG>(alpha, lgam, beta, u, xg, ngrid, rk, mn, mx, ngrid);
  int lo = m1 * step1, hi = min((m1 + 1) * step1 - 1, ngrid - 1);                // the answer is in [lo, hi]; hi is true
  // ---- further W-ary rounds until the last round can take the rest of the bracket along (W = 64: 15 points, as
  // the lanes of the last round look at hi, hi - 1, ..., hi - 63; narrower groups search to the end)
  constexpr int NARROW = (W == 64) ? 15 : 0;
  while (hi - lo > NARROW) {
    const int step = (hi - lo + W - 1) >> LW;
    const int idx = min(lo + (j + 1) * step - 1, hi - 1);
    const bool f = cval(idx) < u;
    const int m = __popcll((__ballot(f) >> gsh) & gmask);
    if (m == W) { lo = hi; break; }     // every probe up to hi - 1 is false: the answer is hi
    const int nlo = lo + m * step;
    hi = min(lo + (m + 1) * step - 1, hi - 1);
    lo = nlo;
  }
  // ---- last round.  Lane j: the table at hi - j.  The lanes 0 .. hi - lo finish the search (the first true one from
  // the top is hi itself); the lanes below give the previous distinct value, its first occurrence (where the run of
  // equal values ends) and both bracket values by shuffles.
  const int span = hi - lo;
  const double cv = cval(max(hi - j, 0));
  const bool tr = (j <= span) && !(cv < u);
  const int mt = __popcll((__ballot(tr) >> gsh) & gmask);                        // true probes: lanes 0 .. mt - 1
  const int sh = mt - 1;                                                         // lane of the answer
  hi -= sh;
  if (mt == 0 || hi == 0) return inversion_draw_general<W, XG>(alpha, lgam, beta, u, xg, ngrid, rk, mn, mx, mt == 0 ? ngrid : 0);
  const double chi = __shfl(cv, gsh + sh, 64), below = __shfl(cv, gsh + sh + 1, 64);   // below: the previous distinct value
  // its first occurrence: the run of lanes sh + 1, sh + 2, ... that hold a value >= below (monotone table: == below)
  const bool ge = (j > sh) && (hi + sh - j >= 0) && !(cv < below);
  const unsigned long long bits = (((__ballot(ge) >> gsh) & gmask) >> 1) >> sh;  // bit i: lane sh + 1 + i
  const int run = (int)__builtin_ctzll(~bits);                                   // (the top bit of bits is clear)
  if (run == W - 1 - sh && hi - run > 0)  // equal values as far down as the group looked: the general search
    return inversion_draw_general<W, XG>(alpha, lgam, beta, u, xg, ngrid, rk, mn, mx, hi);
  const int lo2 = hi - run;
  const double clo = __shfl(cv, gsh + sh + run, 64);
  const double xlo = xg[lo2], xhi = xg[hi];
  const double slope = (xhi - xlo) / (chi - clo);
  return slope * (u - clo) + xlo;
}

// The ln posterior's sum over the channels from its sums over groups of 16 channels: four groups make a block of 64,
// ((g0 + g1) + (g2 + g3)), the blocks are added in order -- one definition for the kernel that has a baseline to itself
// and for the combination of a sliced run (hpx_chain.hip: k_lnpost_combine uses the same function).
struct DrawArgs {
  const double *bpart, *lnpart, *betam, *uni, *igy, *xgrid, *ps_forced;
  double *beta, *lnp1;
  int npart, pair_slots;             // pair_slots: the residual kernel's slots hold eight time columns each (see k_fft_resid)
  const int32_t* pmap;
  double *ia, *ps_cur, *ps_out;
  long ps_bstride, forced_bstride;   // strides between baselines in ps_out / ps_forced
  int N, T, ngrid, prior_shared, any_flags;
  double lgam_T;
  double* lnhist;                    // several slices: this iteration's [nbl][ceil(N / 16) + 1] group sums (k_lnpost_combine)
  double* lnpost_out;                // one slice: the caller's ln-posterior history, offset to this iteration
  long lnpost_pitch;
};

// Grid (slices, baselines): a baseline's channels are dealt to `gridDim.x` workgroups in blocks of 64 -- one per
// baseline for large batches, up to sixteen for batches that would leave most CUs idle (config 2).  The ln-posterior's
// sum over the channels is formed per group of 16 (a fixed shuffle tree) and the groups are added in a fixed order
// (lnpost_blocks): by
// this kernel when a baseline has one slice, else by k_lnpost_combine at the end of the run from the block sums every
// slice leaves in the plan's history (no hand-off between workgroups: an agent-scope release per workgroup is an L2
// write-back, serialised between the CUs of an XCD -- 34 us for a 10 us kernel when it was tried).  Either way the
// result does not depend on the number of slices.
__global__ __launch_bounds__(256) void k_draw(const DrawArgs A) {
  extern __shared__ double dyn[];     // the channels with a prior: N ints (channel), N ints (grid row), N doubles (beta)
  __shared__ int pcount;
  __shared__ double rk_s[HPX_RK_MAX];
  __shared__ double part[256];        // the slice's sums over groups of 16 channels
  const int b = blockIdx.y, tid = threadIdx.x, N = A.N;
  const int nsub = (N + 15) >> 4;     // groups of 16 channels: what the slices are made of
  const int sb0 = (int)(((long)nsub * blockIdx.x) / gridDim.x), sb1 = (int)(((long)nsub * (blockIdx.x + 1)) / gridDim.x);
  const int k0 = sb0 * 16, k1 = min(N, sb1 * 16);
  for (int k = tid; k < HPX_RK_MAX; k += 256) rk_s[k] = 1.0 / (double)(k > 0 ? k : 1);
  const double* rk = (A.T <= HPX_RK_MAX) ? rk_s : nullptr;
  if (tid == 0) pcount = 0;
  __syncthreads();
  double* beta = A.beta + (long)b * N;
  const double* bm = A.any_flags ? A.betam + (long)b * N : beta;
  const int32_t* pmap = A.pmap + (A.prior_shared ? 0 : (long)b * N);
  double* ps_out = A.ps_out + (long)b * A.ps_bstride;
  // One pass over the slice's channels, everything a channel needs requested together (the prior map and the
  // inverse-gamma variate do not wait for the sums):
  //   beta_k = N sum_t |z_kt|^2 from the partial sums of the residual kernel (one slot per block of a baseline
  //   there), added in slot order;
  //   channels without a prior: x = beta * invgamma.ppf(U, a=T-1)   (pspec.py:125);
  //   channels with a prior: truncated draw with shape alpha+1 = T   (pspec.py:121-123) -- collected (channel, grid
  //   row, beta) in LDS, drawn below by groups of lanes; each draw depends only on its own channel, so their order
  //   is immaterial.
  int* plist = reinterpret_cast<int*>(dyn);           // [N] channel, [N] grid row, then [N] doubles: beta
  int* prow = plist + N;
  double* pbeta = dyn + N;
  for (int k = k0 + tid; k < k1; k += 256) {
    const int pm = pmap[k];
    const double y = A.igy[k];
    double sum = 0.0;
    if (A.pair_slots) {               // (slots of eight time columns: two of them are what a slot of sixteen holds)
      for (int j = 0; j < A.npart; j += 2)
        sum += A.bpart[((long)b * HPX_NPART + j) * N + k] + (j + 1 < A.npart ? A.bpart[((long)b * HPX_NPART + j + 1) * N + k] : 0.0);
    } else {
      for (int j = 0; j < A.npart; ++j) sum += A.bpart[((long)b * HPX_NPART + j) * N + k];
    }
    const double bk = (double)N * sum;
    beta[k] = bk;
    if (pm < 0) ps_out[k] = y * bk;
    else {
      const int slot = atomicAdd(&pcount, 1);
      plist[slot] = k;
      prow[slot] = pm;
      pbeta[slot] = bk;
    }
  }
  __syncthreads();
  const int np = pcount;
  // one prior channel per group of 64, 32 or 16 lanes: the widest that takes them all at once (wider groups need
  // fewer search rounds); any width gives the same sample
  auto draw_w = [&](auto wc) {
    constexpr int W = decltype(wc)::value;
    for (int i = tid / W; i < np; i += 256 / W) {
      const int k = plist[i], row = prow[i];
      const double v = inversion_draw<W, const double*>(A.T, A.lgam_T, pbeta[i], A.uni[k],
                                                        A.xgrid + (long)row * A.ngrid, A.ngrid, rk);
      if ((tid & (W - 1)) == 0) ps_out[k] = v;
    }
  };
  if (np <= 4) draw_w(std::integral_constant<int, 64>{});
  else if (np <= 8) draw_w(std::integral_constant<int, 32>{});
  else draw_w(std::integral_constant<int, 16>{});
  __syncthreads();
  // second ln-posterior term, the next 1 / a: sixteen lanes per group of 16 channels, a fixed shuffle tree each
  for (int sb = sb0 + (tid >> 4); sb < sb1; sb += 16) {
    const int k = sb * 16 + (tid & 15);
    double v = 0.0;
    if (k < N) {
      const double pn = ps_out[k];
      v = bm[k] / pn;
      const double nx = A.ps_forced ? A.ps_forced[(long)b * A.forced_bstride + k] : pn;
      A.ps_cur[(long)b * N + k] = nx;
      A.ia[(long)b * N + k] = inv_a(nx, (double)N);
    }
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) v += __shfl_down(v, o, 16);
    if ((tid & 15) == 0) part[(sb - sb0) & 255] = v;      // (a slice holds at most 256 groups: 4096 channels)
  }
  __syncthreads();
  if (tid != 0) return;
  double tot = 0.0;                  // chi^2 total of the residual kernel's partial sums, in slot order
  if (A.pair_slots) {
    for (int j = 0; j < A.npart; j += 2)
      tot += A.lnpart[(long)b * HPX_NPART + j] + (j + 1 < A.npart ? A.lnpart[(long)b * HPX_NPART + j + 1] : 0.0);
  } else {
    for (int j = 0; j < A.npart; ++j) tot += A.lnpart[(long)b * HPX_NPART + j];
  }
  if (gridDim.x == 1) {
    const double lnp = -tot - lnpost_blocks(part, nsub);     // -> ln posterior
    A.lnp1[b] = lnp;
    if (A.lnpost_out) A.lnpost_out[(long)b * A.lnpost_pitch] = lnp;
  } else {
    double* h = A.lnhist + (long)b * (nsub + 1);
    for (int sb = sb0; sb < sb1; ++sb) h[sb] = part[sb - sb0];
    if (blockIdx.x == 0) h[nsub] = tot;
  }
}

__global__ void k_inv_test(const int alpha, const double lgam, const double* __restrict__ beta,
                           const double* __restrict__ u, const double* __restrict__ xgrid,
                           const int ngrid, double* __restrict__ out) {
  __shared__ double rk_s[HPX_RK_MAX];
  const int i = blockIdx.x;
  for (int k = threadIdx.x; k < HPX_RK_MAX; k += 256) rk_s[k] = 1.0 / (double)(k > 0 ? k : 1);
  __syncthreads();
  // the three group widths, one wave each; they must agree to the bit (NaN otherwise)
  __shared__ double res[3];
  const double* rk = alpha <= HPX_RK_MAX ? rk_s : nullptr;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  if (wave == 0 && lane < 16) {
    const double v = inversion_draw<16, const double*>(alpha, lgam, beta[i], u[i], xgrid + (long)i * ngrid, ngrid, rk);
    if (lane == 0) res[0] = v;
  } else if (wave == 1 && lane < 32) {
    const double v = inversion_draw<32, const double*>(alpha, lgam, beta[i], u[i], xgrid + (long)i * ngrid, ngrid, rk);
    if (lane == 0) res[1] = v;
  } else if (wave == 2) {
    const double v = inversion_draw<64, const double*>(alpha, lgam, beta[i], u[i], xgrid + (long)i * ngrid, ngrid, rk);
    if (lane == 0) res[2] = v;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    const bool same = __double_as_longlong(res[0]) == __double_as_longlong(res[1]) &&
                      __double_as_longlong(res[0]) == __double_as_longlong(res[2]);
    out[i] = same ? res[0] : __longlong_as_double(0x7ff8000000000000ll);
  }
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

}  // namespace


#ifndef HPX_DFT_RESID
#define HPX_DFT_RESID 1       // small N without an FFT: dense transform + residual in one kernel
#endif
#ifndef HPX_FUSE_TC
#define HPX_FUSE_TC 8      // fewest time columns per block for which the fused transform + residual kernel is used
#endif
int hpx_post_solve(hpx_plan* p, int it_abs, const IterOut& O, hipStream_t st) {
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
#ifndef HPX_FR_THREADS
// threads of a workgroup of k_fft_resid: 256.  512 (twice the waves per CU to hide the tile operands' latency) leaves 128
// registers a lane: 44 spilled, config 3 0.37 against 0.27 ms, config 2 22.7 against 18.6 us
#define HPX_FR_THREADS 256
#endif
#ifndef HPX_FR_SPLIT8
#define HPX_FR_SPLIT8 1     // 0: the four-MFMA form at every block size (A/B)
#endif
#ifndef HPX_FR_HALVE
#define HPX_FR_HALVE 1      // 0: never take 8-column blocks for small batches (A/B, tests)
#endif
#ifndef HPX_FR_ELEMS
#define HPX_FR_ELEMS 4096      // complex elements of the signal block a workgroup of k_fft_resid holds in LDS
#endif
#ifndef HPX_FR_BIG
#define HPX_FR_BIG 1           // 0: never the 8192-element block (A/B)
#endif
  // (1024 channels: a block of 8192 elements, 128 KB of LDS, one 512-thread workgroup per CU -- with 4096 only four time
  // columns fit and the transform and the residual are two kernels with the signal through HBM between them)
  const int fr_elems = (HPX_FR_BIG && NP == 1024 && HPX_FR_ELEMS == 4096) ? 8192 : HPX_FR_ELEMS;
  int npart = 1, TC = fr_elems / NP, pair_slots = 0;
  if (TC > 16) TC = 16;
  // a batch whose blocks of 16 columns would not reach every CU takes blocks of 8 (config 2: 128 -> 256 workgroups,
  // 23.8 -> 18.9 us); the sums the blocks leave are formed per group of eight columns either way, so a baseline's
  // results do not depend on the batch it is in
  {
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    if (HPX_FR_HALVE && TC == 16 && M <= 16 && (long)nbl * (TP / 16) < (long)cus) TC = 8;
  }
  const bool pow2 = N == NP && (N & (N - 1)) == 0 && N >= 32 && N <= 4096;
  const bool generic_post = p->dense_noise || p->per_time;      // modes only the two-kernel form implements
  if (pow2 && hpx_dft_use_fft && TC >= HPX_FUSE_TC && TP / TC <= HPX_NPART && !generic_post) {   // fewer columns per block: two kernels win
    while ((1 << R.logN) < N) ++R.logN;
    while ((1 << R.tcs) < TC) ++R.tcs;
    npart = TP / TC;
    pair_slots = (TC == 8);
    // s = U z, residual, chi^2, |z|^2 sums in one pass (k_fft_resid); the two event marks
    // book it under "transform"
    // (the mode rows' staging area behind the amplitudes only where the model term is not on the matrix pipe)
    R.nbl = nbl; R.npart = npart;
    // (blocks of 8 columns because the channel count leaves no other choice: the component-per-lane form; where 16
    // would fit and 8 is taken for a small batch, the sums must come out as a block of 16 leaves them: the other form)
    const bool big = fr_elems > 4096;
    const size_t lds = ((size_t)N * TC * 2 + N + (size_t)2 * M * TC + (M > 16 ? (size_t)2 * M * ((big ? 512 : HPX_FR_THREADS) / TC) : 0)) * sizeof(double);
    const dim3 grid(((nbl + 7) / 8) * 8 * npart), block(big ? 512 : HPX_FR_THREADS);
#define HPX_FR_GO(SP_, OPT_)                                                                                    \
  {                                                                                                             \
    if (big) {                                                                                                  \
      static hpx_lds_limit limb_;                                                                               \
      HPX_TRY(limb_.ensure(reinterpret_cast<const void*>(&k_fft_resid<512, SP_, OPT_, 8192>), lds));            \
      hipLaunchKernelGGL((k_fft_resid<512, SP_, OPT_, 8192>), grid, block, lds, st, R);                         \
    } else {                                                                                                    \
      static hpx_lds_limit lim_;                                                                                \
      HPX_TRY(lim_.ensure(reinterpret_cast<const void*>(&k_fft_resid<HPX_FR_THREADS, SP_, OPT_>), lds));        \
      hipLaunchKernelGGL((k_fft_resid<HPX_FR_THREADS, SP_, OPT_>), grid, block, lds, st, R);                    \
    }                                                                                                           \
  }
    if (HPX_FR_SPLIT8 && TC == 8 && fr_elems / NP == 8 && M <= 16) {
      const int opt = (p->any_flags ? 1 : 0) | ((R.cr_out || R.chisq_out) ? 2 : 0);
      if (opt == 0) HPX_FR_GO(true, 0)
      else if (opt == 1) HPX_FR_GO(true, 1)
      else if (opt == 2) HPX_FR_GO(true, 2)
      else HPX_FR_GO(true, 3)
    } else {
      const int opt = (p->any_flags ? 1 : 0) | ((R.cr_out || R.chisq_out) ? 2 : 0);
      if (opt == 0) HPX_FR_GO(false, 0)
      else if (opt == 1) HPX_FR_GO(false, 1)
      else if (opt == 2) HPX_FR_GO(false, 2)
      else HPX_FR_GO(false, 3)
    }
#undef HPX_FR_GO
    HPX_HIP(hipGetLastError());
    HPX_TRY(hpx_mark(p, st));
  } else if (HPX_DFT_RESID && NP <= 256 && M <= 16 && hpx_dft_use_fft && !generic_post) {
    // small N without an in-LDS FFT: dense transform fused with the residual (k_dft_resid), booked
    // under "transform"
    npart = (NP / 16 + 3) / 4;
    R.nbl = nbl; R.npart = npart;
    hipLaunchKernelGGL(k_dft_resid, dim3(npart, nbl), dim3(256), 0, st, R);
    HPX_HIP(hipGetLastError());
    HPX_TRY(hpx_mark(p, st));
  } else {
    // s = U z = conj(F) X / sqrt(N)   (rows >= N of X meet the zero padding of the operator)
    HPX_TRY(hpx_launch_dft(nbl, NP, TP, p->Fopre, p->Fopim, 1, p->Xre, p->Xim, (long)p->npad * TP,
                           TP, nullptr, 0, p->Sre, p->Sim, (long)NP * TP, TP, isn, st, N == NP));
    if (p->dense_noise == 2) {
      // dense noise with flags: X = [Y_r | Y_P] so far (unflagged-noise system); the Woodbury correction
      // x = Y_r + Y_P (I - Q^H Y_P)^-1 Q^H Y_r, where Q^H Y is the model U y + F f at the flagged channels
      HPX_TRY(hpx_woodbury_correct(p, T, it_abs + 1, st));
    }
    HPX_TRY(hpx_mark(p, st));
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
  HPX_TRY(hpx_mark(p, st));
  DrawArgs D;
  D.beta = p->beta; D.betam = p->betam; D.lnp1 = p->lnp1;
  D.bpart = p->bpart; D.lnpart = p->lnpart; D.npart = npart; D.pair_slots = pair_slots;
  D.uni = p->uni + (long)it_abs * N; D.igy = p->igy + (long)it_abs * N;
  D.xgrid = p->xgrid; D.pmap = p->pmap;
  D.ps_forced = O.ps_forced; D.forced_bstride = O.forced_bstride;
  D.ia = p->ia; D.ps_cur = p->ps_cur;
  D.ps_out = O.ps_out; D.ps_bstride = O.ps_bstride;
  D.N = N; D.T = T; D.ngrid = p->ngrid; D.prior_shared = p->prior_shared;
  D.any_flags = p->any_flags; D.lgam_T = p->lgam_T;
  // slices per baseline (hpx_plan_set_rng): more than one only for batches that would leave most CUs idle
  const int nslice = p->draw_slices > 0 ? p->draw_slices : 1;
  D.lnhist = nslice > 1 ? p->lnhist + (long)it_abs * nbl * ((N + 15) / 16 + 1) : nullptr;
  D.lnpost_out = nslice > 1 ? nullptr : O.lnpost_out;      // (several slices: k_lnpost_combine at the end of the run)
  D.lnpost_pitch = O.lnpost_pitch;
  const size_t draw_lds = (size_t)N * 16;                   // the prior channels' list (channel, grid row, beta)
  static hpx_lds_limit draw_limit;                          // (with the kernel's static 6 KB beyond 64 KB from N = 3700 on)
  if (draw_lds > 48 * 1024) HPX_TRY(draw_limit.ensure(reinterpret_cast<const void*>(&k_draw), draw_lds));
  hipLaunchKernelGGL(k_draw, dim3(nslice, nbl), dim3(256), draw_lds, st, D);
  HPX_HIP(hipGetLastError());
  HPX_TRY(hpx_mark(p, st));
  return HPX_OK;
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

