// Ceiling probe for a split factorisation: the "tiles below a 64-wide block column" pass as a
// kernel of its own (no diagonal-block work in it), both MFMA operands staged through LDS by
// global_load_lds (no VGPR staging), three LDS buffers, counted vmcnt + raw s_barrier.
//   workgroup = 4 waves, wave = 2 row tiles x 4 column tiles (32 rows x 64 columns, 128 accumulator VGPRs)
//   k-loop chunk = 8 columns: 4 panel tiles (shared by the waves) + 8 row tiles = 24 KB per buffer
//   after the k-loop: X1 = W00 a1 ; a2 -= conj(L10) X1 ; X2 = W11 a2   (operands staged the same way)
// Storage is the factor's 16-row panel-major layout (HPX_LIDX).  Prints the time of every block
// column's pass at the C3 shape and checks one pass against a host computation.
//   hipcc -O3 --offload-arch=gfx950 tools/tile_pass_probe.hip -o tools/tile_pass_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
#include <vector>
#include <complex>
#ifndef PROBE_3M
#define PROBE_3M 1
#endif

typedef double d4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) double lds_f64;
#define LIDX(r, c, npad) ((((long)((r) >> 4) * (npad) + (c)) << 5) + ((r) & 15))
#define ACC_ROW(g, v) ((g) + 4 * (v))
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

static __device__ __forceinline__ d4 mfma64(double a, double b, d4 c) {
  return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
}
static __device__ __forceinline__ void glds16(const double* src, double* dst) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                   (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
}

constexpr int KC = 8;                 // columns per chunk
constexpr int TILE_CH = KC * 32;      // doubles of one 16-row tile in a chunk
constexpr int BUF_D = 12 * TILE_CH;   // 4 panel tiles + 8 row tiles
constexpr int NBUF = 3;

template <int RT>
__device__ __forceinline__ void tile_pass(double* __restrict__ Lb, const double* __restrict__ Wf,
                                          double* lds, const int npad, const int c0, const int rt0,
                                          const int wave, const int lane, const bool active, const int diag = 0) {
  const int li = lane & 15, g = lane >> 4;
  const long ptile = (long)npad * 32;
  // lane's 16 bytes inside a 1 KB piece (4 columns x [re16 | im16]); odd columns are stored
  // [im | re] so that the two halves of a wave read disjoint banks
  const int cl = lane >> 4, j = lane & 15;
  const int src_lane = cl * 32 + 2 * ((j + 8 * (cl & 1)) & 15);
  const int rd_re = g * 32 + li + 16 * (g & 1), rd_im = g * 32 + li + 16 * (1 - (g & 1));
  const double* pan = Lb + (long)((c0 >> 4) + wave) * ptile + src_lane;
  const double* row0 = Lb + (long)rt0 * ptile + src_lane;
  const double* row1 = Lb + (long)(rt0 + (RT > 1 ? 1 : 0)) * ptile + src_lane;
  d4 ar[RT][4], ai[RT][4];
#if PROBE_3M
  d4 a3[RT][4];       // three-product form: ar = Kre/2 - S1, ai = Kre/2 - S2, a3 = Kim + S3 during the k-loop
#endif
#pragma unroll
  for (int t = 0; t < RT; ++t)
#pragma unroll
    for (int ci = 0; ci < 4; ++ci)
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        const long off = LIDX((rt0 + t) * 16 + li, c0 + 16 * ci + ACC_ROW(g, v), npad);
#if PROBE_3M
        ar[t][ci][v] = 0.5 * Lb[off];
        ai[t][ci][v] = ar[t][ci][v];
        a3[t][ci][v] = Lb[off + 16];
#else
        ar[t][ci][v] = Lb[off];
        ai[t][ci][v] = Lb[off + 16];
#endif
      }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  const int nch = c0 / KC;
#define STAGE(chunk, bufi)                                                        \
  {                                                                               \
    const long ko = (long)(chunk) * TILE_CH;                                      \
    double* bb = lds + (bufi) * BUF_D;                                            \
    glds16(pan + ko, bb + wave * TILE_CH);                                        \
    glds16(pan + ko + 128, bb + wave * TILE_CH + 128);                            \
    glds16(row0 + ko, bb + (4 + 2 * wave) * TILE_CH);                             \
    glds16(row0 + ko + 128, bb + (4 + 2 * wave) * TILE_CH + 128);                 \
    glds16(row1 + ko, bb + (5 + 2 * wave) * TILE_CH);                             \
    glds16(row1 + ko + 128, bb + (5 + 2 * wave) * TILE_CH + 128);                 \
  }
  if (nch > 0) {
    STAGE(0, 0)
    STAGE(1, 1)
    int bi = 0;
    for (int ch = 0; ch < nch; ++ch) {
      asm volatile("s_waitcnt vmcnt(6)" ::: "memory");     // chunk ch has landed (ch+1 may be in flight)
      if (!(diag & 8)) __builtin_amdgcn_s_barrier();       // (diag: timing-only ablations, wrong results)
      const int nx = min(ch + 2, nch - 1);
      int bn = bi + 2; if (bn >= NBUF) bn -= NBUF;
      if (!(diag & 4)) STAGE(nx, bn)
      const lds_f64* B = (const lds_f64*)(lds + bi * BUF_D);
      if (active)
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        double pr[4], pi[4], br[RT], bm[RT];
#pragma unroll
        for (int ci = 0; ci < 4; ++ci) {
          pr[ci] = B[ci * TILE_CH + s * 128 + rd_re];
          pi[ci] = B[ci * TILE_CH + s * 128 + rd_im];
        }
#pragma unroll
        for (int t = 0; t < RT; ++t) {
          br[t] = B[(4 + 2 * wave + t) * TILE_CH + s * 128 + rd_re];
          bm[t] = B[(4 + 2 * wave + t) * TILE_CH + s * 128 + rd_im];
        }
#if PROBE_3M
        double bd[RT];
#pragma unroll
        for (int t = 0; t < RT; ++t) bd[t] = br[t] - bm[t];
#endif
#pragma unroll
        for (int ci = 0; ci < 4; ++ci) {
          const double npr = -pr[ci], npi = -pi[ci];
#if PROBE_3M
          const double psm = pr[ci] + pi[ci];
#endif
#pragma unroll
          for (int t = 0; t < RT; ++t) {
#if PROBE_3M
            ar[t][ci] = mfma64(npr, br[t], ar[t][ci]);
            ai[t][ci] = mfma64(npi, bm[t], ai[t][ci]);
            a3[t][ci] = mfma64(psm, bd[t], a3[t][ci]);
#else
            ar[t][ci] = mfma64(npr, br[t], ar[t][ci]);
            ar[t][ci] = mfma64(npi, bm[t], ar[t][ci]);
            ai[t][ci] = mfma64(npr, bm[t], ai[t][ci]);
            ai[t][ci] = mfma64(pi[ci], br[t], ai[t][ci]);
#endif
          }
        }
      }
      bi = (bi + 1 == NBUF) ? 0 : bi + 1;
    }
  }
#if PROBE_3M
#pragma unroll
  for (int t = 0; t < RT; ++t)
#pragma unroll
    for (int ci = 0; ci < 4; ++ci) {
      const d4 re_ = ar[t][ci] + ai[t][ci], im_ = ar[t][ci] - ai[t][ci] + a3[t][ci];
      ar[t][ci] = re_;
      ai[t][ci] = im_;
    }
#endif
  // ---- block operands: W fragments (24 KB) and the two L10 tiles (16 KB) over the staging buffers
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  {
    // 24 pieces of W, 16 pieces of L10: 10 per wave
    for (int p = wave; p < 24; p += 4) glds16(Wf + p * 128 + 2 * lane, lds + p * 128);
    const double* l10 = Lb + (long)((c0 + 32) >> 4) * ptile + (long)c0 * 32 + src_lane;
    for (int p = wave; p < 16; p += 4)
      glds16(l10 + (long)(p >> 3) * ptile + (p & 7) * 128, lds + 3072 + p * 128);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  const lds_f64* WL = (const lds_f64*)lds;
  const lds_f64* LL = (const lds_f64*)(lds + 3072);
#pragma unroll
  for (int t = 0; t < RT; ++t) {
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      d4 xr[2], xi[2];
#pragma unroll
      for (int ci = 0; ci < 2; ++ci) {
        d4 zr = {0., 0., 0., 0.}, zi = {0., 0., 0., 0.};
#pragma unroll
        for (int cj = 0; cj <= ci; ++cj)
#pragma unroll
          for (int v = 0; v < 4; ++v) {
            const int pair = ci + cj;      // (0,0) -> 0, (1,0) -> 1, (1,1) -> 2
            const lds_f64* w = WL + ((half * 3 + pair) * 4 + v) * 128 + 2 * lane;
            const double wr = w[0], wi = w[1];
            zr = mfma64(wr, ar[t][2 * half + cj][v], zr);
            zr = mfma64(-wi, ai[t][2 * half + cj][v], zr);
            zi = mfma64(wr, ai[t][2 * half + cj][v], zi);
            zi = mfma64(wi, ar[t][2 * half + cj][v], zi);
          }
        xr[ci] = zr;
        xi[ci] = zi;
      }
      __builtin_amdgcn_sched_barrier(0);
      if (half == 0) {
        // a[2..3] -= conj(L10) X1
#pragma unroll
        for (int ci = 0; ci < 2; ++ci)
#pragma unroll
          for (int cj = 0; cj < 2; ++cj)
#pragma unroll
            for (int v = 0; v < 4; ++v) {
              const double pr = LL[ci * 1024 + (4 * cj + v) * 128 + rd_re];
              const double pi = LL[ci * 1024 + (4 * cj + v) * 128 + rd_im];
              ar[t][2 + ci] = mfma64(-pr, xr[cj][v], ar[t][2 + ci]);
              ar[t][2 + ci] = mfma64(-pi, xi[cj][v], ar[t][2 + ci]);
              ai[t][2 + ci] = mfma64(-pr, xi[cj][v], ai[t][2 + ci]);
              ai[t][2 + ci] = mfma64(pi, xr[cj][v], ai[t][2 + ci]);
            }
      }
      if (active) {
        long base = LIDX((rt0 + t) * 16 + li, c0 + 32 * half + g, npad);
        asm volatile("" : "+v"(base));       // recomputed here: not kept alive across the k-loop
        double* o = Lb + base;
#pragma unroll
        for (int ci = 0; ci < 2; ++ci)
#pragma unroll
          for (int v = 0; v < 4; ++v) {
            o[(16 * ci + 4 * v) * 32] = xr[ci][v];
            o[(16 * ci + 4 * v) * 32 + 16] = xi[ci][v];
          }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
#undef STAGE
}

__global__ __launch_bounds__(256, 2) void k_tiles(double* __restrict__ L_all,
                                                  const double* __restrict__ Wf_all, const int npad,
                                                  const int ld, const int c0, const int nw,
                                                  const int nbl, const int prio_mode) {
  extern __shared__ double lds[];
  // workgroups of one baseline share an XCD (its L2 holds the panel rows they all read)
  const int id = blockIdx.x, xcd = id & 7, slot = id >> 3;
  const int b = (slot / nw) * 8 + xcd, wg = slot % nw;
  if (b >= nbl) return;
  const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  double* Lb = L_all + (long)b * npad * ld * 2;
  const double* Wf = Wf_all + (long)b * 3072;
  const int nrt = ld >> 4;
  if (prio_mode & 3) {
    // asymmetric issue priority between the two waves that share a SIMD (wave slot parity): the
    // favoured one streams its MFMAs at full rate, the other fills its gaps, instead of both
    // alternating instruction by instruction and reaching their barriers together
    unsigned hw;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    if ((prio_mode & 3) == 1 ? (hw & 1) : ((blockIdx.x >> 8) & 1)) __builtin_amdgcn_s_setprio(2);
  }
  // tiles spread evenly over the waves of the baseline's workgroups (1 or 2 each)
  const int nt = nrt - ((c0 + 64) >> 4), nwv = 4 * nw, q = wg * 4 + wave;
  const int base = nt / nwv, extra = nt % nwv;
  const int cnt = base + (q < extra ? 1 : 0);
  const int rt0 = ((c0 + 64) >> 4) + q * base + min(q, extra);
  if (cnt >= 2) tile_pass<2>(Lb, Wf, lds, npad, c0, rt0, wave, lane, true, prio_mode & 12);
  else          // one tile, or none: an idle wave still stages (a valid tile) and keeps the barriers
    tile_pass<1>(Lb, Wf, lds, npad, c0, (cnt <= 0) ? nrt - 1 : rt0, wave, lane, cnt > 0, prio_mode & 12);
  // (an idle wave still multiplies: only relevant when a baseline has fewer tiles than waves)
}

int main(int argc, char** argv) {
  const int nbl = argc > 1 ? atoi(argv[1]) : 1024;
  const int npad = argc > 2 ? atoi(argv[2]) : 528, TP = 32, ld = npad + TP;
  const int reps = 5;
  const int prio_mode = argc > 3 ? atoi(argv[3]) : 0;
  const size_t per = (size_t)npad * ld * 2;
  std::vector<double> h(per);
  srand(7);
  for (size_t i = 0; i < per; ++i) h[i] = 0.05 * (2.0 * rand() / RAND_MAX - 1.0);
  // dense lower-triangular W00, W11 (stored form), then the fragment image
  std::vector<std::complex<double>> W(2 * 32 * 32);
  for (int blk = 0; blk < 2; ++blk)
    for (int r = 0; r < 32; ++r)
      for (int c = 0; c < 32; ++c)
        W[(blk * 32 + r) * 32 + c] = (c <= r) ? std::complex<double>(0.02 * (2.0 * rand() / RAND_MAX - 1.0),
                                                                       0.02 * (2.0 * rand() / RAND_MAX - 1.0))
                                              : std::complex<double>(0, 0);
  std::vector<double> wf(3072);
  const int pci[3] = {0, 1, 1}, pcj[3] = {0, 0, 1};
  for (int blk = 0; blk < 2; ++blk)
    for (int p = 0; p < 3; ++p)
      for (int v = 0; v < 4; ++v)
        for (int l = 0; l < 64; ++l) {
          const int li = l & 15, g = l >> 4;
          const std::complex<double> w = W[(blk * 32 + 16 * pci[p] + li) * 32 + 16 * pcj[p] + g + 4 * v];
          wf[(((blk * 3 + p) * 4 + v) * 64 + l) * 2] = w.real();
          wf[(((blk * 3 + p) * 4 + v) * 64 + l) * 2 + 1] = w.imag();
        }
  double *dL, *dW;
  CK(hipMalloc(&dL, per * nbl * sizeof(double)));
  CK(hipMalloc(&dW, (size_t)3072 * nbl * sizeof(double)));
  for (int b = 0; b < nbl; ++b) {
    CK(hipMemcpy(dL + per * b, h.data(), per * sizeof(double), hipMemcpyHostToDevice));
    CK(hipMemcpy(dW + (size_t)3072 * b, wf.data(), 3072 * sizeof(double), hipMemcpyHostToDevice));
  }
  const size_t ldsb = (size_t)NBUF * BUF_D * sizeof(double);
  CK(hipFuncSetAttribute((const void*)k_tiles, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb));
  auto launch = [&](int c0) {
    const int rows_t = (ld >> 4) - ((c0 + 64) >> 4);
    const int nw = (rows_t + 7) / 8;
    const int grid = ((nbl + 7) / 8) * 8 * nw;
    hipLaunchKernelGGL(k_tiles, dim3(grid), dim3(256), ldsb, 0, dL, dW, npad, ld, c0, nw, nbl, prio_mode);
  };
  // ---- correctness: one pass at c0 = 128 on fresh data, baseline 3 against the host
  {
    const int c0 = 128;
    launch(c0);
    CK(hipDeviceSynchronize());
    std::vector<double> o(per);
    CK(hipMemcpy(o.data(), dL + per * 3, per * sizeof(double), hipMemcpyDeviceToHost));
    auto in = [&](int r, int c) { const long off = LIDX(r, c, npad); return std::complex<double>(h[off], h[off + 16]); };
    double maxerr = 0, maxval = 0;
    for (int r = c0 + 64; r < ld; ++r) {
      std::complex<double> a[64], x[64];
      for (int c = 0; c < 64; ++c) {
        std::complex<double> s = in(r, c0 + c);
        for (int k = 0; k < c0; ++k) s -= in(r, k) * std::conj(in(c0 + c, k));
        a[c] = s;
      }
      // the kernel forms X^T = Wstored * acc^T, i.e. x[c] = sum_c' W[c][c'] a[c']
      for (int c = 0; c < 32; ++c) { x[c] = 0; for (int q = 0; q <= c; ++q) x[c] += W[c * 32 + q] * a[q]; }
      for (int c = 0; c < 32; ++c)
        for (int q = 0; q < 32; ++q) a[32 + c] -= std::conj(in(c0 + 32 + c, c0 + q)) * x[q];
      for (int c = 0; c < 32; ++c) { x[32 + c] = 0; for (int q = 0; q <= c; ++q) x[32 + c] += W[(32 + c) * 32 + q] * a[32 + q]; }
      for (int c = 0; c < 64; ++c) {
        const long off = LIDX(r, c0 + c, npad);
        const std::complex<double> got(o[off], o[off + 16]);
        maxerr = fmax(maxerr, std::abs(got - x[c]));
        maxval = fmax(maxval, std::abs(x[c]));
      }
    }
    printf("check c0=128: max abs err %.3e (max |x| %.3e)\n", maxerr, maxval);
  }
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  double tot_ms = 0, tot_fl = 0;
  for (int c0 = 0; c0 + 64 <= npad; c0 += 64) {
    launch(c0);
    CK(hipEventRecord(e0));
    for (int r = 0; r < reps; ++r) launch(c0);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    ms /= reps;
    const double rows = ld - c0 - 64;
    const double fl = 8.0 * rows * 64.0 * c0 * nbl;                    // update flops
    const double fl2 = 8.0 * rows * (2 * 32.0 * 33.0 / 2 + 32 * 32) * nbl;  // triangular multiplies + L10 update
    printf("c0 %4d: %.3f ms  update %.1f TFLOP/s  (+block ops %.1f)\n", c0, ms, fl / ms * 1e-9, (fl + fl2) / ms * 1e-9);
    tot_ms += ms;
    tot_fl += fl + fl2;
  }
  printf("all passes: %.3f ms, %.1f TFLOP/s\n", tot_ms, tot_fl / tot_ms * 1e-9);
  return 0;
}
