// Timing probe (round 3, for the next round): the tile passes of the left-looking factorisation with operands in
// REGISTERS only, at two blockings --
//   W32: 32-wide block columns, a wave sweeps 3 row tiles x 2 column tiles (18 MFMAs per k-step, 144 accumulator
//        VGPRs), two workgroups per CU: the shipped kernel's k-loop;
//   W64: 64-wide block columns, 3 row tiles x 4 column tiles (36 MFMAs per k-step, 288 accumulator registers),
//        one workgroup per CU, one wave per SIMD with the 512-register budget: half the row-tile traffic, and a k-step
//        long enough (36 x 64 cycles) to cover a memory round trip with one k-step of prefetch.
// Passes only (no diagonal blocks, no triangular solve: the accumulators are stored as they are), random data,
// C3's shape.  usage: pass64_probe [nbl]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef double d4 __attribute__((ext_vector_type(4)));
#define LIDX(r, c, npad) ((((long)((r) >> 4) * (npad) + (c)) << 5) + ((r) & 15))
#define ACC_ROW(g, v) ((g) + 4 * (v))
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
__device__ __forceinline__ d4 mfma64(double a, double b, d4 c) { return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0); }

template <int CT, int RT, int KC>
__device__ __forceinline__ void sweep(double* __restrict__ Lre, double* __restrict__ Lim, const int npad, const int c0,
                                      const int r0, const int rstride, const int lane) {
  const int li = lane & 15, g = lane >> 4;
  d4 a1[RT][CT], a2[RT][CT], a3[RT][CT];
#pragma unroll
  for (int t = 0; t < RT; ++t)
#pragma unroll
    for (int ci = 0; ci < CT; ++ci) a1[t][ci] = a2[t][ci] = a3[t][ci] = (d4){0., 0., 0., 0.};
  const int nch = (c0 >> 2) / KC;                // chunks of KC k-steps (c0 is a multiple of 32: even for KC | 4)
  const double* pre = Lre + (long)g * 32;
  const double* pim = Lim + (long)g * 32;
  const long kstep = 128, ptile = (long)npad * 32;
  const long boff = (long)(r0 >> 4) * ptile + li, bstr = (long)(rstride >> 4) * ptile;
  const long aoff = (long)(c0 >> 4) * ptile + li;
  double b0r[KC][RT], b0i[KC][RT], b1r[KC][RT], b1i[KC][RT], p0r[KC][CT], p0i[KC][CT], p1r[KC][CT], p1i[KC][CT];
#define LOADC(br_, bi_, pr_, pi_, base_re, base_im)                                         \
  _Pragma("unroll") for (int s = 0; s < KC; ++s) {                                          \
  _Pragma("unroll") for (int t = 0; t < RT; ++t) {                                          \
    br_[s][t] = __builtin_nontemporal_load(&(base_re)[s * kstep + boff + t * bstr]);        \
    bi_[s][t] = __builtin_nontemporal_load(&(base_im)[s * kstep + boff + t * bstr]);        \
  }                                                                                         \
  _Pragma("unroll") for (int ci = 0; ci < CT; ++ci) {                                       \
    pr_[s][ci] = (base_re)[s * kstep + aoff + ci * ptile];                                  \
    pi_[s][ci] = (base_im)[s * kstep + aoff + ci * ptile];                                  \
  } }
#define MMAC(br_, bi_, pr_, pi_)                                                            \
  _Pragma("unroll") for (int s = 0; s < KC; ++s) {                                          \
    double bd_[RT];                                                                         \
    _Pragma("unroll") for (int t = 0; t < RT; ++t) bd_[t] = br_[s][t] - bi_[s][t];          \
    _Pragma("unroll") for (int ci = 0; ci < CT; ++ci) {                                     \
      const double npr = -pr_[s][ci], npi = -pi_[s][ci], psm = pr_[s][ci] + pi_[s][ci];     \
      _Pragma("unroll") for (int t = 0; t < RT; ++t) {                                      \
        a1[t][ci] = mfma64(npr, br_[s][t], a1[t][ci]);                                      \
        a2[t][ci] = mfma64(npi, bi_[s][t], a2[t][ci]);                                      \
        a3[t][ci] = mfma64(psm, bd_[t], a3[t][ci]);                                         \
      }                                                                                     \
    }                                                                                       \
  }
  if (nch > 0) {
    LOADC(b0r, b0i, p0r, p0i, pre, pim)
    for (int ch = 0; ch < nch; ch += 2) {
      const double* qre = pre + KC * kstep;
      const double* qim = pim + KC * kstep;
      LOADC(b1r, b1i, p1r, p1i, qre, qim)
      __builtin_amdgcn_sched_barrier(0);
      MMAC(b0r, b0i, p0r, p0i)
      __builtin_amdgcn_sched_barrier(0);
      const long adv = (ch + 2 < nch) ? 2 * KC * kstep : 0;
      pre += adv;
      pim += adv;
      LOADC(b0r, b0i, p0r, p0i, pre, pim)
      __builtin_amdgcn_sched_barrier(0);
      MMAC(b1r, b1i, p1r, p1i)
      __builtin_amdgcn_sched_barrier(0);
    }
  }
#pragma unroll
  for (int t = 0; t < RT; ++t)
#pragma unroll
    for (int ci = 0; ci < CT; ++ci)
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        const long off = LIDX(r0 + t * rstride + li, c0 + 16 * ci + ACC_ROW(g, v), npad);
        Lre[off] = a1[t][ci][v] + a2[t][ci][v];
        Lim[off] = a1[t][ci][v] - a2[t][ci][v] + a3[t][ci][v];
      }
}

template <int CT, int RT, int WPC, int KC>
__global__ __launch_bounds__(256, WPC) void k_pass(double* __restrict__ L_all, const int npad, const int ld) {
  extern __shared__ double lds_[];
  const int b = blockIdx.x, wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  double* Lre = L_all + (long)b * npad * ld * 2;
  double* Lim = Lre + 16;
  const int nrt = ld >> 4, W = 16 * CT;
  for (int c0 = 0; c0 + W <= npad; c0 += W) {
    int rt = ((c0 + W) >> 4) + wave;
    int cnt = (nrt - rt + 3) / 4;
    if (nrt <= rt) cnt = 0;
    while (cnt > 0) {                          // groups of RT strided tiles (a shorter last group sweeps RT too,
      const int r0 = min(rt, nrt - 4 * (RT - 1) - 1) << 4;    //  clamped into range: timing only)
      sweep<CT, RT, KC>(Lre, Lim, npad, c0, r0, 64, lane);
      rt += 4 * RT;
      cnt -= RT;
    }
    __syncthreads();
  }
  if (lds_[threadIdx.x] == 12345.678) Lre[0] = 1.0;     // (keeps the dynamic LDS request alive)
}

int main(int argc, char** argv) {
  const int nbl = argc > 1 ? atoi(argv[1]) : 1024, npad = 528, ld = 560;
  const size_t per = (size_t)npad * ld * 2;
  double* L;
  CK(hipMalloc(&L, per * nbl * sizeof(double)));
  std::vector<double> h(per);
  for (size_t i = 0; i < per; ++i) h[i] = 1e-3 * ((double)rand() / RAND_MAX - 0.5);
  for (int b = 0; b < nbl; ++b) CK(hipMemcpy(L + b * per, h.data(), per * sizeof(double), hipMemcpyHostToDevice));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  // algorithmic flops of the passes: every tile (r, c) below the diagonal block of its column accumulates over k < c0
  auto run = [&](const char* name, auto kern, size_t lds, int width) {
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    for (int rep = 0; rep < 3; ++rep) {
      CK(hipEventRecord(e0));
      hipLaunchKernelGGL(kern, dim3(nbl), dim3(256), lds, 0, L, npad, ld);
      CK(hipEventRecord(e1));
      CK(hipEventSynchronize(e1));
      float ms;
      CK(hipEventElapsedTime(&ms, e0, e1));
      double fl = 0;
      for (int c0 = 0; c0 + width <= npad; c0 += width) fl += 8.0 * (double)(ld - c0 - width) * width * c0;
      printf("%s rep %d: %.3f ms for %d baselines; passes' algorithmic flops %.3g per baseline -> %.1f TFLOP/s\n", name,
             rep, ms, nbl, fl, fl * nbl / ms * 1e-9);
    }
  };
  run("W32 (3 x 2 tiles, 2 workgroups / CU, 1 k-step ahead)", k_pass<2, 3, 2, 1>, 60 * 1024, 32);
  run("W32 (3 x 2 tiles, 2 workgroups / CU, 2 k-steps ahead)", k_pass<2, 3, 2, 2>, 60 * 1024, 32);
  run("W32 (3 x 2 tiles, 1 workgroup / CU, 1 k-step ahead) ", k_pass<2, 3, 2, 1>, 100 * 1024, 32);
  run("W64 (2 x 4 tiles, 1 workgroup / CU, 1 k-step ahead) ", k_pass<4, 2, 1, 1>, 100 * 1024, 64);
  run("W64 (2 x 4 tiles, 1 workgroup / CU, 2 k-steps ahead)", k_pass<4, 2, 1, 2>, 100 * 1024, 64);
  run("W64 (2 x 4 tiles, 1 workgroup / CU, 4 k-steps ahead)", k_pass<4, 2, 1, 4>, 100 * 1024, 64);
  run("W32 (3 x 2 tiles, 1 workgroup / CU, 512 regs, 4 k-steps ahead)", k_pass<2, 3, 1, 4>, 100 * 1024, 32);
  return 0;
}
