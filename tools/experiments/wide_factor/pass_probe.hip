// Probe of the split factorisation's tile pass (hydra_pspec_amd/csrc/hpx_pass.h) as a kernel of its
// own: per configuration (block-column width, row tiles per wave, k-chunk) one pass is checked against
// a host computation and every block column's pass is timed at the C3 shape.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -Ihydra_pspec_amd/csrc tools/pass_probe.hip -o tools/pass_probe
//   tools/pass_probe [nbl] [npad] [config mask]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
#include <vector>
#include <complex>
#include "hpx_pass.h"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
void hpx_set_error(const char*, ...) {}

template <int CT, int RTMAX, int KC, int DG>
__global__ __launch_bounds__(256, 2) void k_pass(double* __restrict__ L_all, const double* __restrict__ Wf_all,
                                                 const int npad, const int ld, const int c0, const int nw,
                                                 const int nbl, const int nblk32) {
  extern __shared__ double lds[];
  const int id = blockIdx.x, xcd = id & 7, slot = id >> 3;
  const int b = (slot / nw) * 8 + xcd, wg = slot % nw;
  if (b >= nbl) return;
  const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  double* Lb = L_all + (long)b * npad * ld * 2;
  const double* Wf = Wf_all + ((long)b * nblk32 + (c0 >> 5)) * hpx_pass::W_FRAG;
  hpx_gen G = {};
  hpx_pass::pass_workgroup<CT, RTMAX, KC, false, DG>(Lb, Wf, lds, npad, c0, (c0 + 16 * CT) >> 4, ld >> 4, wg, nw,
                                                 wave, lane, G);
}

template <int CT, int RTMAX, int KC, int DG = 0>
static void run(const char* name, int nbl, int npad, int ld, double* dL, const double* dW,
                const std::vector<double>& h, const std::vector<std::complex<double>>& W, int nblk32) {
  typedef hpx_pass::Cfg<CT, RTMAX, KC> C;
  const int CW = 16 * CT;
  const size_t per = (size_t)npad * ld * 2;
  const size_t ldsb = (size_t)C::LDS_D * sizeof(double);
  CK(hipFuncSetAttribute((const void*)k_pass<CT, RTMAX, KC, DG>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb));
  auto launch = [&](int c0) {
    const int rows_t = (ld >> 4) - ((c0 + CW) >> 4);
    const int nw = (rows_t + 4 * RTMAX - 1) / (4 * RTMAX);
    if (nw <= 0) return;
    const int grid = ((nbl + 7) / 8) * 8 * nw;
    hipLaunchKernelGGL((k_pass<CT, RTMAX, KC, DG>), dim3(grid), dim3(256), ldsb, 0, dL, dW, npad, ld, c0, nw, nbl, nblk32);
  };
  auto reset = [&]() {
    for (int b = 0; b < 4 && b < nbl; ++b) CK(hipMemcpy(dL + per * b, h.data(), per * sizeof(double), hipMemcpyHostToDevice));
  };
  printf("== %s: CT %d RTMAX %d KC %d, LDS %zu B\n", name, CT, RTMAX, KC, ldsb);
  // ---- correctness: passes at c0 = 0 and c0 = 2 CW on fresh data, baseline 3 against the host
  for (int c0 : {0, 2 * CW}) {
    if (DG) break;
    reset();
    launch(c0);
    CK(hipDeviceSynchronize());
    std::vector<double> o(per);
    CK(hipMemcpy(o.data(), dL + per * 3, per * sizeof(double), hipMemcpyDeviceToHost));
    auto in = [&](int r, int c) { const long off = HPX_LIDX(r, c, npad); return std::complex<double>(h[off], h[off + 16]); };
    double maxerr = 0, maxval = 0;
    for (int r = c0 + CW; r < ld; ++r) {
      std::vector<std::complex<double>> a(CW), x(CW);
      for (int c = 0; c < CW; ++c) {
        std::complex<double> s = in(r, c0 + c);
        for (int k = 0; k < c0; ++k) s -= in(r, k) * std::conj(in(c0 + c, k));
        a[c] = s;
      }
      for (int s = 0; s < CW / 32; ++s) {
        const std::complex<double>* Ws = &W[(size_t)((c0 >> 5) + s) * 1024];
        for (int c = 0; c < 32; ++c) { x[32 * s + c] = 0; for (int q = 0; q <= c; ++q) x[32 * s + c] += Ws[c * 32 + q] * a[32 * s + q]; }
        for (int c = 32 * (s + 1); c < CW; ++c)
          for (int q = 0; q < 32; ++q) a[c] -= std::conj(in(c0 + c, c0 + 32 * s + q)) * x[32 * s + q];
      }
      for (int c = 0; c < CW; ++c) {
        const long off = HPX_LIDX(r, c0 + c, npad);
        const std::complex<double> got(o[off], o[off + 16]);
        maxerr = fmax(maxerr, std::abs(got - x[c]));
        maxval = fmax(maxval, std::abs(x[c]));
      }
    }
    printf("check c0=%d: max abs err %.3e (max |x| %.3e)\n", c0, maxerr, maxval);
  }
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  const int reps = 4;
  double tot_ms = 0, tot_fl = 0;
  for (int c0 = 0; c0 + CW <= npad; c0 += CW) {
    if ((ld >> 4) - ((c0 + CW) >> 4) <= 0) break;
    launch(c0);
    CK(hipEventRecord(e0));
    for (int r = 0; r < reps; ++r) launch(c0);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    ms /= reps;
    const double rows = ld - c0 - CW;
    const double fl = 8.0 * rows * CW * c0 * nbl + 8.0 * rows * (CW * (CW + 32.0) / 2) * nbl;
    printf("c0 %4d: %.3f ms  %.1f TFLOP/s\n", c0, ms, fl / ms * 1e-9);
    tot_ms += ms;
    tot_fl += fl;
  }
  printf("all passes (%s): %.3f ms, %.1f TFLOP/s\n", name, tot_ms, tot_fl / tot_ms * 1e-9);
  CK(hipGetLastError());
}

int main(int argc, char** argv) {
  const int nbl = argc > 1 ? atoi(argv[1]) : 1024;
  const int npad = argc > 2 ? atoi(argv[2]) : 528, TP = 32, ld = npad + TP;
  const int mask = argc > 3 ? atoi(argv[3]) : 7;
  const size_t per = (size_t)npad * ld * 2;
  std::vector<double> h(per);
  srand(7);
  for (size_t i = 0; i < per; ++i) h[i] = 0.05 * (2.0 * rand() / RAND_MAX - 1.0);
  const int nblk32 = (npad + 31) / 32;
  std::vector<std::complex<double>> W((size_t)nblk32 * 1024);
  for (int blk = 0; blk < nblk32; ++blk)
    for (int r = 0; r < 32; ++r)
      for (int c = 0; c < 32; ++c)
        W[((size_t)blk * 32 + r) * 32 + c] = (c <= r) ? std::complex<double>(0.02 * (2.0 * rand() / RAND_MAX - 1.0),
                                                                              0.02 * (2.0 * rand() / RAND_MAX - 1.0))
                                                      : std::complex<double>(0, 0);
  std::vector<double> wf((size_t)nblk32 * hpx_pass::W_FRAG);
  const int pci[3] = {0, 1, 1}, pcj[3] = {0, 0, 1};
  for (int blk = 0; blk < nblk32; ++blk)
    for (int p = 0; p < 3; ++p)
      for (int v = 0; v < 4; ++v)
        for (int l = 0; l < 64; ++l) {
          const int li = l & 15, g = l >> 4;
          const std::complex<double> w = W[((size_t)blk * 32 + 16 * pci[p] + li) * 32 + 16 * pcj[p] + g + 4 * v];
          wf[(size_t)blk * hpx_pass::W_FRAG + (((p * 4 + v) * 64 + l) * 2)] = w.real();
          wf[(size_t)blk * hpx_pass::W_FRAG + (((p * 4 + v) * 64 + l) * 2) + 1] = w.imag();
        }
  double *dL, *dW;
  CK(hipMalloc(&dL, per * nbl * sizeof(double)));
  CK(hipMalloc(&dW, wf.size() * nbl * sizeof(double)));
  for (int b = 0; b < nbl; ++b) {
    CK(hipMemcpy(dL + per * b, h.data(), per * sizeof(double), hipMemcpyHostToDevice));
    CK(hipMemcpy(dW + wf.size() * b, wf.data(), wf.size() * sizeof(double), hipMemcpyHostToDevice));
  }
  if (mask & 1) run<4, 2, 8>("w64 2x4", nbl, npad, ld, dL, dW, h, W, nblk32);
  if (mask & 2) run<8, 1, 8>("w128 1x8", nbl, npad, ld, dL, dW, h, W, nblk32);
  if (mask & 4) run<8, 1, 4>("w128 1x8 KC4", nbl, npad, ld, dL, dW, h, W, nblk32);
  if (mask & 8) {     // timing-only ablations (wrong results)
    run<8, 1, 8, 1>("w128 1x8 no staging", nbl, npad, ld, dL, dW, h, W, nblk32);
    run<8, 1, 8, 3>("w128 1x8 no staging, no barriers", nbl, npad, ld, dL, dW, h, W, nblk32);
    run<8, 1, 8, 4>("w128 1x8 no k-loop MFMAs", nbl, npad, ld, dL, dW, h, W, nblk32);
    run<8, 1, 8, 8>("w128 1x8 no triangular solve", nbl, npad, ld, dL, dW, h, W, nblk32);
    run<8, 1, 8, 11>("w128 1x8 MFMA loop only", nbl, npad, ld, dL, dW, h, W, nblk32);
  }

  return 0;
}
