// Latency of the one-wave 16 x 16 elimination (elim16w, hpx_factor_tiles.h) on its own: every workgroup (one
// active wave + three idle ones) eliminates the same tile `reps` times; wall_clock64 (100 MHz) around the loop.
// build: hipcc -O3 --offload-arch=gfx950 -I hydra_pspec_amd/csrc -I include tools/experiments/elim/elim16w_probe.hip -o /tmp/elim_probe
#include "hpx_factor_tiles.h"
#include <cstdio>
#include <vector>
#include <cmath>

namespace {
__global__ __launch_bounds__(256, 2) void k_probe_elim(double* L, double* W, double* Vt, const double* tile, int reps,
                                                      long long* ticks, double* sink) {
  WideCtx X;
  X.tid = threadIdx.x; X.wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6); X.lane = threadIdx.x & 63;
  X.npad = 16; X.nct = 1; X.nrt = 1; X.ptile = 16 * 32;
  X.Lb = L + (long)blockIdx.x * 512; X.Vt = Vt + (long)blockIdx.x * 1024; X.Wgre = W + (long)blockIdx.x * 2048;
  X.Wgim = X.Wgre + 1024;
  const int li = X.lane & 15, g = X.lane >> 4;
  d4 re, im;
  for (int v = 0; v < 4; ++v) {
    re[v] = tile[(li * 16 + g + 4 * v) * 2];
    im[v] = tile[(li * 16 + g + 4 * v) * 2 + 1];
  }
  bool bad = false;
  __syncthreads();
  const long long t0 = wall_clock64();
  if (X.wave == 0) {
    for (int r = 0; r < reps; ++r) {
      bad |= elim16w(X, 0, false, re, im);
      asm volatile("" : "+v"(re[0]));
    }
  }
  const long long t1 = wall_clock64();
  if (threadIdx.x == 0) ticks[blockIdx.x] = t1 - t0;
  if (bad) sink[0] = 1.0;
}
}  // namespace

int main(int argc, char** argv) {
  const int nwg = argc > 1 ? atoi(argv[1]) : 256, reps = argc > 2 ? atoi(argv[2]) : 200;
  std::vector<double> t(512);
  for (int i = 0; i < 16; ++i)
    for (int j = 0; j < 16; ++j) {
      t[(i * 16 + j) * 2] = (i == j) ? 20.0 + i : 1.0 / (1 + abs(i - j));
      t[(i * 16 + j) * 2 + 1] = (i == j) ? 0.0 : (i > j ? 0.1 : -0.1) * (1.0 / (1 + abs(i - j)));
    }
  double *L, *W, *Vt, *tile, *sink;
  long long* ticks;
  hipMalloc(&L, (size_t)nwg * 512 * 8); hipMalloc(&W, (size_t)nwg * 2048 * 8); hipMalloc(&Vt, (size_t)nwg * 1024 * 8);
  hipMalloc(&tile, 512 * 8); hipMalloc(&sink, 8); hipMalloc(&ticks, (size_t)nwg * 8);
  hipMemcpy(tile, t.data(), 512 * 8, hipMemcpyHostToDevice);
  for (int rep = 0; rep < 2; ++rep) {
    hipLaunchKernelGGL(k_probe_elim, dim3(nwg), dim3(256), 0, 0, L, W, Vt, tile, reps, ticks, sink);
    hipDeviceSynchronize();
  }
  std::vector<long long> h(nwg);
  hipMemcpy(h.data(), ticks, (size_t)nwg * 8, hipMemcpyDeviceToHost);
  double s = 0, mx = 0;
  for (int i = 0; i < nwg; ++i) { s += h[i]; mx = h[i] > mx ? h[i] : mx; }
  printf("workgroups %d, %d eliminations each: mean %.3f us, max %.3f us per elimination\n", nwg, reps, s / nwg / reps / 100.0,
         mx / reps / 100.0);
  return 0;
}
