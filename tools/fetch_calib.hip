// FETCH_SIZE calibration for k_factor's load shape ("What's weak" 2).
//
// The microarchitecture guide calibrates "FETCH_SIZE reads 1/2 of the bytes" for 16 B/lane
// streams only.  k_factor's k-loop loads are 8 B/lane: lane (li, g) of a wave reads the double at
// panel[(4 ks + g) * 32 + li] (real halves) and + 16 (imaginary halves), i.e. four 128-byte
// segments 256 bytes apart per wave-instruction, a whole 1 KiB of 4 columns per (re, im) pair.
// This program streams a buffer far larger than the Infinity Cache (1 GiB) ONCE with
//   shape 0: exactly that pattern (one 16-row panel strip per wave, k-steps in order),
//   shape 1: 16 B/lane contiguous (the guide's calibrated case, as a control),
//   shape 2: 8 B/lane contiguous (512 B per wave-instruction),
//   shape 3: k_backsolve's pattern (two double2 per lane: 32 contiguous bytes per lane),
// each as its own kernel so that `rocprofv3 --pmc FETCH_SIZE` gives one row per shape:
//   ratio = bytes_streamed / (FETCH_SIZE * 1024).
// Build: hipcc -O3 --offload-arch=gfx950 tools/fetch_calib.hip -o tools/fetch_calib_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

constexpr int NCOL = 4096;                   // columns per panel strip: 4096 * 256 B = 1 MiB
constexpr long PANEL = (long)NCOL * 32;      // doubles per strip

__global__ __launch_bounds__(256) void k_shape_factor(const double* __restrict__ buf, double* __restrict__ out) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, li = lane & 15, g = lane >> 4;
  const double* p = buf + ((long)blockIdx.x * 4 + wave) * PANEL + (long)g * 32 + li;
  double sr = 0.0, si = 0.0;
#pragma unroll 8
  for (int ks = 0; ks < NCOL / 4; ++ks) {
    sr += p[(long)ks * 128];
    si += p[(long)ks * 128 + 16];
  }
  out[(long)blockIdx.x * 256 + threadIdx.x] = sr + si;
}

__global__ __launch_bounds__(256) void k_shape_16B(const double2* __restrict__ buf, double* __restrict__ out) {
  const long per_wave = PANEL / 2;           // double2 per wave
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const double2* p = buf + ((long)blockIdx.x * 4 + wave) * per_wave + lane;
  double s = 0.0;
#pragma unroll 8
  for (long i = 0; i < per_wave / 64; ++i) {
    const double2 v = p[i * 64];
    s += v.x + v.y;
  }
  out[(long)blockIdx.x * 256 + threadIdx.x] = s;
}

__global__ __launch_bounds__(256) void k_shape_8B(const double* __restrict__ buf, double* __restrict__ out) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const double* p = buf + ((long)blockIdx.x * 4 + wave) * PANEL + lane;
  double s = 0.0;
#pragma unroll 8
  for (long i = 0; i < PANEL / 64; ++i) s += p[i * 64];
  out[(long)blockIdx.x * 256 + threadIdx.x] = s;
}

// k_backsolve: lane (li, g) reads rows 4g..4g+3 of column c0 + li: double2 at off and off + 2
// (re), the same + 16 (im), with off = (c * 32 + 4 g): 32-byte pieces, 256 B apart across li.
__global__ __launch_bounds__(256) void k_shape_backsolve(const double* __restrict__ buf, double* __restrict__ out) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, li = lane & 15, g = lane >> 4;
  const double* p = buf + ((long)blockIdx.x * 4 + wave) * PANEL + (long)li * 32 + 4 * g;
  double s = 0.0;
#pragma unroll 4
  for (int cb = 0; cb < NCOL / 16; ++cb) {       // 16 columns (4 KiB) per step
    const double* q = p + (long)cb * 16 * 32;
    const double2 a0 = *reinterpret_cast<const double2*>(q);
    const double2 a1 = *reinterpret_cast<const double2*>(q + 2);
    const double2 b0 = *reinterpret_cast<const double2*>(q + 16);
    const double2 b1 = *reinterpret_cast<const double2*>(q + 18);
    s += a0.x + a0.y + a1.x + a1.y + b0.x + b0.y + b1.x + b1.y;
  }
  out[(long)blockIdx.x * 256 + threadIdx.x] = s;
}

int main() {
  const int nblk = 256;                                 // 1024 waves, one strip each
  const size_t bytes = (size_t)nblk * 4 * PANEL * sizeof(double);   // 1 GiB
  double *buf = nullptr, *out = nullptr, *flush = nullptr;
  CK(hipMalloc(&buf, bytes));
  CK(hipMalloc(&flush, bytes));
  CK(hipMalloc(&out, (size_t)nblk * 256 * sizeof(double)));
  CK(hipMemset(buf, 0, bytes));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  const char* names[4] = {"k_shape_factor (8 B/lane, 4 x 128 B segments)", "k_shape_16B (16 B/lane contiguous)",
                          "k_shape_8B (8 B/lane contiguous)", "k_shape_backsolve (2 x 16 B/lane, 32 B pieces)"};
  for (int rep = 0; rep < 2; ++rep)
    for (int s = 0; s < 4; ++s) {
      CK(hipMemset(flush, rep + 1, bytes));             // push `buf` out of the Infinity Cache
      CK(hipDeviceSynchronize());
      CK(hipEventRecord(e0));
      if (s == 0) hipLaunchKernelGGL(k_shape_factor, dim3(nblk), dim3(256), 0, 0, buf, out);
      else if (s == 1) hipLaunchKernelGGL(k_shape_16B, dim3(nblk), dim3(256), 0, 0, (const double2*)buf, out);
      else if (s == 2) hipLaunchKernelGGL(k_shape_8B, dim3(nblk), dim3(256), 0, 0, buf, out);
      else hipLaunchKernelGGL(k_shape_backsolve, dim3(nblk), dim3(256), 0, 0, buf, out);
      CK(hipGetLastError());
      CK(hipEventRecord(e1));
      CK(hipEventSynchronize(e1));
      float ms = 0.f;
      CK(hipEventElapsedTime(&ms, e0, e1));
      printf("rep %d  %-52s bytes_streamed %zu  %.3f ms  %.2f TB/s\n", rep, names[s], bytes, ms, bytes / ms / 1e9);
    }
  printf("expected KB per dispatch if FETCH_SIZE were exact: %.0f\n", bytes / 1024.0);
  hipFree(buf); hipFree(flush); hipFree(out);
  return 0;
}
