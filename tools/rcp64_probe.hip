// Accuracy of v_rcp_f64 / v_rsq_f64 and of Newton-refined forms against IEEE 1/x, 1/sqrt(x).
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
__global__ void k(const double* x, double* o, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const double d = x[i];
  double r0 = __builtin_amdgcn_rcp(d);
  double e = fma(-d, r0, 1.0);
  double r1 = fma(r0, e, r0);
  e = fma(-d, r1, 1.0);
  double r2 = fma(r1, e, r1);
  double q0 = __builtin_amdgcn_rsq(d);
  // one Newton step for 1/sqrt: q1 = q0 + q0 * (0.5 * (1 - d q0^2))
  double h = fma(-d * q0, q0, 1.0);
  double q1 = fma(q0 * 0.5, h, q0);
  h = fma(-d * q1, q1, 1.0);
  double q2 = fma(q1 * 0.5, h, q1);
  o[i * 6 + 0] = r0; o[i * 6 + 1] = r1; o[i * 6 + 2] = r2;
  o[i * 6 + 3] = q0; o[i * 6 + 4] = q1; o[i * 6 + 5] = q2;
}
int main() {
  const int n = 1 << 20;
  double* hx = (double*)malloc(n * 8); double* ho = (double*)malloc(n * 48);
  srand(1);
  for (int i = 0; i < n; ++i) hx[i] = exp(40.0 * (rand() / (double)RAND_MAX - 0.5)) * (1.0 + rand() / (double)RAND_MAX);
  double *dx, *dob; (void)hipMalloc(&dx, n * 8); (void)hipMalloc(&dob, n * 48);
  (void)hipMemcpy(dx, hx, n * 8, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, dx, dob, n);
  (void)hipMemcpy(ho, dob, n * 48, hipMemcpyDeviceToHost);
  double m[6] = {0, 0, 0, 0, 0, 0};
  for (int i = 0; i < n; ++i) {
    const double r = 1.0 / hx[i], q = 1.0 / sqrt(hx[i]);
    for (int j = 0; j < 3; ++j) m[j] = fmax(m[j], fabs(ho[i * 6 + j] / r - 1.0));
    for (int j = 3; j < 6; ++j) m[j] = fmax(m[j], fabs(ho[i * 6 + j] / q - 1.0));
  }
  printf("max rel err: rcp %.3e, +1 NR %.3e, +2 NR %.3e | rsq %.3e, +1 NR %.3e, +2 NR %.3e\n", m[0], m[1], m[2], m[3], m[4], m[5]);
  return 0;
}
