// Occupancy of k_backsolve_x as the runtime sees it (workgroups per CU, registers, LDS, scratch):
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -Wno-unused-function -Ihydra_pspec_amd/csrc \
//         -o /tmp/occ_probe tools/experiments/ab/occ_probe.hip && /tmp/occ_probe
#include "hpx_backsolve_lds.hip"
#include <cstdio>
void hpx_set_error(const char*, ...) {}
int main() {
  int n = -1;
  hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k_backsolve_x, 512, 0);
  printf("k_backsolve_x: rc %d, workgroups per CU %d\n", (int)e, n);
  hipFuncAttributes a;
  e = hipFuncGetAttributes(&a, (const void*)k_backsolve_x);
  printf("regs %d shared %zu local %zu maxthreads %d\n", a.numRegs, a.sharedSizeBytes, a.localSizeBytes, a.maxThreadsPerBlock);
  return 0;
}
