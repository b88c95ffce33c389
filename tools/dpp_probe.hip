// Prints which source lane each DPP control used by ibs_wave.hpp reads (run on the GPU box).
#include <hip/hip_runtime.h>
#include <cstdio>
template <int CTRL, int RM>
__global__ void probe(int* out) {
  int lane = threadIdx.x;
  out[lane] = __builtin_amdgcn_update_dpp(-1, lane, CTRL, RM, 0xF, false);
}
template <int CTRL, int RM>
void run(const char* name) {
  int* d; int h[64];
  hipMalloc(&d, 64 * sizeof(int));
  hipLaunchKernelGGL((probe<CTRL, RM>), dim3(1), dim3(64), 0, 0, d);
  hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  printf("%-22s:", name);
  for (int i = 0; i < 64; ++i) printf(" %d", h[i]);
  printf("\n");
  hipFree(d);
}
int main() {
  run<0x111, 0xF>("row_shr:1");
  run<0x118, 0xF>("row_shr:8");
  run<0x101, 0xF>("row_shl:1");
  run<0x138, 0xF>("wave_shr:1");
  run<0x130, 0xF>("wave_shl:1");
  run<0x142, 0xA>("row_bcast15 rm=0xA");
  run<0x143, 0xC>("row_bcast31 rm=0xC");
  return 0;
}
