// Semantics of v_permlane32_swap on gfx950 (used by the 3-D kernels to pass tile
// rows between the two 32-lane halves of a wavefront).
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(int* out) {
  int lane = threadIdx.x;
  unsigned a = 100 + lane, b = 200 + lane;
  auto r = __builtin_amdgcn_permlane32_swap(a, b, false, false);
  out[lane] = r[0]; out[64 + lane] = r[1];
}
int main() {
  int* d; hipMalloc(&d, 128 * sizeof(int));
  k<<<1, 64>>>(d);
  int h[128]; hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
  printf("r[0]: lane0=%d lane31=%d lane32=%d lane63=%d\n", h[0], h[31], h[32], h[63]);
  printf("r[1]: lane0=%d lane31=%d lane32=%d lane63=%d\n", h[64], h[95], h[96], h[127]);
  return 0;
}
