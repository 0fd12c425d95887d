// Does the raw-buffer range check of gfx950 include the scalar offset?  (Run on the GPU box.)
// A 1 KiB record count over a 16 KiB array of ones; loads at voffset 0 with soffset 0 /
// 4096 and at voffset 4096 with soffset 0: a checked load returns 0.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(const float* p, float* o) {
  __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)p, 0, 1024, 0x27000);
  o[0] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, 0, 0, 0));
  o[1] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, 0, 4096, 0));
  o[2] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, 4096, 0, 0));
  o[3] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, 1020, 0, 0));
  o[4] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, 1020, 8, 0));
}
int main() {
  float *p, *o, h[4096], r[8];
  for (int i = 0; i < 4096; ++i) h[i] = 1.0f;
  hipMalloc(&p, sizeof h); hipMalloc(&o, sizeof r);
  hipMemcpy(p, h, sizeof h, hipMemcpyHostToDevice);
  k<<<1, 1>>>(p, o);
  hipMemcpy(r, o, sizeof r, hipMemcpyDeviceToHost);
  printf("voff 0 soff 0: %g | voff 0 soff 4096: %g | voff 4096 soff 0: %g | voff 1020: %g | voff 1020 soff 8: %g\n",
         r[0], r[1], r[2], r[3], r[4]);
  return 0;
}
