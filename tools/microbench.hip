// Micro-benchmarks that size the stencil kernels (run on the GPU box):
//   1. VALU issue rate of v_add_f32 / v_pk_add_f32 / v_add_f32 + DPP wave_shr
//   2. semantics of the wave_shr:1 / wave_shl:1 DPP controls on gfx950
//   3. device copy bandwidth (practical HBM ceiling)
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off microbench.hip -o microbench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("ERR %s line %d: %s\n", #x, __LINE__, hipGetErrorString(e)); exit(1);} } while (0)

typedef float float2v __attribute__((ext_vector_type(2)));
typedef float float4v __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ void __launch_bounds__(256) valu_kernel(float* out, int iters, float seed) {
  float a[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) a[i] = seed * (i + 1) + threadIdx.x;
  float2v p[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) p[i] = float2v{a[2 * i], a[2 * i + 1]};
  for (int it = 0; it < iters; ++it) {
    if (MODE == 0) {   // 16 independent v_add_f32 chains
#pragma unroll
      for (int i = 0; i < 16; ++i) a[i] = a[i] + seed;
    } else if (MODE == 1) {  // 8 independent v_pk_add_f32 chains (16 adds)
#pragma unroll
      for (int i = 0; i < 8; ++i) p[i] = p[i] + float2v{seed, seed};
    } else if (MODE == 2) {  // v_add_f32 with DPP wave_shr:1 source
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        float l = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(
            0, __builtin_bit_cast(int, a[(i + 1) & 15]), 0x138, 0xf, 0xf, false));
        a[i] = a[i] + l;
      }
    } else if (MODE == 3) {  // v_mul then dependent v_add (jacobi-like chain of 5)
#pragma unroll
      for (int i = 0; i < 16; i += 4) {
        float s = a[i] + a[i + 1];
        s = s + a[i + 2];
        s = s + a[i + 3];
        s = s + seed;
        a[i] = s * 0.2f;
      }
    }
  }
  float r = 0;
  if (MODE == 1) {
#pragma unroll
    for (int i = 0; i < 8; ++i) r += p[i].x + p[i].y;
  } else {
#pragma unroll
    for (int i = 0; i < 16; ++i) r += a[i];
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

__global__ void dpp_semantics(int* out) {
  int v = threadIdx.x + 100;
  out[threadIdx.x] = __builtin_amdgcn_update_dpp(-1, v, 0x138, 0xf, 0xf, false);        // wave_shr:1
  out[64 + threadIdx.x] = __builtin_amdgcn_update_dpp(-1, v, 0x130, 0xf, 0xf, false);   // wave_shl:1
  out[128 + threadIdx.x] = __builtin_amdgcn_update_dpp(-1, v, 0x138, 0xf, 0xf, true);   // bound_ctrl
  out[192 + threadIdx.x] = __builtin_amdgcn_update_dpp(-1, v, 0x13C, 0xf, 0xf, false);  // wave_ror:1
  out[256 + threadIdx.x] = __builtin_amdgcn_update_dpp(-1, v, 0x134, 0xf, 0xf, false);  // wave_rol:1
}

__global__ void __launch_bounds__(256) copy_kernel(const float4v* __restrict__ in, float4v* __restrict__ out, size_t n) {
  size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  size_t stride = (size_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) out[i] = in[i];
}

template <int MODE>
double run_valu(const char* name, int waves_per_simd, float* dout) {
  const int cu = 256;
  const int blocks = cu * waves_per_simd;  // 256 threads = 4 waves = 1 per SIMD
  const int iters = 20000;
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  valu_kernel<MODE><<<blocks, 256>>>(dout, 100, 1.0f);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  valu_kernel<MODE><<<blocks, 256>>>(dout, iters, 1.0f);
  CK(hipEventRecord(e1));
  CK(hipDeviceSynchronize());
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  const double ops_per_iter = (MODE == 3) ? 20.0 : 16.0;  // fp32 lane-ops per lane per iter
  double lane_ops = (double)blocks * 256 * iters * ops_per_iter;
  double tops = lane_ops / (ms * 1e-3) / 1e12;
  printf("%-28s waves/SIMD=%d  %.3f ms  %.2f T lane-ops/s  (%.1f lane-ops/clk/CU @2.4GHz)\n",
         name, waves_per_simd, ms, tops, tops * 1e12 / 256 / 2.4e9);
  return tops;
}

int main() {
  float* dout; CK(hipMalloc(&dout, 256 * 8 * 256 * sizeof(float) * 4));
  for (int w : {1, 2, 4, 8}) {
    run_valu<0>("v_add_f32", w, dout);
    run_valu<1>("v_pk_add_f32", w, dout);
    run_valu<2>("v_add_f32 dpp wave_shr", w, dout);
    run_valu<3>("jacobi-like dependent chain", w, dout);
  }
  int* dd; CK(hipMalloc(&dd, 320 * sizeof(int)));
  dpp_semantics<<<1, 64>>>(dd);
  std::vector<int> h(320);
  CK(hipMemcpy(h.data(), dd, 320 * sizeof(int), hipMemcpyDeviceToHost));
  const char* names[5] = {"wave_shr:1", "wave_shl:1", "wave_shr:1 bound_ctrl", "wave_ror:1", "wave_rol:1"};
  for (int m = 0; m < 5; ++m) {
    printf("%-22s lane0=%d lane1=%d lane15=%d lane16=%d lane31=%d lane32=%d lane62=%d lane63=%d\n", names[m],
           h[m*64+0], h[m*64+1], h[m*64+15], h[m*64+16], h[m*64+31], h[m*64+32], h[m*64+62], h[m*64+63]);
  }
  // copy bandwidth
  for (size_t mb : {256, 1024, 4096}) {
    size_t bytes = mb << 20;
    float4v *a, *b; CK(hipMalloc(&a, bytes)); CK(hipMalloc(&b, bytes));
    CK(hipMemset(a, 1, bytes)); CK(hipMemset(b, 0, bytes));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    copy_kernel<<<2048, 256>>>(a, b, bytes / 16);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int r = 0; r < 10; ++r) copy_kernel<<<2048, 256>>>(a, b, bytes / 16);
    CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("copy %zu MiB: %.3f ms/iter, %.2f TB/s (read+write)\n", mb, ms / 10, 2.0 * bytes * 10 / (ms * 1e-3) / 1e12);
    CK(hipFree(a)); CK(hipFree(b));
  }
  return 0;
}
