// What a DEPENDENT chain of v_add_f32 costs a wavefront, against the same number of
// adds spread over 2 / 4 / 8 independent accumulators, at 1..4 wavefronts per SIMD
// (run on the GPU box).  The fused 3-D kernels were found serialised on one
// accumulator register by the default scheduler (tools/isa_stats.py: 56 % of the
// VALU stream back-to-back dependent); this prices that.
// Build: hipcc --offload-arch=gfx950 -O3 depbench.hip -o depbench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("ERR %s line %d: %s\n", #x, __LINE__, hipGetErrorString(e)); exit(1);} } while (0)

// CHAINS independent accumulators, 64 adds per loop trip in round-robin order;
// volatile asm keeps the order as written.  DPP = 1: every 8th add of a chain
// takes its operand through a wave shift (v_add_f32_dpp), as the stencil rows do.
template <int CHAINS, int DPP>
__global__ void __launch_bounds__(256) dep(float* out, int iters, float seed) {
  float acc[CHAINS];
#pragma unroll
  for (int k = 0; k < CHAINS; ++k) acc[k] = seed * (k + 1) + threadIdx.x;
  float x = seed * 0.5f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 64 / CHAINS; ++i) {
#pragma unroll
      for (int k = 0; k < CHAINS; ++k) {
        if (DPP && (i & 7) == 7)
          asm volatile("s_nop 1\n\tv_add_f32_dpp %0, %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1"
                       : "+v"(acc[k]) : "v"(x));
        else
          asm volatile("v_add_f32 %0, %1, %0" : "+v"(acc[k]) : "v"(x));
      }
    }
  }
  float r = 0;
#pragma unroll
  for (int k = 0; k < CHAINS; ++k) r += acc[k];
  out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

template <int CHAINS, int DPP>
void run(int waves_per_simd, float* dout) {
  const int blocks = 256 * waves_per_simd;   // 256-thread blocks: one wavefront per SIMD each
  const int iters = 20000;
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  dep<CHAINS, DPP><<<blocks, 256>>>(dout, 100, 1.0f);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  dep<CHAINS, DPP><<<blocks, 256>>>(dout, iters, 1.0f);
  CK(hipEventRecord(e1));
  CK(hipDeviceSynchronize());
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  const double tops = (double)blocks * 256 * iters * 64.0 / (ms * 1e-3) / 1e12;
  // cycles per instruction per wavefront at an assumed 2.1 GHz
  const double cyc = ms * 1e-3 * 2.1e9 / (iters * 64.0);
  printf("chains=%d dpp=%d waves/SIMD=%d %8.3f ms %7.2f T lane-ops/s  ~%.2f cyc/instr/wave @2.1GHz\n",
         CHAINS, DPP, waves_per_simd, ms, tops, cyc);
}

int main() {
  float* dout; CK(hipMalloc(&dout, 256 * 8 * 256 * sizeof(float)));
  for (int w : {1, 2, 3, 4}) {
    run<1, 0>(w, dout); run<2, 0>(w, dout); run<4, 0>(w, dout); run<8, 0>(w, dout);
    run<1, 1>(w, dout); run<4, 1>(w, dout);
  }
  return 0;
}
