// VALU issue-rate probes with the operand patterns of the fused stencil kernels
// (run on the GPU box).  Each mode reports lane-ops/s and the shader clock seen
// inside the kernel (s_memtime ticks per s_memrealtime tick, 100 MHz).
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fno-slp-vectorize \
//        valubench.hip -o valubench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("ERR %s line %d: %s\n", #x, __LINE__, hipGetErrorString(e)); exit(1);} } while (0)

typedef float pk2 __attribute__((ext_vector_type(2)));

__device__ inline int shr1(int v) { return __builtin_amdgcn_update_dpp(0, v, 0x138, 0xf, 0xf, true); }
__device__ inline pk2 shr1(pk2 v) {
  struct h2 { int lo, hi; };
  h2 h = __builtin_bit_cast(h2, v);
  h.lo = shr1(h.lo); h.hi = shr1(h.hi);
  return __builtin_bit_cast(pk2, h);
}

// MODE 0: scalar add, both sources VGPRs      a[i] += a[i+1]
// MODE 1: packed add, both sources VGPR pairs p[i] += p[i+1]
// MODE 2: packed add, one SGPR-pair source    p[i] += {s,s}
// MODE 3: packed jacobi row: 4 cells x (4 pk_add + 1 pk_mul) + 2 dpp pair moves
// MODE 4: scalar jacobi row: 4 cells x (4 add + 1 mul), dpp fused
// MODE 5: packed fma, three VGPR pair sources
template <int MODE>
__global__ void __launch_bounds__(256) valu(float* out, long long* clocks, int iters, float seed) {
  pk2 p[16];
  float a[32];
#pragma unroll
  for (int i = 0; i < 32; ++i) a[i] = seed * (i + 1) + threadIdx.x;
#pragma unroll
  for (int i = 0; i < 16; ++i) p[i] = pk2{a[2 * i], a[2 * i + 1]};
  const long long t0 = __builtin_readcyclecounter();
  const long long r0 = __builtin_amdgcn_s_memrealtime();
#pragma unroll 2
  for (int it = 0; it < iters; ++it) {
    if (MODE == 0) {
#pragma unroll
      for (int i = 0; i < 16; ++i) a[i] = a[i] + a[16 + ((i + 1) & 15)];
    } else if (MODE == 1) {
#pragma unroll
      for (int i = 0; i < 8; ++i) p[i] = p[i] + p[8 + ((i + 1) & 7)];
    } else if (MODE == 2) {
#pragma unroll
      for (int i = 0; i < 8; ++i) p[i] = p[i] + pk2{seed, seed};
    } else if (MODE == 3) {
      // three rows of 4 packed cells rotate through p[0..11]; 3 steps per trip
      auto step = [&](int up, int mid, int down) {
        const pk2 left = shr1(p[mid + 3]);
        const pk2 right = shr1(p[mid]);
        pk2 o[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const pk2 l = c == 0 ? left : p[mid + c - 1];
          const pk2 r = c == 3 ? right : p[mid + c + 1];
          o[c] = ((((p[up + c] + r) + p[mid + c]) + p[down + c]) + l) * 0.2f;
        }
#pragma unroll
        for (int c = 0; c < 4; ++c) p[up + c] = o[c];
      };
      step(0, 4, 8); step(4, 8, 0); step(8, 0, 4);
    } else if (MODE == 4) {
#pragma unroll
      for (int half = 0; half < 2; ++half) {
        float* q = a + 16 * half;
        auto step = [&](int up, int mid, int down) {
          const float left = __builtin_bit_cast(float, shr1(__builtin_bit_cast(int, q[mid + 3])));
          const float right = __builtin_bit_cast(float, shr1(__builtin_bit_cast(int, q[mid])));
          float o[4];
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            const float l = c == 0 ? left : q[mid + c - 1];
            const float r = c == 3 ? right : q[mid + c + 1];
            o[c] = ((((q[up + c] + r) + q[mid + c]) + q[down + c]) + l) * 0.2f;
          }
#pragma unroll
          for (int c = 0; c < 4; ++c) q[up + c] = o[c];
        };
        step(0, 4, 8); step(4, 8, 0); step(8, 0, 4);
      }
    } else if (MODE == 5) {
#pragma unroll
      for (int i = 0; i < 8; ++i)
        p[i] = __builtin_elementwise_fma(p[i], p[8 + i], p[8 + ((i + 1) & 7)]);
    }
  }
  const long long t1 = __builtin_readcyclecounter();
  const long long r1 = __builtin_amdgcn_s_memrealtime();
  float r = 0;
#pragma unroll
  for (int i = 0; i < 16; ++i) r += p[i].x + p[i].y;
#pragma unroll
  for (int i = 0; i < 32; ++i) r += a[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = r;
  if (blockIdx.x == 0 && threadIdx.x == 0) { clocks[0] = t1 - t0; clocks[1] = r1 - r0; }
}

template <int MODE>
void run(const char* name, int waves_per_simd, double ops_per_iter, float* dout, long long* dclk) {
  const int blocks = 256 * waves_per_simd;
  const int iters = 20000;
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  valu<MODE><<<blocks, 256>>>(dout, dclk, 100, 1.0f);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  valu<MODE><<<blocks, 256>>>(dout, dclk, iters, 1.0f);
  CK(hipEventRecord(e1));
  CK(hipDeviceSynchronize());
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  long long clk[2]; CK(hipMemcpy(clk, dclk, sizeof clk, hipMemcpyDeviceToHost));
  const double tops = (double)blocks * 256 * iters * ops_per_iter / (ms * 1e-3) / 1e12;
  printf("%-34s waves/SIMD=%d %8.3f ms %7.2f T lane-ops/s   memtime/realtime %.2f (x100 MHz)\n",
         name, waves_per_simd, ms, tops, (double)clk[0] / (double)clk[1]);
}

int main() {
  float* dout; CK(hipMalloc(&dout, 256 * 8 * 256 * sizeof(float) * 4));
  long long* dclk; CK(hipMalloc(&dclk, 16));
  for (int w : {2, 3, 4, 8}) {
    run<0>("v_add_f32 vgpr,vgpr", w, 16, dout, dclk);
    run<1>("v_pk_add_f32 vgpr,vgpr", w, 16, dout, dclk);
    run<2>("v_pk_add_f32 vgpr,sgpr", w, 16, dout, dclk);
    run<5>("v_pk_fma_f32 3 vgpr pairs", w, 16, dout, dclk);
    run<3>("packed jacobi row (8 cells)", w, 120, dout, dclk);
    run<4>("scalar jacobi rows (8 cells)", w, 120, dout, dclk);
  }
  return 0;
}
