// A hand-written depth-1 jacobi2d (5-point) in the row-copy structure that reaches
// 5.5-6 TB/s as a pure copy (tools/copybench.hip: copy_rows), to tell apart what holds
// the generated depth-1 kernels at ~4.5 TB/s of unique bytes: the arithmetic, the halo /
// store alignment of the strips, or the generator's loop structure.  (Run on the GPU box.)
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off k1bench.hip -o k1bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("ERR %s line %d: %s\n", #x, __LINE__, hipGetErrorString(e)); exit(1);} } while (0)
typedef float f4 __attribute__((ext_vector_type(4)));

__device__ inline float from_below(float v) { return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x138, 0xf, 0xf, true)); }
__device__ inline float from_above(float v) { return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x130, 0xf, 0xf, true)); }

// strips of 256 loaded columns at xs - HALO, W_OUT stored from xs; WPB wavefronts per
// block side by side; PF rows in flight beyond the three of the window; ARITH = 0: copy
template <int PF, int WPB, int NT, int ARITH>
__global__ void __launch_bounds__(64 * WPB) k1(const float* __restrict__ in, float* __restrict__ out,
                                               long W, long H, long chunk, int w_out, int halo) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long xs = ((long)blockIdx.x * WPB + wave) * w_out + halo;
  if (xs + 256 - halo > W) return;
  const long x = xs - halo + lane * 4;
  const bool st = x >= xs && x + 4 <= xs + w_out;
  const long y0 = 1 + (long)blockIdx.y * chunk, y1 = y0 + chunk < H - 1 ? y0 + chunk : H - 1;
  constexpr int N = PF + 3;
  f4 r[N];
  auto load = [&](long y) {
    if (y > H - 1) y = H - 1;
    const f4* q = (const f4*)(in + y * W + x);
    return NT ? __builtin_nontemporal_load(q) : *q;
  };
#pragma unroll
  for (int p = 0; p < N - 1; ++p) r[p] = load(y0 - 1 + p);
  for (long y = y0; y < y1; y += N) {
#pragma unroll
    for (int p = 0; p < N; ++p) {
      // rows y+p-1, y+p, y+p+1 sit in slots p, p+1, p+2 (mod N); fetch row y+p+N-2
      r[(p + N - 1) % N] = load(y + p + N - 2);
      const f4 up = r[p % N], mid = r[(p + 1) % N], down = r[(p + 2) % N];
      f4 v = mid;
      if (ARITH) {
        const float left = from_below(mid[3]), right = from_above(mid[0]);
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const float l = c == 0 ? left : mid[c - 1], rr = c == 3 ? right : mid[c + 1];
          v[c] = ((((down[c] + rr) + mid[c]) + up[c]) + l) * 0.2f;
        }
      }
      if (st && y + p < y1) {
        f4* o = (f4*)(out + (y + p) * W + x);
        if (NT) __builtin_nontemporal_store(v, o); else *o = v;
      }
    }
  }
}

template <int PF, int WPB, int NT, int ARITH>
void run(const char* name, const float* a, float* b, long W, long H, long chunk, int w_out, int halo) {
  const long strips = (W - 256 + w_out) / w_out - (halo ? 1 : 0);
  dim3 grid((unsigned)((strips + WPB - 1) / WPB), (unsigned)((H - 2 + chunk - 1) / chunk));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int i = 0; i < 3; ++i) k1<PF, WPB, NT, ARITH><<<grid, 64 * WPB>>>(a, b, W, H, chunk, w_out, halo);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  for (int i = 0; i < 10; ++i) k1<PF, WPB, NT, ARITH><<<grid, 64 * WPB>>>(a, b, W, H, chunk, w_out, halo);
  CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  const double unique = 2.0 * (double)strips * w_out * (H - 2) * 4;
  printf("%-58s chunk %5ld  %7.1f us  %.2f TB/s of stored+matching read bytes\n", name, chunk, ms * 100,
         unique * 10 / (ms * 1e-3) / 1e12);
}

int main() {
  const long W = 16384, H = 16384; const size_t bytes = (size_t)W * H * 4;
  float *a, *b; CK(hipMalloc(&a, bytes)); CK(hipMalloc(&b, bytes)); CK(hipMemset(b, 0, bytes));
  {   // random operands in [0, 1): constant data runs at other clocks and bus power
    float* host = (float*)malloc(bytes);
    unsigned long long state = 88172645463325252ull;
    for (size_t i = 0; i < bytes / 4; ++i) {
      state ^= state << 13; state ^= state >> 7; state ^= state << 17;
      host[i] = (float)(state >> 40) * (1.0f / 16777216.0f);
    }
    if (getenv("K1_CONSTANT")) for (size_t i = 0; i < bytes / 4; ++i) host[i] = 2.36943e-38f;
    CK(hipMemcpy(a, host, bytes, hipMemcpyHostToDevice));
    free(host);
  }
  for (long chunk : {180L, 512L}) {
    run<3, 4, 0, 0>("copy   pf3  256 of 256 (aligned, no halo)", a, b, W, H, chunk, 256, 0);
    run<3, 4, 0, 1>("jacobi pf3  256 of 256 (aligned, edges garbage)", a, b, W, H, chunk, 256, 0);
    run<6, 4, 1, 1>("jacobi pf6 nt 256 of 256", a, b, W, H, chunk, 256, 0);
    run<3, 4, 0, 1>("jacobi pf3  248 of 256, halo 4 (stores 16 B into lines)", a, b, W, H, chunk, 248, 4);
    run<3, 4, 0, 1>("jacobi pf3  192 of 256, halo 32 (whole lines)", a, b, W, H, chunk, 192, 32);
    run<3, 4, 0, 1>("jacobi pf3  224 of 256, halo 16 (64-byte seams)", a, b, W, H, chunk, 224, 16);
    run<6, 4, 1, 1>("jacobi pf6 nt 192 of 256, halo 32", a, b, W, H, chunk, 192, 32);
    run<6, 4, 1, 1>("jacobi pf6 nt 224 of 256, halo 16", a, b, W, H, chunk, 224, 16);
    run<3, 1, 0, 1>("jacobi pf3  192 of 256, halo 32, 1 wave/blk", a, b, W, H, chunk, 192, 32);
    run<3, 4, 0, 0>("copy   pf3  192 of 256, halo 32", a, b, W, H, chunk, 192, 32);
    run<3, 4, 0, 0>("copy   pf3  224 of 256, halo 16", a, b, W, H, chunk, 224, 16);
  }
  return 0;
}
