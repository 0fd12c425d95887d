// What a lane-crossing operand costs the VALU on gfx950: v_add_f32 with a DPP source in
// every control the fused stencil kernels could use (row-local shifts, whole-wave shifts,
// broadcasts), the two-instruction forms (v_mov_b32_dpp + add, ds_bpermute_b32 + add,
// v_permlane32_swap), and the 3-D block kernel's instruction mix (per two cells of a
// 7-point level: 10 plain adds, 2 adds with a lane-crossing operand, 2 multiplies) with
// each of them.  Reported: wave-instructions per SIMD cycle at the shader clock the
// kernel itself sees, and T lane-ops/s.  (Run on the GPU box.)
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off dppbench.hip -o dppbench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("ERR %s line %d: %s\n", #x, __LINE__, hipGetErrorString(e)); exit(1);} } while (0)

#define DPP_ADD(CTRL) \
  asm volatile("v_add_f32_dpp %0, %1, %0 " CTRL : "+v"(a[i]) : "v"(b[i]))
#define DPP_MOV(CTRL) \
  asm volatile("v_mov_b32_dpp %0, %1 " CTRL : "=v"(t) : "v"(b[i]))

enum { PLAIN, QUAD, ROW_SHR, ROW_SHL, ROW_ROR, WAVE_SHR, WAVE_SHL, WAVE_ROR, ROW_BCAST15,
       ROW_MIRROR, MOV_WAVE_SHR, MOV_ROW_SHR, BPERMUTE, PERMLANE32, MIX_NONE, MIX_WAVE,
       MIX_ROW, MIX_BPERMUTE, MIX_MOV_ROW2, N_MODES };
static const char* NAMES[N_MODES] = {
    "v_add_f32 (no dpp)", "v_add_f32_dpp quad_perm", "v_add_f32_dpp row_shr:1",
    "v_add_f32_dpp row_shl:1", "v_add_f32_dpp row_ror:1", "v_add_f32_dpp wave_shr:1",
    "v_add_f32_dpp wave_shl:1", "v_add_f32_dpp wave_ror:1", "v_add_f32_dpp row_bcast:15",
    "v_add_f32_dpp row_mirror", "v_mov_b32_dpp wave_shr:1 + v_add", "v_mov_b32_dpp row_shr:1 + v_add",
    "ds_bpermute_b32 + v_add", "v_permlane32_swap + v_add",
    "7-point mix, no lane crossing", "7-point mix, wave_shr/shl fused", "7-point mix, row_shr/shl fused",
    "7-point mix, ds_bpermute", "7-point mix, row_bcast + row_shr movs"};
// VALU instructions per loop trip (for the rate) and lane-crossing ones among them
static const int INSTR[N_MODES] = {8, 8, 8, 8, 8, 8, 8, 8, 8, 8, 16, 16, 8, 16, 56, 56, 56, 56, 72};

template <int MODE>
__global__ void __launch_bounds__(256) bench(float* out, long long* clocks, int iters) {
  float a[8], b[8];
  const int lane = threadIdx.x & 63;
  const int below = ((lane - 1) & 63) << 2, above = ((lane + 1) & 63) << 2;
#pragma unroll
  for (int i = 0; i < 8; ++i) { a[i] = threadIdx.x * 0.001f + i; b[i] = 1.0f / (threadIdx.x + i + 1); }
  const long long t0 = __builtin_readcyclecounter();
  const long long r0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < iters; ++it) {
    if (MODE < MIX_NONE) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        float t;
        if (MODE == PLAIN) asm volatile("v_add_f32 %0, %1, %0" : "+v"(a[i]) : "v"(b[i]));
        if (MODE == QUAD) DPP_ADD("quad_perm:[1,2,3,0] row_mask:0xf bank_mask:0xf");
        if (MODE == ROW_SHR) DPP_ADD("row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1");
        if (MODE == ROW_SHL) DPP_ADD("row_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1");
        if (MODE == ROW_ROR) DPP_ADD("row_ror:1 row_mask:0xf bank_mask:0xf");
        if (MODE == WAVE_SHR) DPP_ADD("wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1");
        if (MODE == WAVE_SHL) DPP_ADD("wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1");
        if (MODE == WAVE_ROR) DPP_ADD("wave_ror:1 row_mask:0xf bank_mask:0xf");
        if (MODE == ROW_BCAST15) DPP_ADD("row_bcast:15 row_mask:0xa bank_mask:0xf");
        if (MODE == ROW_MIRROR) DPP_ADD("row_mirror row_mask:0xf bank_mask:0xf");
        if (MODE == MOV_WAVE_SHR) {
          DPP_MOV("wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1");
          asm volatile("s_nop 1\n\tv_add_f32 %0, %1, %0" : "+v"(a[i]) : "v"(t));
        }
        if (MODE == MOV_ROW_SHR) {
          DPP_MOV("row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1");
          asm volatile("s_nop 1\n\tv_add_f32 %0, %1, %0" : "+v"(a[i]) : "v"(t));
        }
        if (MODE == BPERMUTE) {
          t = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(below, __builtin_bit_cast(int, b[i])));
          a[i] += t;
        }
        if (MODE == PERMLANE32) {
          const auto r = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, b[i]),
                                                          __builtin_bit_cast(unsigned, a[(i + 1) & 7]), false, false);
          a[i] += __builtin_bit_cast(float, (unsigned)r[0]);
        }
      }
    } else {
      // four pairs of cells (C = 2 columns per lane): cell 0 takes its left neighbour
      // from the lane below, cell 1 its right neighbour from the lane above; the other
      // five operands of each are the lane's own registers.  Chains as the C++ text has
      // them: ((((((zm + ym) + xm) + c) + xp) + yp) + zp) * k
#pragma unroll
      for (int i = 0; i < 8; i += 2) {
        float s0 = a[i] + b[i], s1 = a[i + 1] + b[i + 1];
        float l, r;
        if (MODE == MIX_NONE) { l = b[(i + 3) & 7]; r = b[(i + 4) & 7]; }
        if (MODE == MIX_WAVE) {
          l = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, a[i + 1]), 0x138, 0xf, 0xf, true));
          r = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, a[i]), 0x130, 0xf, 0xf, true));
        }
        if (MODE == MIX_ROW) {
          l = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, a[i + 1]), 0x111, 0xf, 0xf, true));
          r = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, a[i]), 0x101, 0xf, 0xf, true));
        }
        if (MODE == MIX_BPERMUTE) {
          l = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(below, __builtin_bit_cast(int, a[i + 1])));
          r = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(above, __builtin_bit_cast(int, a[i])));
        }
        if (MODE == MIX_MOV_ROW2) {
          // whole-wave shift from row-local pieces: the row's first lane takes the
          // previous row's last (row_bcast:15), the others their neighbour (row_shr:1,
          // lanes without a source keep `old`); upwards: row_shl:1 after a wave_shl of
          // ... (priced here as two moves each way, whatever the exact pair)
          int t = __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, a[i + 1]), 0x142, 0xe, 0xf, false);
          l = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(t, __builtin_bit_cast(int, a[i + 1]), 0x111, 0xf, 0xf, false));
          int u = __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, a[i]), 0x142, 0xe, 0xf, false);
          r = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(u, __builtin_bit_cast(int, a[i]), 0x101, 0xf, 0xf, false));
        }
        s0 = ((((s0 + l) + a[i]) + a[i + 1]) + b[(i + 2) & 7]) + b[(i + 5) & 7];
        s1 = ((((s1 + a[i]) + a[i + 1]) + r) + b[(i + 3) & 7]) + b[(i + 6) & 7];
        a[i] = s0 * 0.1428f;
        a[i + 1] = s1 * 0.1428f;
      }
    }
  }
  const long long t1 = __builtin_readcyclecounter();
  const long long r1 = __builtin_amdgcn_s_memrealtime();
  float r = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) r += a[i] + b[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = r;
  if (blockIdx.x == 0 && threadIdx.x == 0) { clocks[0] = t1 - t0; clocks[1] = r1 - r0; }
}

template <int MODE>
void run(int waves_per_simd, float* dout, long long* dclk) {
  const int blocks = 256 * waves_per_simd, iters = 20000;
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  bench<MODE><<<blocks, 256>>>(dout, dclk, 200);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  bench<MODE><<<blocks, 256>>>(dout, dclk, iters);
  CK(hipEventRecord(e1));
  CK(hipDeviceSynchronize());
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  long long clk[2]; CK(hipMemcpy(clk, dclk, sizeof clk, hipMemcpyDeviceToHost));
  const double ghz = (double)clk[0] / (double)clk[1] * 0.1;      // s_memrealtime: 100 MHz
  const double per_simd = (double)waves_per_simd * iters * INSTR[MODE];   // wave-instructions per SIMD
  const double cycles = ms * 1e-3 * ghz * 1e9;
  printf("%-42s waves/SIMD=%d %8.3f ms  %5.2f cycles per VALU instruction per SIMD  (%.2f GHz)\n",
         NAMES[MODE], waves_per_simd, ms, cycles / per_simd, ghz);
  fflush(stdout);
}

template <int MODE>
void sweep(float* dout, long long* dclk) {
  for (int w : {1, 2, 4}) run<MODE>(w, dout, dclk);
  if constexpr (MODE + 1 < N_MODES) sweep<MODE + 1>(dout, dclk);
}

int main() {
  float* dout; CK(hipMalloc(&dout, 256 * 8 * 256 * sizeof(float)));
  long long* dclk; CK(hipMalloc(&dclk, 16));
  sweep<0>(dout, dclk);
  return 0;
}
