// Copy-kernel variants: what HBM rate can a streaming kernel reach on this box?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("ERR %s line %d: %s\n", #x, __LINE__, hipGetErrorString(e)); exit(1);} } while (0)
typedef float float4v __attribute__((ext_vector_type(4)));

template <int NT_LD, int NT_ST, int UNROLL>
__global__ void __launch_bounds__(256) copy_flat(const float4v* __restrict__ in, float4v* __restrict__ out, size_t n) {
  size_t i = (blockIdx.x * (size_t)blockDim.x + threadIdx.x);
  size_t stride = (size_t)gridDim.x * blockDim.x;
  for (; i + (UNROLL - 1) * stride < n; i += UNROLL * stride) {
    float4v v[UNROLL];
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) v[u] = NT_LD ? __builtin_nontemporal_load(&in[i + u * stride]) : in[i + u * stride];
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) { if (NT_ST) __builtin_nontemporal_store(v[u], &out[i + u * stride]); else out[i + u * stride] = v[u]; }
  }
}

// each wave walks down rows of a 2-D array: 1 KiB per row per wave, like the stencil kernels
template <int NT_LD, int NT_ST, int PF>
__global__ void __launch_bounds__(256) copy_rows(const float* __restrict__ in, float* __restrict__ out, long W, long H, long chunk) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long x = ((long)blockIdx.x * 4 + wave) * 256 + lane * 4;
  if (x >= W) return;
  const long y0 = (long)blockIdx.y * chunk, y1 = y0 + chunk < H ? y0 + chunk : H;
  float4v ring[PF];
#pragma unroll
  for (int p = 0; p < PF; ++p) { long y = y0 + p < H ? y0 + p : H - 1; const float4v* q = (const float4v*)(in + y * W + x); ring[p] = NT_LD ? __builtin_nontemporal_load(q) : *q; }
  for (long y = y0; y < y1; y += PF) {
#pragma unroll
    for (int p = 0; p < PF; ++p) {
      float4v v = ring[p];
      long yn = y + p + PF; if (yn > H - 1) yn = H - 1;
      const float4v* q = (const float4v*)(in + yn * W + x);
      ring[p] = NT_LD ? __builtin_nontemporal_load(q) : *q;
      if (y + p < y1) { float4v* o = (float4v*)(out + (y + p) * W + x); if (NT_ST) __builtin_nontemporal_store(v, o); else *o = v; }
    }
  }
}

// the fused kernels' geometry: overlapped strips, each wave loads 256 floats from
// xs - halo_lo and stores the w_out floats that start at xs; WPB waves per block
template <int PF, int WPB>
__global__ void __launch_bounds__(64 * WPB) copy_strips(const float* __restrict__ in, float* __restrict__ out, long W, long H, long chunk, int w_out, int halo_lo) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long xs = ((long)blockIdx.x * WPB + wave) * w_out + halo_lo;
  if (xs >= W - 256) return;
  const long x = xs - halo_lo + lane * 4;
  const bool st = x >= xs && x + 4 <= xs + w_out;
  const long y0 = (long)blockIdx.y * chunk, y1 = y0 + chunk < H ? y0 + chunk : H;
  float4v ring[PF];
#pragma unroll
  for (int p = 0; p < PF; ++p) { long y = y0 + p < H ? y0 + p : H - 1; ring[p] = *(const float4v*)(in + y * W + x); }
  for (long y = y0; y < y1; y += PF) {
#pragma unroll
    for (int p = 0; p < PF; ++p) {
      float4v v = ring[p];
      long yn = y + p + PF; if (yn > H - 1) yn = H - 1;
      ring[p] = *(const float4v*)(in + yn * W + x);
      if (st && y + p < y1) *(float4v*)(out + (y + p) * W + x) = v;
    }
  }
}

// rows through LDS-direct loads (global_load_lds_dwordx4: no registers), PF rows
// in flight per wavefront, one wavefront per 256 columns, 4 wavefronts per block
template <int PF>
__global__ void __launch_bounds__(256) copy_rows_lds(const float* __restrict__ in, float* __restrict__ out, long W, long H, long chunk) {
  __attribute__((shared)) float ring[4][PF + 1][256];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long x = ((long)blockIdx.x * 4 + wave) * 256 + lane * 4;
  if (x >= W) return;
  const long y0 = (long)blockIdx.y * chunk, y1 = y0 + chunk < H ? y0 + chunk : H;
  auto issue = [&](long y, int slot) {
    if (y > H - 1) y = H - 1;
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(in + y * W + x),
                                     (__attribute__((address_space(3))) void*)&ring[wave][slot][0], 16, 0, 0);
  };
#pragma unroll
  for (int p = 0; p < PF; ++p) issue(y0 + p, p);
  for (long y = y0; y < y1; y += PF + 1) {
#pragma unroll
    for (int p = 0; p < PF + 1; ++p) {
      issue(y + p + PF, (p + PF) % (PF + 1));
      // row y+p was issued PF loads ago: at most PF may still be in flight
      __builtin_amdgcn_s_waitcnt((PF & 15) | (7 << 4) | (15 << 8) | ((PF >> 4) << 14));
      const float4v v = *(const float4v*)&ring[wave][p][lane * 4];
      if (y + p < y1) *(float4v*)(out + (y + p) * W + x) = v;
    }
  }
  __builtin_amdgcn_s_waitcnt(0);
}

template <typename F> void timeit(const char* name, size_t bytes, F f) {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  f(); CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0)); for (int r = 0; r < 10; ++r) f(); CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  printf("%-44s %.3f ms  %.2f TB/s (read+write)\n", name, ms / 10, 2.0 * bytes * 10 / (ms * 1e-3) / 1e12);
}

int main() {
  const long W = 16384, H = 16384; const size_t bytes = (size_t)W * H * 4, n = bytes / 16;
  float *a, *b; CK(hipMalloc(&a, bytes)); CK(hipMalloc(&b, bytes)); CK(hipMemset(a, 1, bytes)); CK(hipMemset(b, 0, bytes));
  const float4v* a4 = (const float4v*)a; float4v* b4 = (float4v*)b;
  for (int blocks : {1024, 2048, 4096, 8192}) {
    char nm[96];
    snprintf(nm, sizeof nm, "flat plain u1 blocks=%d", blocks); timeit(nm, bytes, [&] { copy_flat<0,0,1><<<blocks,256>>>(a4, b4, n); });
    snprintf(nm, sizeof nm, "flat plain u4 blocks=%d", blocks); timeit(nm, bytes, [&] { copy_flat<0,0,4><<<blocks,256>>>(a4, b4, n); });
    snprintf(nm, sizeof nm, "flat nt-ld nt-st u4 blocks=%d", blocks); timeit(nm, bytes, [&] { copy_flat<1,1,4><<<blocks,256>>>(a4, b4, n); });
    snprintf(nm, sizeof nm, "flat nt-st u4 blocks=%d", blocks); timeit(nm, bytes, [&] { copy_flat<0,1,4><<<blocks,256>>>(a4, b4, n); });
  }
  for (long chunk : {128L, 256L, 512L, 1024L}) {
    dim3 grid((unsigned)(W / 1024), (unsigned)((H + chunk - 1) / chunk));
    char nm[96];
    snprintf(nm, sizeof nm, "rows plain pf3 chunk=%ld", chunk); timeit(nm, bytes, [&] { copy_rows<0,0,3><<<grid,256>>>(a, b, W, H, chunk); });
    snprintf(nm, sizeof nm, "rows plain pf6 chunk=%ld", chunk); timeit(nm, bytes, [&] { copy_rows<0,0,6><<<grid,256>>>(a, b, W, H, chunk); });
    snprintf(nm, sizeof nm, "rows nt-ld nt-st pf6 chunk=%ld", chunk); timeit(nm, bytes, [&] { copy_rows<1,1,6><<<grid,256>>>(a, b, W, H, chunk); });
    snprintf(nm, sizeof nm, "rows nt-st pf6 chunk=%ld", chunk); timeit(nm, bytes, [&] { copy_rows<0,1,6><<<grid,256>>>(a, b, W, H, chunk); });
  }
  for (long chunk : {256L, 1024L}) {
    dim3 grid((unsigned)(W / 1024), (unsigned)((H + chunk - 1) / chunk));
    char nm[96];
    snprintf(nm, sizeof nm, "rows lds-direct pf3 chunk=%ld", chunk); timeit(nm, bytes, [&] { copy_rows_lds<3><<<grid,256>>>(a, b, W, H, chunk); });
    snprintf(nm, sizeof nm, "rows lds-direct pf7 chunk=%ld", chunk); timeit(nm, bytes, [&] { copy_rows_lds<7><<<grid,256>>>(a, b, W, H, chunk); });
    snprintf(nm, sizeof nm, "rows lds-direct pf15 chunk=%ld", chunk); timeit(nm, bytes, [&] { copy_rows_lds<15><<<grid,256>>>(a, b, W, H, chunk); });
  }
  struct G { int w_out, halo_lo; };
  for (G g : {G{256, 0}, G{248, 4}, G{232, 12}, G{224, 16}, G{224, 0}, G{192, 32}}) {
    for (long chunk : {256L, 576L}) {
      const long strips = (W - 256) / g.w_out;
      char nm[128];
      dim3 g1((unsigned)strips, (unsigned)((H + chunk - 1) / chunk));
      snprintf(nm, sizeof nm, "strips w_out=%d halo_lo=%d chunk=%ld 1 wave/blk", g.w_out, g.halo_lo, chunk);
      timeit(nm, (size_t)strips * g.w_out * H * 4, [&] { copy_strips<3, 1><<<g1, 64>>>(a, b, W, H, chunk, g.w_out, g.halo_lo); });
      dim3 g4((unsigned)((strips + 3) / 4), (unsigned)((H + chunk - 1) / chunk));
      snprintf(nm, sizeof nm, "strips w_out=%d halo_lo=%d chunk=%ld 4 waves/blk", g.w_out, g.halo_lo, chunk);
      timeit(nm, (size_t)strips * g.w_out * H * 4, [&] { copy_strips<3, 4><<<g4, 256>>>(a, b, W, H, chunk, g.w_out, g.halo_lo); });
    }
  }
  return 0;
}
