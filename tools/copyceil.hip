// What "all of HBM" is for a kernel on this chip: plain float4 copies of 1 GiB into
// another 1 GiB (MI355X_MICROARCH.md quotes 6.29 TB/s for a float4 copy), in the shapes
// that matter for the depth-1 stencil kernels: how many bytes a CU keeps in flight, how a
// wavefront walks the array (grid-stride, contiguous chunks per workgroup, 1 KiB-wide
// strips walked row by row as the 2-D kernels do), temporal or non-temporal accesses.
// tools/k1bench.hip stops at 5.3-5.45 TB/s; this probe looks for the rest.
// Build: hipcc --offload-arch=gfx950 -O3 copyceil.hip -o copyceil   (run on the GPU box)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("ERR %s line %d: %s\n", #x, __LINE__, hipGetErrorString(e)); exit(1);} } while (0)
typedef float f4 __attribute__((ext_vector_type(4)));

template <int NT> __device__ inline f4 ld(const f4* p) {
  return NT & 1 ? __builtin_nontemporal_load(p) : *p;
}
template <int NT> __device__ inline void st(f4* p, f4 v) {
  if (NT & 2) __builtin_nontemporal_store(v, p); else *p = v;
}

// grid-stride: thread t copies float4 t, t + T, t + 2 T, ...; U loads in flight per lane
template <int U, int NT>
__global__ void __launch_bounds__(256) copy_gs(const f4* __restrict__ in, f4* __restrict__ out, size_t n) {
  const size_t T = (size_t)gridDim.x * blockDim.x;
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  for (; i + (U - 1) * T < n; i += U * T) {
    f4 r[U];
#pragma unroll
    for (int u = 0; u < U; ++u) r[u] = ld<NT>(in + i + u * T);
#pragma unroll
    for (int u = 0; u < U; ++u) st<NT>(out + i + u * T, r[u]);
  }
  for (; i < n; i += T) st<NT>(out + i, ld<NT>(in + i));
}

// contiguous chunk per workgroup: block b copies float4s [b * per, (b + 1) * per), its
// 256 threads side by side (4 KiB per pass), U passes in flight
template <int U, int NT>
__global__ void __launch_bounds__(256) copy_chunk(const f4* __restrict__ in, f4* __restrict__ out, size_t n,
                                                   size_t per) {
  const size_t lo = (size_t)blockIdx.x * per, hi = lo + per < n ? lo + per : n;
  size_t i = lo + threadIdx.x;
  for (; i + (U - 1) * 256 < hi; i += U * 256) {
    f4 r[U];
#pragma unroll
    for (int u = 0; u < U; ++u) r[u] = ld<NT>(in + i + u * 256);
#pragma unroll
    for (int u = 0; u < U; ++u) st<NT>(out + i + u * 256, r[u]);
  }
  for (; i < hi; i += 256) st<NT>(out + i, ld<NT>(in + i));
}

// strips: a wavefront owns 256 columns (1 KiB) of a W-column array and walks `chunk` rows,
// PF rows in flight (software pipeline in registers) - the 2-D kernels' structure
template <int PF, int WPB, int NT>
__global__ void __launch_bounds__(64 * WPB) copy_rows(const float* __restrict__ in, float* __restrict__ out,
                                                      long W, long H, long chunk) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long x = ((long)blockIdx.x * WPB + wave) * 256 + lane * 4;
  if (x + 4 > W) return;
  const long y0 = (long)blockIdx.y * chunk, y1 = y0 + chunk < H ? y0 + chunk : H;
  f4 r[PF];
#pragma unroll
  for (int p = 0; p < PF; ++p) r[p] = ld<NT>((const f4*)(in + (y0 + p < H ? y0 + p : H - 1) * W + x));
  for (long y = y0; y < y1; y += PF) {
#pragma unroll
    for (int p = 0; p < PF; ++p) {
      const f4 v = r[p];
      const long yn = y + p + PF;
      r[p] = ld<NT>((const f4*)(in + (yn < H ? yn : H - 1) * W + x));
      if (y + p < y1) st<NT>((f4*)(out + (y + p) * W + x), v);
    }
  }
}

// rows through LDS-direct loads (no registers for data in flight): a wavefront keeps N
// rows of its strip in flight in an N-slot LDS ring and copies them out with ds_read_b128
template <int N, int WPB, int NT>
__global__ void __launch_bounds__(64 * WPB) copy_rows_lds(const float* __restrict__ in, float* __restrict__ out,
                                                          long W, long H, long chunk) {
  __shared__ f4 ring[WPB][N][64];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const long x = ((long)blockIdx.x * WPB + wave) * 256 + lane * 4;
  if (x + 4 > W) return;
  const long y0 = (long)blockIdx.y * chunk, y1 = y0 + chunk < H ? y0 + chunk : H;
  auto issue = [&](int slot, long y) {
    if (y > H - 1) y = H - 1;
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(in + y * W + x),
                                     (__attribute__((address_space(3))) void*)&ring[wave][slot][0], 16, 0,
                                     NT & 1 ? 2 : 0);
  };
#pragma unroll
  for (int p = 0; p < N; ++p) issue(p, y0 + p);
  for (long y = y0; y < y1; y += N) {
#pragma unroll
    for (int p = 0; p < N; ++p) {
      // row y + p was issued N loads (and N stores) ago
      __builtin_amdgcn_s_waitcnt(((2 * N - 2) & 15) | (7 << 4) | (15 << 8) | (((2 * N - 2) >> 4) << 14));
      f4 v;
      asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v)
                   : "v"((unsigned)(unsigned long long)&ring[wave][p][lane]) : "memory");
      issue(p, y + p + N);
      f4* o = (f4*)(out + (y + p < y1 ? y + p : y1 - 1) * W + x);
      st<NT>(o, v);
    }
  }
  __builtin_amdgcn_s_waitcnt(0);
}

// tiles: a wavefront loads R + 2 rows of its 256 columns (all issued at once: nothing is
// carried from tile to tile), stores R, and ends; workgroups are dispatched x fastest, so
// the whole chip sweeps down the array together as the one-float4-per-thread copy does -
// the shape a depth-1 stencil kernel WITHOUT a row pipeline would have (halo rows are
// re-read by the tile below, on the same XCD: workgroup id mod 8 = strip mod 8)
template <int R, int WPB, int NT>
__global__ void __launch_bounds__(64 * WPB) copy_tiles(const float* __restrict__ in, float* __restrict__ out,
                                                       long W, long H) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long x = ((long)blockIdx.x * WPB + wave) * 256 + lane * 4;
  const long y0 = (long)blockIdx.y * R;
  f4 r[R + 2];
#pragma unroll
  for (int p = 0; p < R + 2; ++p) {
    long y = y0 - 1 + p; if (y < 0) y = 0; if (y > H - 1) y = H - 1;
    r[p] = ld<NT>((const f4*)(in + y * W + x));
  }
#pragma unroll
  for (int p = 0; p < R; ++p) {
    const f4 v = (r[p] + r[p + 1]) + r[p + 2];        // something of all three rows
    if (y0 + p < H) st<NT>((f4*)(out + (y0 + p) * W + x), v);
  }
}

template <int NT>
__global__ void __launch_bounds__(256) fill_gs(f4* __restrict__ out, size_t n) {
  const size_t T = (size_t)gridDim.x * blockDim.x;
  const f4 v = {1.f, 2.f, 3.f, 4.f};
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += T) st<NT>(out + i, v);
}

template <int U>
__global__ void __launch_bounds__(256) read_gs(const f4* __restrict__ in, float* __restrict__ sink, size_t n) {
  const size_t T = (size_t)gridDim.x * blockDim.x;
  f4 acc = {0, 0, 0, 0};
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  for (; i + (U - 1) * T < n; i += U * T) {
    f4 r[U];
#pragma unroll
    for (int u = 0; u < U; ++u) r[u] = in[i + u * T];
#pragma unroll
    for (int u = 0; u < U; ++u) acc += r[u];
  }
  if (acc[0] + acc[1] + acc[2] + acc[3] == 12345.678f) *sink = acc[0];
}

static hipEvent_t e0, e1;
template <class F>
void timeit(const char* name, double bytes, F launch) {
  for (int i = 0; i < 3; ++i) launch();
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  const int reps = 20;
  for (int i = 0; i < reps; ++i) launch();
  CK(hipEventRecord(e1));
  CK(hipDeviceSynchronize());
  CK(hipGetLastError());
  float ms;
  CK(hipEventElapsedTime(&ms, e0, e1));
  printf("%-64s %8.1f us  %.2f TB/s\n", name, ms * 1000 / reps, bytes * reps / (ms * 1e-3) / 1e12);
  fflush(stdout);
}

int main(int argc, char** argv) {
  const long W = 16384, H = 16384;
  const size_t bytes = (size_t)W * H * 4, n = bytes / 16;
  float *a, *b;
  CK(hipMalloc(&a, bytes)); CK(hipMalloc(&b, bytes));
  CK(hipMemset(a, 0x3c, bytes)); CK(hipMemset(b, 0, bytes));
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const f4* in = (const f4*)a; f4* out = (f4*)b;
  char name[128];
  printf("== 1 GiB -> 1 GiB, float4 lanes; TB/s counts read + written bytes\n");
  for (int blocks : {1024, 2048, 4096, 8192, 16384}) {
    snprintf(name, sizeof name, "grid-stride, %d blocks of 256, 1 in flight", blocks);
    timeit(name, 2.0 * bytes, [&] { copy_gs<1, 0><<<blocks, 256>>>(in, out, n); });
    snprintf(name, sizeof name, "grid-stride, %d blocks of 256, 4 in flight", blocks);
    timeit(name, 2.0 * bytes, [&] { copy_gs<4, 0><<<blocks, 256>>>(in, out, n); });
    snprintf(name, sizeof name, "grid-stride, %d blocks of 256, 8 in flight", blocks);
    timeit(name, 2.0 * bytes, [&] { copy_gs<8, 0><<<blocks, 256>>>(in, out, n); });
  }
  timeit("grid-stride, one float4 per thread (262144 blocks)", 2.0 * bytes,
         [&] { copy_gs<1, 0><<<(unsigned)(n / 256), 256>>>(in, out, n); });
  timeit("grid-stride, 4096 blocks, 4 in flight, nt loads", 2.0 * bytes,
         [&] { copy_gs<4, 1><<<4096, 256>>>(in, out, n); });
  timeit("grid-stride, 4096 blocks, 4 in flight, nt stores", 2.0 * bytes,
         [&] { copy_gs<4, 2><<<4096, 256>>>(in, out, n); });
  timeit("grid-stride, 4096 blocks, 4 in flight, nt both", 2.0 * bytes,
         [&] { copy_gs<4, 3><<<4096, 256>>>(in, out, n); });
  timeit("grid-stride, 4096 blocks, 8 in flight, nt both", 2.0 * bytes,
         [&] { copy_gs<8, 3><<<4096, 256>>>(in, out, n); });
  for (size_t kib : {64, 256, 1024}) {
    const size_t per = kib * 1024 / 16;
    const unsigned blocks = (unsigned)((n + per - 1) / per);
    snprintf(name, sizeof name, "chunk of %zu KiB per workgroup (%u blocks), 4 in flight", kib, blocks);
    timeit(name, 2.0 * bytes, [&] { copy_chunk<4, 0><<<blocks, 256>>>(in, out, n, per); });
    snprintf(name, sizeof name, "chunk of %zu KiB per workgroup (%u blocks), 4 in flight, nt stores", kib, blocks);
    timeit(name, 2.0 * bytes, [&] { copy_chunk<4, 2><<<blocks, 256>>>(in, out, n, per); });
  }
  printf("== strips of 256 columns walked row by row (16384 x 16384)\n");
  for (long chunk : {128L, 256L, 512L, 1024L}) {
    dim3 grid((unsigned)(W / 256 / 4), (unsigned)((H + chunk - 1) / chunk));
    snprintf(name, sizeof name, "rows: 4 waves/blk, 3 rows in flight, chunk %ld", chunk);
    timeit(name, 2.0 * bytes, [&] { copy_rows<3, 4, 0><<<grid, 256>>>(a, b, W, H, chunk); });
    snprintf(name, sizeof name, "rows: 4 waves/blk, 6 rows in flight, chunk %ld", chunk);
    timeit(name, 2.0 * bytes, [&] { copy_rows<6, 4, 0><<<grid, 256>>>(a, b, W, H, chunk); });
    snprintf(name, sizeof name, "rows: 4 waves/blk, 6 rows in flight, nt stores, chunk %ld", chunk);
    timeit(name, 2.0 * bytes, [&] { copy_rows<6, 4, 2><<<grid, 256>>>(a, b, W, H, chunk); });
    snprintf(name, sizeof name, "rows: 4 waves/blk, 12 rows in flight, nt stores, chunk %ld", chunk);
    timeit(name, 2.0 * bytes, [&] { copy_rows<12, 4, 2><<<grid, 256>>>(a, b, W, H, chunk); });
    snprintf(name, sizeof name, "rows via LDS ring of 8, 4 waves/blk, nt stores, chunk %ld", chunk);
    timeit(name, 2.0 * bytes, [&] { copy_rows_lds<8, 4, 2><<<grid, 256>>>(a, b, W, H, chunk); });
    snprintf(name, sizeof name, "rows via LDS ring of 16, 4 waves/blk, nt stores, chunk %ld", chunk);
    timeit(name, 2.0 * bytes, [&] { copy_rows_lds<16, 4, 2><<<grid, 256>>>(a, b, W, H, chunk); });
  }
  printf("== tiles of R rows, R + 2 loaded, no row pipeline (TB/s of the 2 GiB of unique bytes)\n");
  {
    dim3 g8((unsigned)(W / 256 / 4), (unsigned)(H / 8)), g16((unsigned)(W / 256 / 4), (unsigned)(H / 16)),
        g32((unsigned)(W / 256 / 4), (unsigned)(H / 32));
    timeit("tiles:  8 rows, 4 waves/blk", 2.0 * bytes, [&] { copy_tiles<8, 4, 0><<<g8, 256>>>(a, b, W, H); });
    timeit("tiles:  8 rows, 4 waves/blk, nt stores", 2.0 * bytes, [&] { copy_tiles<8, 4, 2><<<g8, 256>>>(a, b, W, H); });
    timeit("tiles: 16 rows, 4 waves/blk", 2.0 * bytes, [&] { copy_tiles<16, 4, 0><<<g16, 256>>>(a, b, W, H); });
    timeit("tiles: 16 rows, 4 waves/blk, nt stores", 2.0 * bytes, [&] { copy_tiles<16, 4, 2><<<g16, 256>>>(a, b, W, H); });
    timeit("tiles: 32 rows, 4 waves/blk", 2.0 * bytes, [&] { copy_tiles<32, 4, 0><<<g32, 256>>>(a, b, W, H); });
    timeit("tiles: 32 rows, 4 waves/blk, nt stores", 2.0 * bytes, [&] { copy_tiles<32, 4, 2><<<g32, 256>>>(a, b, W, H); });
    dim3 h16((unsigned)(W / 256), (unsigned)(H / 16));
    timeit("tiles: 16 rows, 1 wave/blk", 2.0 * bytes, [&] { copy_tiles<16, 1, 0><<<h16, 64>>>(a, b, W, H); });
    timeit("tiles: 16 rows, 1 wave/blk, nt stores", 2.0 * bytes, [&] { copy_tiles<16, 1, 2><<<h16, 64>>>(a, b, W, H); });
    dim3 q16((unsigned)(W / 256 / 16), (unsigned)(H / 16));
    timeit("tiles: 16 rows, 16 waves/blk (a whole row of the array)", 2.0 * bytes, [&] { copy_tiles<16, 16, 0><<<q16, 1024>>>(a, b, W, H); });
    timeit("tiles: 16 rows, 16 waves/blk, nt stores", 2.0 * bytes, [&] { copy_tiles<16, 16, 2><<<q16, 1024>>>(a, b, W, H); });
  }
  printf("== one direction only\n");
  timeit("fill 1 GiB, 4096 blocks", 1.0 * bytes, [&] { fill_gs<0><<<4096, 256>>>(out, n); });
  timeit("fill 1 GiB, 4096 blocks, nt", 1.0 * bytes, [&] { fill_gs<2><<<4096, 256>>>(out, n); });
  timeit("read 1 GiB, 4096 blocks, 4 in flight", 1.0 * bytes, [&] { read_gs<4><<<4096, 256>>>(in, b, n); });
  timeit("read 1 GiB, 4096 blocks, 8 in flight", 1.0 * bytes, [&] { read_gs<8><<<4096, 256>>>(in, b, n); });
  return 0;
}
