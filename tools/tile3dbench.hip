// Memory side of the 3-D depth-4 block kernels without their arithmetic (run on the
// GPU box): a workgroup of G wavefronts streams a (64*C) x (G*R) plane tile along z,
// keeping PF planes in flight in registers, and stores the interior (halo HALO on
// every side) HALO planes late - the access pattern of kernel_stream3d_blk.  Answers:
// what does the chip give this pattern as a function of planes in flight, lane width,
// tile shape, a per-plane barrier and workgroup placement?
// Build: hipcc --offload-arch=gfx950 -O3 tile3dbench.hip -o tile3dbench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("ERR %s line %d: %s\n", #x, __LINE__, hipGetErrorString(e)); exit(1);} } while (0)

template <int C> struct vecof;
template <> struct vecof<2> { typedef float type __attribute__((ext_vector_type(2))); };
template <> struct vecof<4> { typedef float type __attribute__((ext_vector_type(4))); };

struct Geo {
  int W, H, D;          // array
  int lo, hi;           // box [lo, hi) in every dimension
  int chunk;            // planes per z chunk
  int ntx, nty, ntz;    // tiles
  int ld_shift, st_shift;  // columns added to the load / store addresses (alignment probes)
  int order;            // 0: x fastest, 1: z-chunk fastest (neighbours in z adjacent ids),
                        // 2: ids dealt so that an XCD (id % 8) owns whole z chunks
};

template <int C, int R, int G, int PF, int HALO, int BARRIER, int NT, int PITCH = 0>
__global__ void __launch_bounds__(G * 64) tilecopy(const float* __restrict__ in,
                                                   float* __restrict__ out, Geo g) {
  typedef typename vecof<C>::type vec;
  // PITCH > 0: out tiles PITCH columns wide starting on multiples of 32 columns
  // (whole 128-byte lines per store except at the box edges)
  constexpr int TW = 64 * C, TR = G * R, WOUT = PITCH ? PITCH : TW - 2 * HALO, ROUT = TR - 2 * HALO;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  unsigned id = blockIdx.x, bx, by, bz;
  if (g.order == 0) { bx = id % g.ntx; by = (id / g.ntx) % g.nty; bz = id / (g.ntx * g.nty); }
  else if (g.order == 1) { bz = id % g.ntz; bx = (id / g.ntz) % g.ntx; by = id / (g.ntz * g.ntx); }
  else if (g.order == 3) {  // each XCD (id % 8) owns a RUN of consecutive tiles (x fastest)
    const unsigned total = g.ntx * g.nty * g.ntz, per = (total + 7) / 8;
    const unsigned T = (id & 7) * per + (id >> 3);
    if ((id >> 3) >= per || T >= total) return;
    bx = T % g.ntx; by = (T / g.ntx) % g.nty; bz = T / (g.ntx * g.nty);
  }
  else {  // XCD x owns z chunks x, x+8, ...; within an XCD x fastest
    const unsigned xcd = id & 7, k = id >> 3, per = g.ntx * g.nty;
    bz = xcd + 8 * (k / per); bx = (k % per) % g.ntx; by = (k % per) / g.ntx;
    if (bz >= (unsigned)g.ntz) return;
  }
  const long x_origin = PITCH ? g.lo - g.lo % 32 : g.lo;
  const long xs = x_origin + (long)bx * WOUT, ys = g.lo + (long)by * ROUT;
  if (xs >= g.hi || ys >= g.hi) return;
  long wx = xs - (PITCH ? (TW - PITCH) / 2 / C * C : HALO), wy = ys - HALO;
  if (wx + TW > g.W) wx = g.W - TW;
  if (wx < 0) wx = 0;
  if (wy + TR > g.H) wy = g.H - TR;
  if (wy < 0) wy = 0;
  const int z0 = g.lo + bz * g.chunk;
  const int z1 = z0 + g.chunk < g.hi ? z0 + g.chunk : g.hi;
  if (z0 >= g.hi) return;
  const long x = wx + lane * C, yb = wy + wave * R;
  const bool st_x = x >= xs && x >= g.lo && x + C <= xs + WOUT && x + C <= g.hi;
  const long plane = (long)g.W * g.H;
  vec ring[PF + 1][R];
  const long lane_off = yb * g.W + x;
  auto load = [&](int slot, int z) {
    if (z > g.D - 1) z = g.D - 1;
    if (z < 0) z = 0;
    const float* p = in + z * plane + lane_off + g.ld_shift;
#pragma unroll
    for (int r = 0; r < R; ++r) {
      if (NT) ring[slot][r] = __builtin_nontemporal_load((const vec*)(p + (long)r * g.W));
      else ring[slot][r] = *(const vec*)(p + (long)r * g.W);
    }
  };
  const int zfirst = z0 - HALO;            // first plane read
  const int steps = (z1 - z0) + 2 * HALO;  // planes read = planes of the chunk + fill
#pragma unroll
  for (int k = 0; k < PF; ++k) load(k, zfirst + k);
  int t = 0;
  // the loop is unrolled by PF+1 so that ring slots are compile-time registers
  for (; t < steps; t += PF + 1) {
#pragma unroll
    for (int u = 0; u <= PF; ++u) {
      const int step = t + u;
      if (step < steps) {
        load((u + PF) % (PF + 1), zfirst + step + PF);
        // plane zfirst+step sits in slot u; it is stored as plane z = zfirst+step-HALO
        // (standing in for the output that lags the input by the pipeline depth)
        const int z = zfirst + step - HALO;
        if (z >= z0 && z < z1) {
          float* q = out + z * plane + lane_off + g.st_shift;
#pragma unroll
          for (int r = 0; r < R; ++r) {
            const long y = yb + r;
            if (st_x && y >= ys && y < ys + ROUT && y < g.hi) {
              if (NT) __builtin_nontemporal_store(ring[u][r], (vec*)(q + (long)r * g.W));
              else *(vec*)(q + (long)r * g.W) = ring[u][r];
            }
          }
        }
        if (BARRIER) __syncthreads();
      }
    }
  }
}

template <int C, int R, int G, int PF, int HALO, int BARRIER, int NT, int PITCH = 0>
void run(const char* name, const float* a, float* b, int N, int box_lo, int wgs_per_cu, int order,
         int ld_shift = 0, int st_shift = 0) {
  constexpr int TW = 64 * C, TR = G * R, WOUT = PITCH ? PITCH : TW - 2 * HALO, ROUT = TR - 2 * HALO;
  Geo g;
  g.W = g.H = g.D = N; g.lo = box_lo; g.hi = N - box_lo; g.order = order; g.ld_shift = ld_shift; g.st_shift = st_shift;
  const int ext = g.hi - g.lo;
  g.ntx = ((PITCH ? g.hi - (g.lo - g.lo % 32) : ext) + WOUT - 1) / WOUT; g.nty = (ext + ROUT - 1) / ROUT;
  const int slots = 256 * wgs_per_cu;
  // z chunks: the whole grid in (about) one round of resident workgroups
  int ntz = slots / (g.ntx * g.nty);
  if (ntz < 1) ntz = 1;
  if (order == 2) ntz = (ntz / 8) * 8 > 0 ? (ntz / 8) * 8 : 8;
  g.chunk = (ext + ntz - 1) / ntz;
  g.ntz = (ext + g.chunk - 1) / g.chunk;
  unsigned blocks = g.ntx * g.nty * g.ntz;
  if (order == 2) blocks = 8 * g.ntx * g.nty * ((g.ntz + 7) / 8);
  if (order == 3) blocks = 8 * ((blocks + 7) / 8);
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  if (getenv("TILE3D_PMC")) printf("PMC dispatch order: %s\n", name);
  else for (int i = 0; i < 2; ++i) tilecopy<C, R, G, PF, HALO, BARRIER, NT, PITCH><<<blocks, G * 64>>>(a, b, g);
  CK(hipDeviceSynchronize());
  const int reps = getenv("TILE3D_PMC") ? 1 : 5;
  CK(hipEventRecord(e0));
  for (int i = 0; i < reps; ++i) tilecopy<C, R, G, PF, HALO, BARRIER, NT, PITCH><<<blocks, G * 64>>>(a, b, g);
  CK(hipEventRecord(e1));
  CK(hipDeviceSynchronize());
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  const double us = ms * 1e3 / reps;
  const double unique = 2.0 * 4.0 * ext * (double)ext * ext;
  const double tile_cells = (double)g.ntx * g.nty * TW * TR * (ext + 2.0 * HALO * g.ntz) * 4.0;
  printf("%-44s N=%d tiles %dx%dx%d chunk %d wgs %u  %7.1f us  unique %.2f TB/s  issued-read+write %.2f TB/s\n",
         name, N, g.ntx, g.nty, g.ntz, g.chunk, blocks, us, unique / us / 1e6,
         (tile_cells + unique / 2) / us / 1e6);
}

int main(int argc, char** argv) {
  const int N = argc > 1 ? atoi(argv[1]) : 512;
  const size_t bytes = (size_t)N * N * N * 4;
  float *a, *b;
  CK(hipMalloc(&a, bytes)); CK(hipMalloc(&b, bytes));
  CK(hipMemset(a, 1, bytes)); CK(hipMemset(b, 0, bytes));
  CK(hipDeviceSynchronize());
  if (getenv("TILE3D_PMC")) {
    run<2, 8, 8, 1, 4, 1, 0>("C2 R8 G8 pf1 barrier", a, b, N, 4, 1, 0);
    run<2, 8, 8, 1, 4, 1, 0>("C2 R8 G8 pf1 barrier xcd-runs", a, b, N, 4, 1, 3);
    run<4, 4, 8, 1, 4, 1, 0>("C4 R4 G8 pf1 barrier", a, b, N, 4, 1, 0);
    run<4, 4, 8, 1, 4, 1, 0>("C4 R4 G8 pf1 barrier xcd-runs", a, b, N, 4, 1, 3);
    run<2, 8, 8, 3, 0, 1, 0>("C2 R8 G8 pf3 barrier halo0", a, b, N, 0, 1, 0);
    return 0;
  }
  // block-form geometry: 8 wavefronts x (128 x 8) bands, halo 4, one workgroup per CU
  run<2, 8, 8, 1, 4, 1, 0>("C2 R8 G8 pf1 barrier", a, b, N, 4, 1, 0);
  run<2, 8, 8, 1, 4, 0, 0>("C2 R8 G8 pf1", a, b, N, 4, 1, 0);
  run<2, 8, 8, 2, 4, 1, 0>("C2 R8 G8 pf2 barrier", a, b, N, 4, 1, 0);
  run<2, 8, 8, 3, 4, 1, 0>("C2 R8 G8 pf3 barrier", a, b, N, 4, 1, 0);
  run<2, 8, 8, 4, 4, 1, 0>("C2 R8 G8 pf4 barrier", a, b, N, 4, 1, 0);
  run<2, 8, 8, 6, 4, 1, 0>("C2 R8 G8 pf6 barrier", a, b, N, 4, 1, 0);
  run<2, 8, 8, 3, 4, 1, 1>("C2 R8 G8 pf3 barrier nt", a, b, N, 4, 1, 0);
  run<2, 8, 8, 3, 4, 1, 0>("C2 R8 G8 pf3 barrier z-fastest ids", a, b, N, 4, 1, 1);
  run<2, 8, 8, 3, 4, 1, 0>("C2 R8 G8 pf3 barrier xcd-owns-chunks", a, b, N, 4, 1, 2);
  run<2, 8, 8, 3, 4, 1, 0>("C2 R8 G8 pf3 barrier 2 wg/cu", a, b, N, 4, 2, 0);
  run<2, 8, 8, 1, 4, 1, 0>("C2 R8 G8 pf1 barrier xcd-runs", a, b, N, 4, 1, 3);
  run<2, 8, 8, 3, 4, 1, 0>("C2 R8 G8 pf3 barrier xcd-runs", a, b, N, 4, 1, 3);
  run<2, 8, 8, 1, 4, 1, 0>("C2 R8 G8 pf1 barrier xcd-runs 2wg/cu", a, b, N, 4, 2, 3);
  run<4, 4, 8, 1, 4, 1, 0>("C4 R4 G8 pf1 barrier xcd-runs", a, b, N, 4, 1, 3);
  run<4, 8, 8, 2, 4, 1, 0>("C4 R8 G8 pf2 barrier (256x64) xcd-runs", a, b, N, 4, 1, 3);
  run<2, 16, 1, 1, 4, 0, 0>("C2 R16 G1 pf1 (128x16, 12/cu) xcd-runs", a, b, N, 4, 12, 3);
  run<2, 8, 4, 1, 4, 1, 0>("C2 R8 G4 pf1 barrier (128x32) 2wg/cu xcd-runs", a, b, N, 4, 2, 3);
  run<2, 8, 4, 1, 4, 1, 0>("C2 R8 G4 pf1 barrier (128x32) 2wg/cu", a, b, N, 4, 2, 0);
  // 16-byte lanes: 256 x 32 tile
  run<4, 4, 8, 1, 4, 1, 0>("C4 R4 G8 pf1 barrier", a, b, N, 4, 1, 0);
  run<4, 4, 8, 3, 4, 1, 0>("C4 R4 G8 pf3 barrier", a, b, N, 4, 1, 0);
  run<4, 4, 8, 6, 4, 1, 0>("C4 R4 G8 pf6 barrier", a, b, N, 4, 1, 0);
  run<4, 4, 16, 3, 4, 1, 0>("C4 R4 G16 pf3 barrier (256x64)", a, b, N, 4, 1, 0);
  run<4, 8, 8, 2, 4, 1, 0>("C4 R8 G8 pf2 barrier (256x64)", a, b, N, 4, 1, 0);
  // wave-pipelined geometry: 64 x 32 tiles (two row blocks are not modelled: 128 x 16)
  run<2, 16, 1, 1, 4, 0, 0>("C2 R16 G1 pf1 (128x16, 3 wg... 12/cu)", a, b, N, 4, 12, 0);
  run<2, 16, 1, 2, 4, 0, 0>("C2 R16 G1 pf2 (128x16, 12/cu)", a, b, N, 4, 12, 0);
  // no halo, but the box starts at 4 (or 16, 32): misaligned loads AND stores, no overlap
  run<2, 8, 8, 3, 0, 1, 0>("C2 R8 G8 pf3 barrier halo0 box at 4", a, b, N, 4, 1, 0);
  run<2, 8, 8, 3, 0, 1, 0>("C2 R8 G8 pf3 barrier halo0 box at 16", a, b, N, 16, 1, 0);
  run<2, 8, 8, 3, 0, 1, 0>("C2 R8 G8 pf3 barrier halo0 box at 32", a, b, N, 32, 1, 0);
  run<4, 4, 8, 3, 0, 1, 0>("C4 R4 G8 pf3 barrier halo0 box at 4", a, b, N, 4, 1, 0);
  // out tiles on whole lines
  run<2, 8, 8, 1, 4, 1, 0, 96>("C2 R8 G8 pf1 barrier pitch96", a, b, N, 4, 1, 0);
  run<2, 8, 8, 1, 4, 1, 0, 96>("C2 R8 G8 pf1 barrier pitch96 xcd-runs", a, b, N, 4, 1, 3);
  run<2, 8, 8, 1, 4, 1, 0, 112>("C2 R8 G8 pf1 barrier pitch112", a, b, N, 4, 1, 0);
  run<2, 8, 8, 1, 4, 1, 0, 112>("C2 R8 G8 pf1 barrier pitch112 xcd-runs", a, b, N, 4, 1, 3);
  run<4, 4, 8, 1, 4, 1, 0, 224>("C4 R4 G8 pf1 barrier pitch224", a, b, N, 4, 1, 0);
  run<4, 4, 8, 1, 4, 1, 0, 224>("C4 R4 G8 pf1 barrier pitch224 xcd-runs", a, b, N, 4, 1, 3);
  run<4, 4, 8, 1, 4, 1, 0, 240>("C4 R4 G8 pf1 barrier pitch240 xcd-runs", a, b, N, 4, 1, 3);
  run<2, 8, 8, 3, 0, 1, 0>("C2 R8 G8 pf3 barrier halo0 box at 4 xcd-runs", a, b, N, 4, 1, 3);
  // aligned geometry (box at 32: one tile column less), loads or stores shifted by 4 / 16 columns
  run<2, 8, 8, 3, 0, 1, 0>("C2 R8 G8 pf3 barrier halo0 box32 ld+4", a, b, N, 32, 1, 0, 4, 0);
  run<2, 8, 8, 3, 0, 1, 0>("C2 R8 G8 pf3 barrier halo0 box32 st+4", a, b, N, 32, 1, 0, 0, 4);
  run<2, 8, 8, 3, 0, 1, 0>("C2 R8 G8 pf3 barrier halo0 box32 ld+4 st+4", a, b, N, 32, 1, 0, 4, 4);
  run<2, 8, 8, 3, 0, 1, 0>("C2 R8 G8 pf3 barrier halo0 box32 st+16", a, b, N, 32, 1, 0, 0, 16);
  run<2, 8, 8, 3, 0, 1, 0>("C2 R8 G8 pf3 barrier halo0 box32 st+8", a, b, N, 32, 1, 0, 0, 8);
  run<4, 4, 8, 3, 0, 1, 0>("C4 R4 G8 pf3 barrier halo0 box32", a, b, N, 32, 1, 0, 0, 0);
  run<4, 4, 8, 3, 0, 1, 0>("C4 R4 G8 pf3 barrier halo0 box32 ld+4", a, b, N, 32, 1, 0, 4, 0);
  run<4, 4, 8, 3, 0, 1, 0>("C4 R4 G8 pf3 barrier halo0 box32 st+4", a, b, N, 32, 1, 0, 0, 4);
  // no halo at all: what a plain tiled copy gets
  run<2, 8, 8, 3, 0, 1, 0>("C2 R8 G8 pf3 barrier halo0", a, b, N, 0, 1, 0);
  run<4, 4, 8, 3, 0, 1, 0>("C4 R4 G8 pf3 barrier halo0", a, b, N, 0, 1, 0);
  run<4, 4, 8, 6, 0, 0, 1>("C4 R4 G8 pf6 halo0 nt no barrier", a, b, N, 0, 1, 0);
  return 0;
}
