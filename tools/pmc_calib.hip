// Calibration of rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950 for the access
// widths and shapes the SODA kernels use (MI355X_MICROARCH.md, HBM: FETCH_SIZE is
// half the bytes for 16-byte-per-lane streaming reads; "other access widths are
// uncalibrated: calibrate on a known byte count in your own access pattern").
// Every kernel moves a KNOWN number of bytes; run under
//   rocprofv3 --pmc FETCH_SIZE ... and --pmc WRITE_SIZE ... (separate passes)
// and compare Counter_Value (KiB) with the "bytes" this program prints.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("ERR %s line %d: %s\n", #x, __LINE__, hipGetErrorString(e)); exit(1);} } while (0)
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));

__global__ void __launch_bounds__(256) calib_copy16(const f4* __restrict__ in, f4* __restrict__ out, size_t n) {
  size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x, s = (size_t)gridDim.x * blockDim.x;
  for (; i < n; i += s) out[i] = in[i];
}
__global__ void __launch_bounds__(256) calib_copy8(const f2* __restrict__ in, f2* __restrict__ out, size_t n) {
  size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x, s = (size_t)gridDim.x * blockDim.x;
  for (; i < n; i += s) out[i] = in[i];
}
__global__ void __launch_bounds__(256) calib_copy4(const float* __restrict__ in, float* __restrict__ out, size_t n) {
  size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x, s = (size_t)gridDim.x * blockDim.x;
  for (; i < n; i += s) out[i] = in[i];
}
// The 3-D depth-4 kernel's plane-tile shape: a wave reads 32 rows of 64 floats
// (two 32-lane halves, 8 bytes per lane, 16 rows each) at column 56*bx - 4 and
// writes the 56 x 24 interior; planes streamed along z.  W = H = 512.
__global__ void __launch_bounds__(64) calib_tile3d(const float* __restrict__ in, float* __restrict__ out, int W, int H, int D, int zchunk) {
  const int lane = threadIdx.x, lx = lane & 31, ly = lane >> 5;
  long wx = 56L * blockIdx.x - 4, wy = 24L * blockIdx.y - 4;
  if (wx + 64 > W) wx = W - 64;
  if (wx < 0) wx = 0;
  if (wy + 32 > H) wy = H - 32;
  if (wy < 0) wy = 0;
  const long xs = 56L * blockIdx.x, ys = 24L * blockIdx.y;
  const int z0 = blockIdx.z * zchunk, z1 = z0 + zchunk < D ? z0 + zchunk : D;
  for (int z = z0; z < z1; ++z) {
    const float* p = in + (long)z * W * H;
    float* q = out + (long)z * W * H;
    f2 v[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) v[r] = *(const f2*)(p + (wy + ly * 16 + r) * W + wx + lx * 2);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const long y = wy + ly * 16 + r, x = wx + lx * 2;
      if (y >= ys && y < ys + 24 && y < H && x >= xs && x + 2 <= xs + 56 && x + 2 <= W)
        *(f2*)(q + y * W + x) = v[r];
    }
  }
}
// 2-D strip shape of the depth-1 kernels: 256 loaded floats per wave-row at
// xs - halo, w_out stored (16 bytes per lane)
__global__ void __launch_bounds__(64) calib_strip2d(const float* __restrict__ in, float* __restrict__ out, long W, long H, long chunk, int w_out, int halo) {
  const int lane = threadIdx.x;
  const long xs = (long)blockIdx.x * w_out + halo;
  if (xs + 256 - halo > W) return;
  const long x = xs - halo + lane * 4;
  const bool st = x >= xs && x + 4 <= xs + w_out;
  const long y0 = (long)blockIdx.y * chunk, y1 = y0 + chunk < H ? y0 + chunk : H;
  for (long y = y0; y < y1; ++y) {
    const f4 v = *(const f4*)(in + y * W + x);
    if (st) *(f4*)(out + y * W + x) = v;
  }
}

int main() {
  const size_t bytes = 512ull << 20;
  float *a, *b;
  CK(hipMalloc(&a, bytes)); CK(hipMalloc(&b, bytes));
  CK(hipMemset(a, 1, bytes)); CK(hipMemset(b, 0, bytes));
  CK(hipDeviceSynchronize());
  for (int rep = 0; rep < 2; ++rep) {
    calib_copy16<<<4096, 256>>>((const f4*)a, (f4*)b, bytes / 16);
    calib_copy8<<<4096, 256>>>((const f2*)a, (f2*)b, bytes / 8);
    calib_copy4<<<4096, 256>>>(a, b, bytes / 4);
    // 512^3 floats = 512 MiB; tiles: ceil(512/56)=10 x ceil(512/24)=22, z chunks of 128
    calib_tile3d<<<dim3(10, 22, 4), 64>>>(a, b, 512, 512, 512, 128);
    // 8192 x 16384 floats = 512 MiB: strips of 248 out / halo 4 and 192 out / halo 32
    calib_strip2d<<<dim3((8192 - 8) / 248, 16384 / 256), 64>>>(a, b, 8192, 16384, 256, 248, 4);
    calib_strip2d<<<dim3((8192 - 64) / 192, 16384 / 256), 64>>>(a, b, 8192, 16384, 256, 192, 32);
    CK(hipDeviceSynchronize());
  }
  const double MiB = 1048576.0;
  printf("calib_copy16/8/4: read %.1f MiB, write %.1f MiB each\n", bytes / MiB, bytes / MiB);
  printf("calib_tile3d: read (algorithmic, tiles incl. halo) %.1f MiB, write %.1f MiB\n",
         10.0 * 22 * 64 * 32 * 512 * 4 / MiB, 512.0 * 512 * 512 * 4 / MiB);
  const long s1 = (8192 - 8) / 248, s2 = (8192 - 64) / 192;
  printf("calib_strip2d 248/4: read %.1f MiB, write %.1f MiB; 192/32: read %.1f MiB, write %.1f MiB\n",
         s1 * 256.0 * 16384 * 4 / MiB, s1 * 248.0 * 16384 * 4 / MiB,
         s2 * 256.0 * 16384 * 4 / MiB, s2 * 192.0 * 16384 * 4 / MiB);
  return 0;
}
