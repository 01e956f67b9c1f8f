// scripts/calib_gather.hip -- micro-benchmark (measurement aid, not product code):
// random-gather ceiling of MI355X HBM for the access shapes the lookup kernel can choose from.
//   hipcc --offload-arch=gfx950 -O3 scripts/calib_gather.hip -o /tmp/calib_gather && /tmp/calib_gather [GB]
// Each lane issues `ILP` independent loads per iteration at hashed positions of a table of `GB` gigabytes.
// Shapes: 16 B per lane at 16-B-aligned random offsets; 12 B at 4-B-aligned offsets (on-disk pairs);
//         "line64"/"line128": groups of 4 / 8 lanes read one aligned 64 B / 128 B line together.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__device__ __forceinline__ uint64_t mix(uint64_t k) {
  k ^= k >> 33; k *= 0xff51afd7ed558ccdULL; k ^= k >> 33; k *= 0xc4ceb9fe1a85ec53ULL; k ^= k >> 33; return k;
}

template <int SHAPE, int ILP>
__global__ __launch_bounds__(256) void gather(const uint32_t *__restrict__ tab, uint64_t n_bytes, uint64_t iters,
                                              uint32_t *out) {
  uint64_t gid = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
  uint32_t acc = 0;
  for (uint64_t it = 0; it < iters; ++it) {
    uint32_t v[ILP][4];
#pragma unroll
    for (int j = 0; j < ILP; ++j) {
      uint64_t q = (it * ILP + j) * (uint64_t)gridDim.x * blockDim.x;
      if (SHAPE == 0) {  // 16 B per lane, random 16-B slot
        uint64_t slot = __umul64hi(mix(gid + q + 1), n_bytes / 16);
        uint4 x = *reinterpret_cast<const uint4 *>(tab + slot * 4);
        v[j][0] = x.x; v[j][1] = x.y; v[j][2] = x.z; v[j][3] = x.w;
      } else if (SHAPE == 1) {  // 12 B per lane, random 12-B record
        uint64_t rec = __umul64hi(mix(gid + q + 1), n_bytes / 12 - 1);
        const uint32_t *p = tab + rec * 3;
        v[j][0] = p[0]; v[j][1] = p[1]; v[j][2] = p[2]; v[j][3] = 0;
      } else {  // SHAPE 2: 4 lanes share a 64-B line; SHAPE 3: 8 lanes share a 128-B line
        const int G = SHAPE == 2 ? 4 : 8;
        uint64_t line = __umul64hi(mix((gid / G) + q + 1), n_bytes / (16 * G));
        uint4 x = *reinterpret_cast<const uint4 *>(tab + (line * G + (gid % G)) * 4);
        v[j][0] = x.x; v[j][1] = x.y; v[j][2] = x.z; v[j][3] = x.w;
      }
    }
#pragma unroll
    for (int j = 0; j < ILP; ++j) acc += v[j][0] ^ v[j][1] ^ v[j][2] ^ v[j][3];
  }
  if (acc == 0x12345678u) out[0] = acc;
}

template <int SHAPE, int ILP> void run(const char *name, const uint32_t *tab, uint64_t n_bytes, uint32_t *out, int waves_per_simd) {
  int blocks = 256 * waves_per_simd;  // 256 CUs x (4 SIMDs x waves / 4 waves per block)
  uint64_t iters = 64;
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  hipLaunchKernelGGL((gather<SHAPE, ILP>), dim3(blocks), dim3(256), 0, 0, tab, n_bytes, (uint64_t)4, out);
  CK(hipEventRecord(a));
  hipLaunchKernelGGL((gather<SHAPE, ILP>), dim3(blocks), dim3(256), 0, 0, tab, n_bytes, iters, out);
  CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
  float ms; CK(hipEventElapsedTime(&ms, a, b));
  double loads = (double)blocks * 256 * iters * ILP;
  int G = SHAPE == 2 ? 4 : SHAPE == 3 ? 8 : 1;
  printf("%-28s ilp=%d waves/simd=%d : %8.2f G lane-loads/s  %8.2f G distinct-requests/s  %7.1f GB/s useful\n", name, ILP,
         waves_per_simd, loads / ms / 1e6, loads / G / ms / 1e6, loads * (SHAPE == 1 ? 12 : 16) / ms / 1e6);
}

int main(int argc, char **argv) {
  double gb = argc > 1 ? atof(argv[1]) : 16.0;
  uint64_t n_bytes = (uint64_t)(gb * (1ull << 30)) / 3072 * 3072;
  uint32_t *tab, *out;
  CK(hipMalloc(&tab, n_bytes)); CK(hipMalloc(&out, 64));
  CK(hipMemset(tab, 1, n_bytes));
  printf("table %.1f GiB\n", n_bytes / double(1ull << 30));
  for (int w : {2, 4, 8}) {
    if (w == 2) { run<0, 4>("16B/lane random", tab, n_bytes, out, w); run<1, 4>("12B/lane random (unaligned)", tab, n_bytes, out, w); }
    if (w == 4) { run<0, 1>("16B/lane random", tab, n_bytes, out, w); run<0, 4>("16B/lane random", tab, n_bytes, out, w); run<1, 4>("12B/lane random (unaligned)", tab, n_bytes, out, w);
                  run<2, 4>("64B line / 4 lanes", tab, n_bytes, out, w); run<3, 4>("128B line / 8 lanes", tab, n_bytes, out, w); }
    if (w == 8) { run<0, 4>("16B/lane random", tab, n_bytes, out, w); run<0, 8>("16B/lane random", tab, n_bytes, out, w); run<2, 8>("64B line / 4 lanes", tab, n_bytes, out, w); run<3, 8>("128B line / 8 lanes", tab, n_bytes, out, w); }
  }
  return 0;
}
