// valu_rates.hip -- issue cost of the integer instructions the classify kernels lean on (gfx950), in SIMD cycles per
// wave64 instruction: N independent chains of one opcode per wave, enough waves to fill every SIMD, wall clock.
// Build: hipcc -O3 --offload-arch=gfx950 scripts/valu_rates.hip -o gpurun_out/valu_rates
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

#define REP 256
template <int OP> __global__ __launch_bounds__(256) void k(uint32_t *out, uint32_t seed, int iters) {
  uint32_t a0 = threadIdx.x + seed, a1 = a0 * 3 + 1, a2 = a0 ^ 0x55, a3 = a0 + 77, a4 = a0 * 5, a5 = a0 ^ 0x1234, a6 = a0 + 9, a7 = a0 * 7;
  uint32_t b = seed | 1u;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < REP / 8; ++r) {
      if (OP == 0) { asm volatile("v_add_u32 %0, %0, %8\n v_add_u32 %1, %1, %8\n v_add_u32 %2, %2, %8\n v_add_u32 %3, %3, %8\n v_add_u32 %4, %4, %8\n v_add_u32 %5, %5, %8\n v_add_u32 %6, %6, %8\n v_add_u32 %7, %7, %8" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b)); }
      if (OP == 1) { asm volatile("v_mul_lo_u32 %0, %0, %8\n v_mul_lo_u32 %1, %1, %8\n v_mul_lo_u32 %2, %2, %8\n v_mul_lo_u32 %3, %3, %8\n v_mul_lo_u32 %4, %4, %8\n v_mul_lo_u32 %5, %5, %8\n v_mul_lo_u32 %6, %6, %8\n v_mul_lo_u32 %7, %7, %8" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b)); }
      if (OP == 2) { asm volatile("v_mul_hi_u32 %0, %0, %8\n v_mul_hi_u32 %1, %1, %8\n v_mul_hi_u32 %2, %2, %8\n v_mul_hi_u32 %3, %3, %8\n v_mul_hi_u32 %4, %4, %8\n v_mul_hi_u32 %5, %5, %8\n v_mul_hi_u32 %6, %6, %8\n v_mul_hi_u32 %7, %7, %8" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b)); }
      if (OP == 3) { asm volatile("v_mul_u32_u24 %0, %0, %8\n v_mul_u32_u24 %1, %1, %8\n v_mul_u32_u24 %2, %2, %8\n v_mul_u32_u24 %3, %3, %8\n v_mul_u32_u24 %4, %4, %8\n v_mul_u32_u24 %5, %5, %8\n v_mul_u32_u24 %6, %6, %8\n v_mul_u32_u24 %7, %7, %8" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b)); }
      if (OP == 4) { asm volatile("v_bfrev_b32 %0, %0\n v_bfrev_b32 %1, %1\n v_bfrev_b32 %2, %2\n v_bfrev_b32 %3, %3\n v_bfrev_b32 %4, %4\n v_bfrev_b32 %5, %5\n v_bfrev_b32 %6, %6\n v_bfrev_b32 %7, %7" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b)); }
      if (OP == 5) { asm volatile("v_alignbit_b32 %0, %0, %1, %8\n v_alignbit_b32 %1, %1, %2, %8\n v_alignbit_b32 %2, %2, %3, %8\n v_alignbit_b32 %3, %3, %4, %8\n v_alignbit_b32 %4, %4, %5, %8\n v_alignbit_b32 %5, %5, %6, %8\n v_alignbit_b32 %6, %6, %7, %8\n v_alignbit_b32 %7, %7, %0, %8" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b)); }
      if (OP == 6) { asm volatile("v_lshlrev_b64 %0, %4, %0\n v_lshlrev_b64 %1, %4, %1\n v_lshlrev_b64 %2, %4, %2\n v_lshlrev_b64 %3, %4, %3\n v_lshlrev_b64 %0, %4, %0\n v_lshlrev_b64 %1, %4, %1\n v_lshlrev_b64 %2, %4, %2\n v_lshlrev_b64 %3, %4, %3" : "+v"(*(uint64_t *)&a0), "+v"(*(uint64_t *)&a2), "+v"(*(uint64_t *)&a4), "+v"(*(uint64_t *)&a6) : "v"(b & 1)); }
      if (OP == 7) { asm volatile("v_mad_u64_u32 %0, s[20:21], %4, %4, %0\n v_mad_u64_u32 %1, s[20:21], %4, %4, %1\n v_mad_u64_u32 %2, s[20:21], %4, %4, %2\n v_mad_u64_u32 %3, s[20:21], %4, %4, %3\n v_mad_u64_u32 %0, s[20:21], %4, %4, %0\n v_mad_u64_u32 %1, s[20:21], %4, %4, %1\n v_mad_u64_u32 %2, s[20:21], %4, %4, %2\n v_mad_u64_u32 %3, s[20:21], %4, %4, %3" : "+v"(*(uint64_t *)&a0), "+v"(*(uint64_t *)&a2), "+v"(*(uint64_t *)&a4), "+v"(*(uint64_t *)&a6) : "v"(b) : "s20", "s21"); }
      if (OP == 8) { asm volatile("v_min_u32 %0, %0, %8\n v_min_u32 %1, %1, %8\n v_min_u32 %2, %2, %8\n v_min_u32 %3, %3, %8\n v_min_u32 %4, %4, %8\n v_min_u32 %5, %5, %8\n v_min_u32 %6, %6, %8\n v_min_u32 %7, %7, %8" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b)); }
      if (OP == 9) { asm volatile("v_perm_b32 %0, %0, %1, %8\n v_perm_b32 %1, %1, %2, %8\n v_perm_b32 %2, %2, %3, %8\n v_perm_b32 %3, %3, %4, %8\n v_perm_b32 %4, %4, %5, %8\n v_perm_b32 %5, %5, %6, %8\n v_perm_b32 %6, %6, %7, %8\n v_perm_b32 %7, %7, %0, %8" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b)); }
      if (OP == 10) { asm volatile("v_mad_u32_u24 %0, %0, %8, %1\n v_mad_u32_u24 %1, %1, %8, %2\n v_mad_u32_u24 %2, %2, %8, %3\n v_mad_u32_u24 %3, %3, %8, %4\n v_mad_u32_u24 %4, %4, %8, %5\n v_mad_u32_u24 %5, %5, %8, %6\n v_mad_u32_u24 %6, %6, %8, %7\n v_mad_u32_u24 %7, %7, %8, %0" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b)); }
      if (OP == 11) { asm volatile("v_lshl_or_b32 %0, %0, 3, %1\n v_lshl_or_b32 %1, %1, 3, %2\n v_lshl_or_b32 %2, %2, 3, %3\n v_lshl_or_b32 %3, %3, 3, %4\n v_lshl_or_b32 %4, %4, 3, %5\n v_lshl_or_b32 %5, %5, 3, %6\n v_lshl_or_b32 %6, %6, 3, %7\n v_lshl_or_b32 %7, %7, 3, %0" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b)); }
      if (OP == 12) { asm volatile("v_lshrrev_b64 %0, %4, %0\n v_lshrrev_b64 %1, %4, %1\n v_lshrrev_b64 %2, %4, %2\n v_lshrrev_b64 %3, %4, %3\n v_lshrrev_b64 %0, %4, %0\n v_lshrrev_b64 %1, %4, %1\n v_lshrrev_b64 %2, %4, %2\n v_lshrrev_b64 %3, %4, %3" : "+v"(*(uint64_t *)&a0), "+v"(*(uint64_t *)&a2), "+v"(*(uint64_t *)&a4), "+v"(*(uint64_t *)&a6) : "v"(b & 1)); }
      if (OP == 13) { asm volatile("v_or_b32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_or_b32_dpp %1, %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_or_b32_dpp %2, %2, %2 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_or_b32_dpp %3, %3, %3 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_or_b32_dpp %4, %4, %4 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_or_b32_dpp %5, %5, %5 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_or_b32_dpp %6, %6, %6 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_or_b32_dpp %7, %7, %7 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b)); }
    }
  }
  out[blockIdx.x * 256 + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;
}

template <int OP> double run(const char *name, uint32_t *d_out, int n_cu, double ghz) {
  const int blocks = n_cu * 8, iters = 200;  // 8 blocks x 4 waves per CU = 8 waves per SIMD
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  k<OP><<<blocks, 256>>>(d_out, 1, 2);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  k<OP><<<blocks, 256>>>(d_out, 1, iters);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  // wave-instructions per SIMD: 8 waves x iters x REP
  const double per_simd = 8.0 * iters * REP;
  const double cyc = ms * 1e-3 * ghz * 1e9 / per_simd;
  printf("%-18s %8.3f ms  %6.2f cycles per wave-instruction (at %.2f GHz)\n", name, ms, cyc, ghz);
  return cyc;
}

int main() {
  hipDeviceProp_t p;
  hipGetDeviceProperties(&p, 0);
  const double ghz = p.clockRate * 1e-6;
  printf("%s: %d CUs, %.2f GHz\n", p.name, p.multiProcessorCount, ghz);
  uint32_t *d_out;
  hipMalloc(&d_out, (size_t)p.multiProcessorCount * 8 * 256 * 4);
  run<0>("v_add_u32", d_out, p.multiProcessorCount, ghz);
  run<1>("v_mul_lo_u32", d_out, p.multiProcessorCount, ghz);
  run<2>("v_mul_hi_u32", d_out, p.multiProcessorCount, ghz);
  run<3>("v_mul_u32_u24", d_out, p.multiProcessorCount, ghz);
  run<10>("v_mad_u32_u24", d_out, p.multiProcessorCount, ghz);
  run<7>("v_mad_u64_u32", d_out, p.multiProcessorCount, ghz);
  run<4>("v_bfrev_b32", d_out, p.multiProcessorCount, ghz);
  run<5>("v_alignbit_b32", d_out, p.multiProcessorCount, ghz);
  run<9>("v_perm_b32", d_out, p.multiProcessorCount, ghz);
  run<6>("v_lshlrev_b64", d_out, p.multiProcessorCount, ghz);
  run<12>("v_lshrrev_b64", d_out, p.multiProcessorCount, ghz);
  run<8>("v_min_u32", d_out, p.multiProcessorCount, ghz);
  run<11>("v_lshl_or_b32", d_out, p.multiProcessorCount, ghz);
  run<13>("v_or_b32_dpp", d_out, p.multiProcessorCount, ghz);
  return 0;
}
