// ku_report.hip -- clade roll-up of the report on the device.
//
// The reference's TaxReport constructor (src/taxdb.hpp:928-982) merges every counted taxon's HyperLogLog sketch into the
// sketch of each of its ancestors, one 4 KiB register merge per (taxon, ancestor) pair on the host.  Here the per-taxon
// registers never leave HBM: the host only lists, per counted clade, the slots of its members (a CSR over the slots'
// root paths) and the device
//   * ku_rollup_dense_kernel: one workgroup per clade with a dense member -- byte-wise maximum over the members'
//     registers (16 registers per lane, uint4 loads, SWAR max), reduced on the spot to the 54-bin register histogram the
//     Ertl estimator works from (src/hyperloglogplus.cpp:722-753);
//   * ku_rollup_sparse_kernel: clades whose members all kept the sparse representation merge by set union
//     (hyperloglogplus.cpp:601-604): every (slot, encoded hash) pair of the run-wide set is inserted under each of the
//     slot's all-sparse ancestor clades into a device hash set; first insertions add to that clade's histogram of
//     encoded ranks (sparseRegisterHistogram, :356-366).
// 80 bins * 4 B per clade go back to the host.
#include <hip/hip_runtime.h>

#include "ku_internal.h"

namespace {

// byte-wise unsigned maximum of packed registers (values <= 64 < 128: setting bit 7 of every byte of `a` keeps the
// subtraction from borrowing across bytes, bit 7 of the difference then says a >= b)
__device__ __forceinline__ uint32_t bmax4(uint32_t a, uint32_t b) {
  const uint32_t ge = (((a | 0x80808080u) - b) & 0x80808080u) >> 7;
  const uint32_t m = ge * 0xFFu;
  return (a & m) | (b & ~m);
}

__global__ __launch_bounds__(256) void ku_rollup_dense_kernel(const uint8_t *__restrict__ registers,
                                                               const uint32_t *__restrict__ member_off,
                                                               const uint32_t *__restrict__ member_slot,
                                                               const uint8_t *__restrict__ clade_dense,
                                                               uint32_t *__restrict__ hist) {
  __shared__ uint32_t bins[KU_ROLLUP_BINS];
  const uint32_t c = blockIdx.x, t = threadIdx.x;
  if (!clade_dense[c]) return;
  if (t < KU_ROLLUP_BINS) bins[t] = 0;
  uint4 acc = make_uint4(0, 0, 0, 0);
  const uint32_t lo = member_off[c], hi = member_off[c + 1];
  for (uint32_t j = lo; j < hi; ++j) {
    const uint4 v = reinterpret_cast<const uint4 *>(registers + (size_t)member_slot[j] * KU_HLL_M)[t];
    acc.x = bmax4(acc.x, v.x); acc.y = bmax4(acc.y, v.y); acc.z = bmax4(acc.z, v.z); acc.w = bmax4(acc.w, v.w);
  }
  __syncthreads();
  const uint32_t w[4] = {acc.x, acc.y, acc.z, acc.w};
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      const uint32_t r = (w[i] >> (8 * b)) & 0xFFu;
      atomicAdd(&bins[r < 64 - KU_HLL_P + 1 ? r : 64 - KU_HLL_P + 1], 1u);
    }
  __syncthreads();
  if (t < KU_ROLLUP_BINS) hist[(size_t)c * KU_ROLLUP_BINS + t] = bins[t];
}

// rank of an encoded hash relative to p = 12 (getEncodedRank, hyperloglogplus.cpp:152-161)
__device__ __forceinline__ uint32_t encoded_rank(uint32_t e) {
  if (e & 1u) return (25 - KU_HLL_P) + ((e >> 1) & 0x3Fu);
  const uint32_t bits = e << KU_HLL_P;
  return (bits == 0 ? 32 - KU_HLL_P : (uint32_t)__clz(bits)) + 1;
}

// The clades near the root take an entry from (almost) every pair: their histogram bins are counted in the block's LDS
// and flushed once per block -- one global address per (clade, rank) would serialise the whole grid.  `clade_hot[c]` is
// the clade's row in that LDS table (0xFFFF: none), `hot_clades[h]` the reverse map.
__global__ __launch_bounds__(256) void ku_rollup_sparse_kernel(const unsigned long long *__restrict__ pairs, uint64_t n_pairs,
                                                                const uint32_t *__restrict__ slot_off,
                                                                const uint32_t *__restrict__ slot_clade,
                                                                const uint16_t *__restrict__ clade_hot,
                                                                const uint32_t *__restrict__ hot_clades, uint32_t n_hot,
                                                                unsigned long long *__restrict__ set, uint64_t mask,
                                                                uint32_t *__restrict__ hist, uint32_t *__restrict__ err) {
  __shared__ uint32_t hot[KU_ROLLUP_HOT * KU_ROLLUP_BINS];
  for (uint32_t i = threadIdx.x; i < KU_ROLLUP_HOT * KU_ROLLUP_BINS; i += blockDim.x) hot[i] = 0;
  __syncthreads();
  for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n_pairs; i += (uint64_t)gridDim.x * blockDim.x) {
    const unsigned long long p = pairs[i];
    const uint32_t slot = (uint32_t)(p >> 32), enc = (uint32_t)p;
    uint32_t r = encoded_rank(enc);
    if (r > KU_ROLLUP_BINS - 1) r = KU_ROLLUP_BINS - 1;
    for (uint32_t j = slot_off[slot]; j < slot_off[slot + 1]; ++j) {
      const uint32_t c = slot_clade[j];
      const unsigned long long key = ((unsigned long long)(c + 1) << 32) | enc;
      uint64_t h = (key * 0x9E3779B97F4A7C15ull) >> 20;
      bool done = false;
      for (uint32_t probe = 0; probe < 4096 && !done; ++probe, ++h) {
        const unsigned long long old = atomicCAS(&set[h & mask], 0ull, key);
        if (old == 0ull) {
          const uint32_t hi = clade_hot[c];
          if (hi != 0xFFFFu) atomicAdd(&hot[hi * KU_ROLLUP_BINS + r], 1u);
          else atomicAdd(&hist[(size_t)c * KU_ROLLUP_BINS + r], 1u);
          done = true;
        } else if (old == key) done = true;
      }
      if (!done) atomicOr(err, 1u);
    }
  }
  __syncthreads();
  for (uint32_t i = threadIdx.x; i < n_hot * KU_ROLLUP_BINS; i += blockDim.x)
    if (hot[i]) atomicAdd(&hist[(size_t)hot_clades[i / KU_ROLLUP_BINS] * KU_ROLLUP_BINS + i % KU_ROLLUP_BINS], hot[i]);
}

__global__ __launch_bounds__(256) void ku_count_pairs_kernel(const unsigned long long *__restrict__ pairs, uint64_t n_pairs,
                                                              uint32_t *__restrict__ per_slot) {
  for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n_pairs; i += (uint64_t)gridDim.x * blockDim.x)
    atomicAdd(&per_slot[(uint32_t)(pairs[i] >> 32)], 1u);
}

}  // namespace

int ku_launch_rollup_dense(const uint8_t *d_registers, const uint32_t *d_member_off, const uint32_t *d_member_slot,
                           const uint8_t *d_clade_dense, uint32_t n_clades, uint32_t *d_hist, hipStream_t stream) {
  if (!n_clades) return KU_OK;
  ku_rollup_dense_kernel<<<n_clades, 256, 0, stream>>>(d_registers, d_member_off, d_member_slot, d_clade_dense, d_hist);
  return hipGetLastError() == hipSuccess ? KU_OK : KU_EHIP;
}

int ku_launch_rollup_sparse(const unsigned long long *d_pairs, uint64_t n_pairs, const uint32_t *d_slot_off,
                            const uint32_t *d_slot_clade, const uint16_t *d_clade_hot, const uint32_t *d_hot_clades,
                            uint32_t n_hot, unsigned long long *d_set, uint64_t mask, uint32_t *d_hist, uint32_t *d_err,
                            int n_cu, hipStream_t stream) {
  if (!n_pairs) return KU_OK;
  const uint64_t want = (n_pairs + 255) / 256;
  const unsigned blocks = (unsigned)(want < (uint64_t)n_cu * 8 ? want : (uint64_t)n_cu * 8);
  ku_rollup_sparse_kernel<<<blocks, 256, 0, stream>>>(d_pairs, n_pairs, d_slot_off, d_slot_clade, d_clade_hot, d_hot_clades, n_hot, d_set,
                                                       mask, d_hist, d_err);
  return hipGetLastError() == hipSuccess ? KU_OK : KU_EHIP;
}

int ku_launch_count_pairs(const unsigned long long *d_pairs, uint64_t n_pairs, uint32_t *d_per_slot, int n_cu, hipStream_t stream) {
  if (!n_pairs) return KU_OK;
  const uint64_t want = (n_pairs + 255) / 256;
  const unsigned blocks = (unsigned)(want < (uint64_t)n_cu * 8 ? want : (uint64_t)n_cu * 8);
  ku_count_pairs_kernel<<<blocks, 256, 0, stream>>>(d_pairs, n_pairs, d_per_slot);
  return hipGetLastError() == hipSuccess ? KU_OK : KU_EHIP;
}
