// ku_dbsort.hip -- db_sort on the GPU: a Jellyfish-format k-mer list -> database.kdb + database.idx.
//
// The reference (src/db_sort.cpp:34-128 + KrakenDB::make_index, src/krakendb.cpp:118-148) counts the records per
// minimizer bin, scatters them into their bins and qsort()s every bin by k-mer.  Here the same order -- ascending
// (bin key, k-mer) -- comes from two stable LSD radix sorts on the device (rocPRIM: by k-mer, then by bin key), the
// index from a histogram of the bin keys + an exclusive scan.  Offline tool, not on the classify hot path.
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include <rocprim/rocprim.hpp>

#include "ku_device.h"
#include "ku_host.h"
#include "ku_internal.h"

namespace {

struct Rec {  // what travels through the second sort
  uint64_t kmer;
  uint32_t val;
  uint32_t pad;
};

// records of key_len + 4 bytes -> k-mer keys and values
__global__ void dbsort_unpack_kernel(const uint8_t *__restrict__ raw, uint64_t n, uint32_t key_len, uint64_t *kmers,
                                     uint32_t *vals, int zero_vals) {
  const uint64_t ps = key_len + 4;
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
    const uint8_t *p = raw + i * ps;
    uint64_t kmer = 0;
    for (uint32_t b = 0; b < key_len; ++b) kmer |= (uint64_t)p[b] << (8 * b);
    uint32_t v = 0;
    for (uint32_t b = 0; b < 4; ++b) v |= (uint32_t)p[key_len + b] << (8 * b);
    kmers[i] = kmer;
    vals[i] = zero_vals ? 0u : v;  // db_sort -z (src/db_sort.cpp:103-104)
  }
}

// bin key of the stored k-mer (KrakenDB::bin_key(kmer, nt), src/krakendb.cpp:182-196, with the KRAKIX2 scramble mask)
__global__ void dbsort_binkey_kernel(const uint64_t *__restrict__ kmers, const uint32_t *__restrict__ vals, uint64_t n,
                                     uint32_t k, uint32_t nt, uint32_t xor_mask, uint32_t *bins, Rec *recs,
                                     unsigned long long *hist) {
  const uint32_t w = k - nt + 1, mask = (uint32_t)((1ull << (2 * nt)) - 1);
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
    const uint64_t c = kmers[i];
    uint32_t best = 0xFFFFFFFFu;
    for (uint32_t j = 0; j < w; ++j) {
      const uint32_t mm = (uint32_t)(c >> (2 * j)) & mask;
      const uint32_t rc = ku_revcomp32(mm, nt);
      const uint32_t v = (mm < rc ? mm : rc) ^ xor_mask;
      best = v < best ? v : best;
    }
    bins[i] = best;
    recs[i] = Rec{c, vals[i], 0u};
    atomicAdd(&hist[best], 1ull);
  }
}

// sorted records -> the on-disk form (key_len little-endian key bytes, 4 value bytes)
__global__ void dbsort_pack_kernel(const Rec *__restrict__ recs, uint64_t n, uint32_t key_len, uint8_t *raw) {
  const uint64_t ps = key_len + 4;
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
    const Rec r = recs[i];
    uint8_t *p = raw + i * ps;
    for (uint32_t b = 0; b < key_len; ++b) p[b] = (uint8_t)(r.kmer >> (8 * b));
    for (uint32_t b = 0; b < 4; ++b) p[key_len + b] = (uint8_t)(r.val >> (8 * b));
  }
}

struct Dev {  // frees what it owns on every exit path
  std::vector<void *> ptrs;
  template <typename T> hipError_t alloc(T **p, size_t bytes) {
    hipError_t e = hipMalloc((void **)p, bytes ? bytes : 1);
    if (e == hipSuccess) ptrs.push_back(*p);
    return e;
  }
  void release(void *p) {
    for (auto &q : ptrs)
      if (q == p) { (void)hipFree(p); q = nullptr; }
  }
  ~Dev() {
    for (void *p : ptrs)
      if (p) (void)hipFree(p);
  }
};

int fail(int code, const std::string &msg) {
  ku_set_error(msg);
  return code;
}

bool write_all(int fd, const void *buf, size_t n) {
  const char *p = (const char *)buf;
  while (n) {
    ssize_t w = ::write(fd, p, n < ((size_t)1 << 30) ? n : ((size_t)1 << 30));
    if (w <= 0) return false;
    p += w;
    n -= (size_t)w;
  }
  return true;
}

}  // namespace

#define DS_HIP(expr)                                                                             \
  do {                                                                                           \
    hipError_t e_ = (expr);                                                                      \
    if (e_ != hipSuccess) return fail(e_ == hipErrorOutOfMemory ? KU_ENOMEM : KU_EHIP,           \
                                      std::string("db_sort: ") + #expr + ": " + hipGetErrorString(e_)); \
  } while (0)

extern "C" int ku_db_sort_files(int device, const char *in_path, const char *out_kdb_path, const char *out_idx_path,
                                uint32_t nt, int zero_vals) {
  if (!in_path || !out_kdb_path || !out_idx_path) return fail(KU_EINVAL, "ku_db_sort_files: null argument");
  if (nt < 1 || nt > 15) return fail(KU_EINVAL, "bin key length out of range (1..15: 32-bit minimizers, krakendb.cpp:203)");
  int n_dev = 0;
  if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev <= 0) return fail(KU_EHIP, "no HIP device available (this library has no CPU fallback)");
  if (device < 0 || device >= n_dev) return fail(KU_EINVAL, "device index out of range");
  DS_HIP(hipSetDevice(device));

  // ---- input: JFLISTDN header (krakendb.cpp:60-78,151-177) + unsorted records
  int fd = ::open(in_path, O_RDONLY);
  if (fd < 0) return fail(KU_ENOINPUT, std::string("can't open ") + in_path);
  struct stat st;
  if (fstat(fd, &st) != 0 || st.st_size < 72) { ::close(fd); return fail(KU_EDATA, "database in improper format"); }
  const size_t in_sz = (size_t)st.st_size;
  void *map = mmap(nullptr, in_sz, PROT_READ, MAP_PRIVATE, fd, 0);
  ::close(fd);
  if (map == MAP_FAILED) return fail(KU_ENOINPUT, std::string("can't map ") + in_path);
  struct Unmap { void *p; size_t n; ~Unmap() { munmap(p, n); } } unmap{map, in_sz};
  const uint8_t *in = (const uint8_t *)map;
  if (memcmp(in, "JFLISTDN", 8) != 0) return fail(KU_EDATA, "database in improper format");
  uint64_t key_bits, val_len, key_ct;
  memcpy(&key_bits, in + 8, 8);
  memcpy(&val_len, in + 16, 8);
  memcpy(&key_ct, in + 48, 8);
  if (val_len != 4) return fail(KU_EDATA, "can only handle 4 byte DB values");
  if (key_bits == 0 || key_bits > 62 || (key_bits & 1)) return fail(KU_EDATA, "unsupported key_bits");
  const uint32_t k = (uint32_t)(key_bits / 2), key_len = (uint32_t)((key_bits + 7) / 8);
  if (nt > k) return fail(KU_EINVAL, "bin key longer than the k-mers");
  const size_t hdr = 72 + 2 * (4 + 8 * key_bits), ps = key_len + 4;
  if (in_sz < hdr || key_ct > (in_sz - hdr) / ps) return fail(KU_EDATA, "database file truncated");
  const uint64_t n = key_ct, n_bins = 1ull << (2 * nt);
  const uint64_t INDEX2_XOR_MASK = 0xe37e28c4271b5a2dULL;  // krakendb.cpp:45 (db_sort always writes KRAKIX2)
  const uint32_t xor_mask = (uint32_t)(INDEX2_XOR_MASK & (n_bins - 1));

  // ---- device pipeline
  Dev dev;
  hipStream_t s = nullptr;
  DS_HIP(hipStreamCreate(&s));
  struct StreamGuard { hipStream_t s; ~StreamGuard() { (void)hipStreamDestroy(s); } } sg{s};
  uint8_t *d_raw = nullptr;
  uint64_t *d_kmer_a = nullptr, *d_kmer_b = nullptr, *d_off = nullptr;
  uint32_t *d_val_a = nullptr, *d_val_b = nullptr, *d_bin_a = nullptr, *d_bin_b = nullptr;
  Rec *d_rec_a = nullptr, *d_rec_b = nullptr;
  unsigned long long *d_hist = nullptr;
  void *d_tmp = nullptr;
  const unsigned grid = (unsigned)std::min<uint64_t>((n + 255) / 256 ? (n + 255) / 256 : 1, 1u << 16);
  DS_HIP(dev.alloc(&d_raw, n * ps));
  DS_HIP(dev.alloc(&d_kmer_a, n * 8));
  DS_HIP(dev.alloc(&d_val_a, n * 4));
  if (n) DS_HIP(hipMemcpyAsync(d_raw, in + hdr, n * ps, hipMemcpyHostToDevice, s));
  hipLaunchKernelGGL(dbsort_unpack_kernel, dim3(grid), dim3(256), 0, s, d_raw, n, key_len, d_kmer_a, d_val_a, zero_vals);
  DS_HIP(hipGetLastError());
  // pass 1: by k-mer (2k significant bits)
  DS_HIP(dev.alloc(&d_kmer_b, n * 8));
  DS_HIP(dev.alloc(&d_val_b, n * 4));
  size_t tmp_bytes = 0;
  DS_HIP(rocprim::radix_sort_pairs(nullptr, tmp_bytes, d_kmer_a, d_kmer_b, d_val_a, d_val_b, (size_t)n, 0u, (unsigned)key_bits, s));
  DS_HIP(dev.alloc(&d_tmp, tmp_bytes));
  DS_HIP(rocprim::radix_sort_pairs(d_tmp, tmp_bytes, d_kmer_a, d_kmer_b, d_val_a, d_val_b, (size_t)n, 0u, (unsigned)key_bits, s));
  DS_HIP(hipStreamSynchronize(s));
  dev.release(d_tmp); d_tmp = nullptr;
  dev.release(d_kmer_a); dev.release(d_val_a);
  // bin keys + histogram
  DS_HIP(dev.alloc(&d_bin_a, n * 4));
  DS_HIP(dev.alloc(&d_bin_b, n * 4));
  DS_HIP(dev.alloc(&d_rec_a, n * sizeof(Rec)));
  DS_HIP(dev.alloc(&d_hist, n_bins * 8));
  DS_HIP(hipMemsetAsync(d_hist, 0, n_bins * 8, s));
  hipLaunchKernelGGL(dbsort_binkey_kernel, dim3(grid), dim3(256), 0, s, d_kmer_b, d_val_b, n, k, nt, xor_mask, d_bin_a,
                     d_rec_a, d_hist);
  DS_HIP(hipGetLastError());
  DS_HIP(hipStreamSynchronize(s));
  dev.release(d_kmer_b); dev.release(d_val_b);
  // pass 2: stable by bin key (2 nt bits) -> (bin, k-mer) order
  DS_HIP(dev.alloc(&d_rec_b, n * sizeof(Rec)));
  tmp_bytes = 0;
  DS_HIP(rocprim::radix_sort_pairs(nullptr, tmp_bytes, d_bin_a, d_bin_b, d_rec_a, d_rec_b, (size_t)n, 0u, 2u * nt, s));
  DS_HIP(dev.alloc(&d_tmp, tmp_bytes));
  DS_HIP(rocprim::radix_sort_pairs(d_tmp, tmp_bytes, d_bin_a, d_bin_b, d_rec_a, d_rec_b, (size_t)n, 0u, 2u * nt, s));
  DS_HIP(hipStreamSynchronize(s));
  dev.release(d_tmp); d_tmp = nullptr;
  dev.release(d_rec_a); dev.release(d_bin_a); dev.release(d_bin_b);
  // offsets[b] = number of records in bins < b, offsets[4^nt] = key_ct (make_index, krakendb.cpp:136-139)
  DS_HIP(dev.alloc(&d_off, (n_bins + 1) * 8));
  tmp_bytes = 0;
  DS_HIP(rocprim::exclusive_scan(nullptr, tmp_bytes, (const uint64_t *)d_hist, d_off, (uint64_t)0, (size_t)n_bins,
                                 rocprim::plus<uint64_t>(), s));
  DS_HIP(dev.alloc(&d_tmp, tmp_bytes));
  DS_HIP(rocprim::exclusive_scan(d_tmp, tmp_bytes, (const uint64_t *)d_hist, d_off, (uint64_t)0, (size_t)n_bins,
                                 rocprim::plus<uint64_t>(), s));
  DS_HIP(hipMemcpyAsync(d_off + n_bins, &key_ct, 8, hipMemcpyHostToDevice, s));
  hipLaunchKernelGGL(dbsort_pack_kernel, dim3(grid), dim3(256), 0, s, d_rec_b, n, key_len, d_raw);
  DS_HIP(hipGetLastError());
  DS_HIP(hipStreamSynchronize(s));

  // ---- outputs: header copied verbatim (src/db_sort.cpp:56-75), then the sorted records; KRAKIX2 index
  std::vector<uint8_t> h_raw(n * ps);
  std::vector<uint64_t> h_off(n_bins + 1);
  if (n) DS_HIP(hipMemcpy(h_raw.data(), d_raw, n * ps, hipMemcpyDeviceToHost));
  DS_HIP(hipMemcpy(h_off.data(), d_off, (n_bins + 1) * 8, hipMemcpyDeviceToHost));
  int ofd = ::open(out_kdb_path, O_WRONLY | O_CREAT | O_TRUNC, 0644);
  if (ofd < 0) return fail(KU_ENOINPUT, std::string("can't write ") + out_kdb_path);
  bool ok = write_all(ofd, in, hdr) && write_all(ofd, h_raw.data(), h_raw.size());
  ok = (::close(ofd) == 0) && ok;
  if (!ok) return fail(KU_ENOINPUT, std::string("write error on ") + out_kdb_path);
  ofd = ::open(out_idx_path, O_WRONLY | O_CREAT | O_TRUNC, 0644);
  if (ofd < 0) return fail(KU_ENOINPUT, std::string("can't write ") + out_idx_path);
  const uint8_t nt8 = (uint8_t)nt;
  ok = write_all(ofd, "KRAKIX2", 7) && write_all(ofd, &nt8, 1) && write_all(ofd, h_off.data(), h_off.size() * 8);
  ok = (::close(ofd) == 0) && ok;
  if (!ok) return fail(KU_ENOINPUT, std::string("write error on ") + out_idx_path);
  return KU_OK;
}
