// ku_setlcas.hip -- set_lcas on the GPU: fold the taxon of every library sequence into the value of
// each database k-mer it contains (src/set_lcas.cpp:429-476), so that a k-mer ends up with the LCA of all the
// sequences that hold it.  The database stays in its on-disk order (bins by minimizer, k-mers ascending inside a
// bin); a k-mer is found as kmer_query does (src/krakendb.cpp:250-321): bin key -> index slice -> binary search.
// Values are kept as taxonomy *nodes* (rank of the taxid in the sorted node universe) while folding; lca() is
// associative and commutative on the Parent_map forest (default ancestor 1, lca(0, x) = x), so concurrent folds
// with compare-and-swap give the sequential result.  Offline tool, not on the classify hot path.
#include <algorithm>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include "ku_device.h"
#include "ku_host.h"
#include "ku_internal.h"

namespace {

constexpr uint32_t TID_CONTAMINANT1 = 32630;  // 'synthetic construct'   (src/set_lcas.cpp:88-89)
constexpr uint32_t TID_CONTAMINANT2 = 81077;  // 'artificial sequences'
constexpr uint32_t NO_CONTAM = 0xFFFFFFFFu;

int fail(int code, const std::string &msg) {
  ku_set_error(msg);
  return code;
}

struct SlDev {
  const uint64_t *kmers;     // key_ct, database order
  uint32_t *nodes;           // key_ct, the values as nodes
  uint32_t *contam;          // key_ct (or null): min over contaminant sequences of order * 2 + (taxid == CONTAMINANT2)
  const uint64_t *offsets;   // 4^nt + 1
  const uint32_t *parent;    // node -> parent node (0 = none)
  uint32_t k, nt, xor_mask;
};

// mode 0: val = lca(node, val); mode 1: reset to 0 (-R); mode 2: contaminant sequence under -T (record the first one);
// mode 3 (-I, UID databases): nothing is folded here -- where[p] = index of the pair that holds the k-mer at position p
// (KU_SL_NOWHERE: ambiguous or not in the database); the host applies uid_mapping in position order (its numbering of the
// taxid sets depends on that order, src/uid_mapping.cpp:32-91)
#define KU_SL_NOWHERE (~0ull)
__global__ void setlcas_kernel(SlDev d, const uint8_t *__restrict__ seq, uint64_t len, uint32_t node, uint32_t mode,
                               uint32_t contam_code, unsigned long long *n_missing, unsigned long long *where) {
  const uint32_t k = d.k, m = d.nt, w = k - m + 1;
  const uint32_t mmask = (uint32_t)((1ull << (2 * m)) - 1);
  const uint64_t n = len >= k ? len - k + 1 : 0;
  for (uint64_t p = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; p < n; p += (uint64_t)gridDim.x * blockDim.x) {
    uint64_t fwd = 0;
    bool amb = false;
    for (uint32_t i = 0; i < k; ++i) {  // KmerScanner semantics: any byte outside ACGTacgt makes the k-mer ambiguous
      const uint32_t c = seq[p + i] & 0xDFu;
      amb |= !(c == 'A' || c == 'C' || c == 'G' || c == 'T');
      fwd = (fwd << 2) | (((c >> 1) ^ (c >> 2)) & 3u);
    }
    if (amb) { if (mode == 3) where[p] = KU_SL_NOWHERE; continue; }
    const uint64_t rc = ku_revcomp64(fwd, k);
    const uint64_t canon = fwd < rc ? fwd : rc;
    uint32_t bin = 0xFFFFFFFFu;
    for (uint32_t j = 0; j < w; ++j) {
      const uint32_t mm = (uint32_t)(canon >> (2 * j)) & mmask;
      const uint32_t mrc = ku_revcomp32(mm, m);
      const uint32_t v = (mm < mrc ? mm : mrc) ^ d.xor_mask;
      bin = v < bin ? v : bin;
    }
    uint64_t lo = d.offsets[bin], hi = d.offsets[bin + 1];
    bool found = false;
    while (lo < hi) {
      const uint64_t mid = (lo + hi) >> 1;
      const uint64_t key = d.kmers[mid];
      if (key == canon) { lo = mid; found = true; break; }
      if (key < canon) lo = mid + 1; else hi = mid;
    }
    if (!found) {  // "kmer found in sequence that is not in database" (src/set_lcas.cpp:441-448)
      atomicAdd(n_missing, 1ull);
      if (mode == 3) where[p] = KU_SL_NOWHERE;
      continue;
    }
    if (mode == 3) { where[p] = lo; continue; }
    if (mode == 1) { d.nodes[lo] = 0; continue; }
    if (mode == 2) { atomicMin(&d.contam[lo], contam_code); continue; }
    uint32_t old = __hip_atomic_load(&d.nodes[lo], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    for (;;) {
      const uint32_t nw = ku_lca_nodes(d.parent, node, old);
      if (nw == old) break;
      const uint32_t prev = atomicCAS(&d.nodes[lo], old, nw);
      if (prev == old) break;
      old = prev;
    }
  }
}

__global__ void setlcas_unpack_kernel(const uint8_t *__restrict__ raw, uint64_t n, uint32_t key_len, uint64_t *kmers,
                                      uint32_t *vals) {
  const uint64_t ps = key_len + 4;
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
    const uint8_t *p = raw + i * ps;
    uint64_t kmer = 0;
    for (uint32_t b = 0; b < key_len; ++b) kmer |= (uint64_t)p[b] << (8 * b);
    uint32_t v = 0;
    for (uint32_t b = 0; b < 4; ++b) v |= (uint32_t)p[key_len + b] << (8 * b);
    kmers[i] = kmer;
    vals[i] = v;
  }
}

// taxid <-> node through the sorted node_taxid table; with `contam` set, values that already are a contaminant taxid
// are pinned (src/set_lcas.cpp:465-466 "keep value")
__global__ void setlcas_to_nodes_kernel(uint32_t *vals, uint64_t n, const uint32_t *__restrict__ node_taxid,
                                        uint32_t n_nodes, uint32_t *contam) {
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
    const uint32_t v = vals[i];
    uint32_t lo = 0, hi = n_nodes;
    while (lo < hi) {
      const uint32_t mid = (lo + hi) >> 1;
      if (node_taxid[mid] < v) lo = mid + 1; else hi = mid;
    }
    vals[i] = lo;  // the node universe contains every database value
    if (contam) contam[i] = v == TID_CONTAMINANT1 ? 0u : (v == TID_CONTAMINANT2 ? 1u : NO_CONTAM);
  }
}
__global__ void setlcas_to_taxids_kernel(uint32_t *vals, uint64_t n, const uint32_t *__restrict__ node_taxid,
                                         const uint32_t *__restrict__ contam) {
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
    uint32_t t = node_taxid[vals[i]];
    if (contam && contam[i] != NO_CONTAM) t = (contam[i] & 1u) ? TID_CONTAMINANT2 : TID_CONTAMINANT1;
    vals[i] = t;
  }
}

}  // namespace

struct ku_setlcas {
  int device = 0;
  uint32_t flags = 0;
  hipStream_t stream = nullptr;
  uint64_t key_ct = 0;
  uint32_t order = 0;  // sequences added so far (file order decides between contaminants)
  std::vector<uint32_t> node_taxid;
  uint64_t *d_kmers = nullptr, *d_offsets = nullptr;
  uint32_t *d_nodes = nullptr, *d_contam = nullptr, *d_parent = nullptr, *d_node_taxid = nullptr;
  unsigned long long *d_missing = nullptr;
  uint8_t *d_seq = nullptr;
  uint64_t seq_cap = 0;
  SlDev dev{};
  // -I (KU_SL_UIDS): the values are UIDs, kept on the host; uid_mapping's containers (src/uid_mapping.cpp:32-91)
  unsigned long long *d_where = nullptr;
  uint64_t where_cap = 0;
  std::vector<unsigned long long> h_where;
  std::vector<uint32_t> h_uids;                                  // per pair
  std::map<std::vector<uint32_t>, uint32_t> taxids_to_uid;       // taxid set (ascending) -> UID
  std::vector<const std::vector<uint32_t> *> uid_to_taxids;      // UID - 1 -> its set
  std::vector<uint32_t> uid_blocks;                              // {taxid, parent UID} per UID, in creation order: the map file
};

#define SL_HIP(expr)                                                                                     \
  do {                                                                                                   \
    hipError_t e_ = (expr);                                                                              \
    if (e_ != hipSuccess) return fail(e_ == hipErrorOutOfMemory ? KU_ENOMEM : KU_EHIP,                   \
                                      std::string("set_lcas: ") + #expr + ": " + hipGetErrorString(e_)); \
  } while (0)

extern "C" void ku_setlcas_close(ku_setlcas *s) {
  if (!s) return;
  (void)hipSetDevice(s->device);
  if (s->stream) (void)hipStreamSynchronize(s->stream);
  for (void *p : {(void *)s->d_kmers, (void *)s->d_offsets, (void *)s->d_nodes, (void *)s->d_contam, (void *)s->d_parent,
                  (void *)s->d_node_taxid, (void *)s->d_missing, (void *)s->d_seq, (void *)s->d_where})
    if (p) (void)hipFree(p);
  if (s->stream) (void)hipStreamDestroy(s->stream);
  delete s;
}

static int setlcas_open_impl(ku_setlcas *s, const ku_db *db, const ku_tax *tax) {
  ku_db_info info;
  if (ku_db_get_info(db, &info) != KU_OK) return KU_EINVAL;
  SL_HIP(hipSetDevice(s->device));
  SL_HIP(hipStreamCreate(&s->stream));
  const uint64_t n = info.key_ct, ps = info.key_len + 4, n_bins = info.n_bins;
  s->key_ct = n;
  const uint8_t *pairs = nullptr;
  const uint64_t *offsets = nullptr;
  if (ku_db_raw(db, &pairs, &offsets) != KU_OK) return KU_EINVAL;
  // node universe: taxDB ids U database values U {0, 1}, ascending (node 0 = taxid 0, node 1 = taxid 1)
  uint64_t nv = 0;
  if (ku_db_values(db, nullptr, &nv) != KU_OK) return KU_EINVAL;
  std::vector<uint32_t> nodes(nv + 1);
  uint64_t cap = nv;
  if (nv && ku_db_values(db, nodes.data(), &cap) != KU_OK) return KU_EINVAL;
  nodes.resize(cap);
  const std::vector<uint32_t> &ids = tax->ids, &parents = tax->parent_map;  // Parent_map values: 0 = no parent pointer
  const uint64_t nt_ids = ids.size();
  nodes.insert(nodes.end(), ids.begin(), ids.end());
  nodes.push_back(0);
  nodes.push_back(1);
  std::sort(nodes.begin(), nodes.end());
  nodes.erase(std::unique(nodes.begin(), nodes.end()), nodes.end());
  auto node_of = [&](uint32_t t) { return (uint32_t)(std::lower_bound(nodes.begin(), nodes.end(), t) - nodes.begin()); };
  std::vector<uint32_t> parent(nodes.size(), 0);
  for (uint64_t i = 0; i < nt_ids; ++i)  // Parent_map semantics: 0 = no parent pointer
    if (ids[i] != 0 && parents[i] != 0) parent[node_of(ids[i])] = node_of(parents[i]);
  s->node_taxid = nodes;

  uint8_t *d_raw = nullptr;
  SL_HIP(hipMalloc((void **)&s->d_kmers, std::max<uint64_t>(n, 1) * 8));
  SL_HIP(hipMalloc((void **)&s->d_nodes, std::max<uint64_t>(n, 1) * 4));
  SL_HIP(hipMalloc((void **)&s->d_offsets, (n_bins + 1) * 8));
  SL_HIP(hipMalloc((void **)&s->d_parent, parent.size() * 4));
  SL_HIP(hipMalloc((void **)&s->d_node_taxid, nodes.size() * 4));
  SL_HIP(hipMalloc((void **)&s->d_missing, 8));
  if (s->flags & KU_SL_FORCE_CONTAMINANT) SL_HIP(hipMalloc((void **)&s->d_contam, std::max<uint64_t>(n, 1) * 4));
  SL_HIP(hipMalloc((void **)&d_raw, std::max<uint64_t>(n * ps, 1)));
  hipStream_t st = s->stream;
  hipError_t e = hipSuccess;
  if (n) e = hipMemcpyAsync(d_raw, pairs, n * ps, hipMemcpyHostToDevice, st);
  if (e == hipSuccess) e = hipMemcpyAsync(s->d_offsets, offsets, (n_bins + 1) * 8, hipMemcpyHostToDevice, st);
  if (e == hipSuccess) e = hipMemcpyAsync(s->d_parent, parent.data(), parent.size() * 4, hipMemcpyHostToDevice, st);
  if (e == hipSuccess) e = hipMemcpyAsync(s->d_node_taxid, nodes.data(), nodes.size() * 4, hipMemcpyHostToDevice, st);
  if (e == hipSuccess) e = hipMemsetAsync(s->d_missing, 0, 8, st);
  const unsigned grid = (unsigned)std::min<uint64_t>(std::max<uint64_t>((n + 255) / 256, 1), 1u << 16);
  if (e == hipSuccess) {
    hipLaunchKernelGGL(setlcas_unpack_kernel, dim3(grid), dim3(256), 0, st, d_raw, n, info.key_len, s->d_kmers, s->d_nodes);
    hipLaunchKernelGGL(setlcas_to_nodes_kernel, dim3(grid), dim3(256), 0, st, s->d_nodes, n, s->d_node_taxid,
                       (uint32_t)nodes.size(), s->d_contam);
    e = hipGetLastError();
  }
  if (e == hipSuccess) e = hipStreamSynchronize(st);
  (void)hipFree(d_raw);
  SL_HIP(e);
  const uint64_t INDEX2_XOR_MASK = 0xe37e28c4271b5a2dULL;  // krakendb.cpp:45
  s->dev = SlDev{s->d_kmers, s->d_nodes, s->d_contam, s->d_offsets, s->d_parent, info.k, info.nt,
                 info.idx_type == 1 ? 0u : (uint32_t)(INDEX2_XOR_MASK & (n_bins - 1))};
  if (s->flags & KU_SL_UIDS) {  // the values as they are in the file (UIDs of an earlier run would need that run's map: uid_mapping exits on them)
    s->h_uids.resize(n);
    for (uint64_t i = 0; i < n; ++i) memcpy(&s->h_uids[i], pairs + i * ps + info.key_len, 4);
  }
  return KU_OK;
}

extern "C" int ku_setlcas_open(int device, const ku_db *db, const ku_tax *tax, uint32_t flags, ku_setlcas **out) {
  if (!db || !tax || !out) return fail(KU_EINVAL, "ku_setlcas_open: null argument");
  *out = nullptr;
  int n_dev = 0;
  if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev <= 0) return fail(KU_EHIP, "no HIP device available (this library has no CPU fallback)");
  if (device < 0 || device >= n_dev) return fail(KU_EINVAL, "device index out of range");
  ku_setlcas *s = new ku_setlcas();
  s->device = device;
  s->flags = flags;
  int st = setlcas_open_impl(s, db, tax);
  if (st != KU_OK) { ku_setlcas_close(s); return st; }
  *out = s;
  return KU_OK;
}

extern "C" int ku_setlcas_add(ku_setlcas *s, const char *seq, uint64_t len, uint32_t taxid) {
  if (!s || (len && !seq)) return fail(KU_EINVAL, "ku_setlcas_add: null argument");
  SL_HIP(hipSetDevice(s->device));
  s->order++;
  if (len < s->dev.k) return KU_OK;
  const auto it = std::lower_bound(s->node_taxid.begin(), s->node_taxid.end(), taxid);
  if (it == s->node_taxid.end() || *it != taxid)
    return fail(KU_EINVAL, "ku_setlcas_add: taxid " + std::to_string(taxid) + " is neither in the taxonomy nor a database value");
  const uint32_t node = (uint32_t)(it - s->node_taxid.begin());
  if (len > s->seq_cap) {
    SL_HIP(hipStreamSynchronize(s->stream));
    if (s->d_seq) (void)hipFree(s->d_seq);
    s->d_seq = nullptr;
    s->seq_cap = 0;
    const uint64_t want = len + len / 4 + 4096;
    SL_HIP(hipMalloc((void **)&s->d_seq, want));
    s->seq_cap = want;
  }
  // the previous sequence's kernel reads d_seq: the copy is ordered behind it on the same stream
  SL_HIP(hipMemcpyAsync(s->d_seq, seq, len, hipMemcpyHostToDevice, s->stream));
  uint32_t mode = 0, code = 0;
  if (s->flags & KU_SL_UIDS) {
    // -I (src/set_lcas.cpp:451-455): value = uid_mapping(value, taxid) for every k-mer of the sequence IN ORDER -- a new set
    // of taxids gets the next UID when it first comes up, so the numbering is the order of the k-mers.  The GPU finds the
    // pairs (canonical k-mer, bin, search: the expensive part), the host walks them.
    const uint64_t n = len - s->dev.k + 1;
    if (n > s->where_cap) {
      if (s->d_where) (void)hipFree(s->d_where);
      s->d_where = nullptr;
      s->where_cap = 0;
      SL_HIP(hipMalloc((void **)&s->d_where, (n + n / 4 + 1024) * 8));
      s->where_cap = n + n / 4 + 1024;
    }
    const unsigned grid = (unsigned)std::min<uint64_t>((n + 255) / 256, 1u << 16);
    hipLaunchKernelGGL(setlcas_kernel, dim3(grid), dim3(256), 0, s->stream, s->dev, s->d_seq, len, node, 3u, 0u, s->d_missing, s->d_where);
    SL_HIP(hipGetLastError());
    s->h_where.resize(n);
    SL_HIP(hipMemcpyAsync(s->h_where.data(), s->d_where, n * 8, hipMemcpyDeviceToHost, s->stream));
    SL_HIP(hipStreamSynchronize(s->stream));
    // within a sequence the taxid is fixed: uid_mapping(old, taxid) is a function of the old UID alone
    std::map<uint32_t, uint32_t> memo;
    for (uint64_t p = 0; p < n; ++p) {
      const unsigned long long at = s->h_where[p];
      if (at == KU_SL_NOWHERE) continue;
      const uint32_t old = s->h_uids[at];
      auto hit = memo.find(old);
      if (hit != memo.end()) { s->h_uids[at] = hit->second; continue; }
      uint32_t nw;
      std::vector<uint32_t> set;
      bool have = false;
      if (old == 0) set.push_back(taxid);
      else {
        if (old > s->uid_to_taxids.size())
          return fail(KU_EINVAL, "set_lcas -I: kmer_uid (" + std::to_string(old) + ") greater than UID vector size (" +
                                     std::to_string(s->uid_to_taxids.size()) + "): the database holds values of an earlier run");
        set = *s->uid_to_taxids[old - 1];
        auto it = std::lower_bound(set.begin(), set.end(), taxid);
        if (it == set.end() || *it != taxid) set.insert(it, taxid);
        else have = true;  // the taxid is part of the k-mer's set already
      }
      if (have) nw = old;
      else {
        const uint32_t next = (uint32_t)s->uid_to_taxids.size() + 1;
        auto ins = s->taxids_to_uid.insert({std::move(set), next});
        if (!ins.second) nw = ins.first->second;  // the set has a UID
        else {
          if (next == 0xFFFFFFFFu) return fail(KU_EUNSUP, "set_lcas -I: maxxed out on UIDs");
          s->uid_to_taxids.push_back(&ins.first->first);
          s->uid_blocks.push_back(taxid);
          s->uid_blocks.push_back(old);
          nw = next;
        }
      }
      memo[old] = nw;
      s->h_uids[at] = nw;
    }
    return KU_OK;
  }
  if (s->flags & KU_SL_RESET) mode = 1;  // -R wins (src/set_lcas.cpp:458-459)
  else if ((s->flags & KU_SL_FORCE_CONTAMINANT) && (taxid == TID_CONTAMINANT1 || taxid == TID_CONTAMINANT2)) {
    mode = 2;
    code = s->order * 2u + (taxid == TID_CONTAMINANT2 ? 1u : 0u);
  }
  const uint64_t n = len - s->dev.k + 1;
  const unsigned grid = (unsigned)std::min<uint64_t>((n + 255) / 256, 1u << 16);
  hipLaunchKernelGGL(setlcas_kernel, dim3(grid), dim3(256), 0, s->stream, s->dev, s->d_seq, len, node, mode, code,
                     s->d_missing, (unsigned long long *)nullptr);
  SL_HIP(hipGetLastError());
  SL_HIP(hipStreamSynchronize(s->stream));  // `seq` may be reused by the caller; offline tool, no pipelining needed
  return KU_OK;
}

extern "C" int ku_setlcas_uid_map(const ku_setlcas *s, const uint32_t **blocks, uint64_t *n_uids) {
  if (!s || !blocks || !n_uids) return fail(KU_EINVAL, "ku_setlcas_uid_map: null argument");
  if (!(s->flags & KU_SL_UIDS)) return fail(KU_ESTATE, "ku_setlcas_uid_map: the fold was not opened with KU_SL_UIDS");
  *blocks = s->uid_blocks.data();
  *n_uids = s->uid_blocks.size() / 2;
  return KU_OK;
}

extern "C" int ku_setlcas_finish(ku_setlcas *s, uint32_t *values_out, uint64_t *n_missing) {
  if (!s || (s->key_ct && !values_out)) return fail(KU_EINVAL, "ku_setlcas_finish: null argument");
  SL_HIP(hipSetDevice(s->device));
  const uint64_t n = s->key_ct;
  if (s->flags & KU_SL_UIDS) {
    unsigned long long miss = 0;
    SL_HIP(hipMemcpyAsync(&miss, s->d_missing, 8, hipMemcpyDeviceToHost, s->stream));
    SL_HIP(hipStreamSynchronize(s->stream));
    if (n) memcpy(values_out, s->h_uids.data(), n * 4);
    if (n_missing) *n_missing = miss;
    return KU_OK;
  }
  const unsigned grid = (unsigned)std::min<uint64_t>(std::max<uint64_t>((n + 255) / 256, 1), 1u << 16);
  hipLaunchKernelGGL(setlcas_to_taxids_kernel, dim3(grid), dim3(256), 0, s->stream, s->d_nodes, n, s->d_node_taxid,
                     s->d_contam);
  SL_HIP(hipGetLastError());
  unsigned long long miss = 0;
  SL_HIP(hipMemcpyAsync(&miss, s->d_missing, 8, hipMemcpyDeviceToHost, s->stream));
  if (n) SL_HIP(hipMemcpyAsync(values_out, s->d_nodes, n * 4, hipMemcpyDeviceToHost, s->stream));
  SL_HIP(hipStreamSynchronize(s->stream));
  if (n_missing) *n_missing = miss;
  // back to nodes, so that further sequences can be added after a snapshot
  hipLaunchKernelGGL(setlcas_to_nodes_kernel, dim3(grid), dim3(256), 0, s->stream, s->d_nodes, n, s->d_node_taxid,
                     (uint32_t)s->node_taxid.size(), (uint32_t *)nullptr);
  SL_HIP(hipGetLastError());
  SL_HIP(hipStreamSynchronize(s->stream));
  return KU_OK;
}
