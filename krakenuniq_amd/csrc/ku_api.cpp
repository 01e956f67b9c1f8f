// ku_api.cpp -- C-ABI entry points (include/krakenuniq_amd.h): host-side database /
// taxonomy objects and the per-GPU context that owns the resident shard, the dense
// taxonomy tables and the per-taxon run state.  Compiled with hipcc together with
// ku_kernels.hip into libkrakenuniq_amd.so.  No CPU classification path exists
// here: every compute entry point needs a usable gfx950 device.
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <time.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <mutex>
#include <thread>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <unordered_map>
#include <vector>

#include "ku_host.h"
#include "ku_internal.h"

// ---------------------------------------------------------------------------- errors
static thread_local std::string g_last_error;
void ku_set_error(const std::string &s) { g_last_error = s; }
static int fail(int code, const std::string &msg) {
  g_last_error = msg;
  return code;
}
#define HIP_TRY(expr)                                                                          \
  do {                                                                                         \
    hipError_t e_ = (expr);                                                                    \
    if (e_ != hipSuccess)                                                                      \
      return fail(e_ == hipErrorOutOfMemory ? KU_ENOMEM : KU_EHIP,                             \
                  std::string(#expr) + ": " + hipGetErrorString(e_));                          \
  } while (0)
#define KU_TRY(expr)            \
  do {                          \
    int s_ = (expr);            \
    if (s_ != KU_OK) return s_; \
  } while (0)

extern "C" const char *ku_strerror(int status) {
  switch (status) {
    case KU_OK: return "ok";
    case KU_EINVAL: return "invalid argument";
    case KU_EDATA: return "malformed database / index / taxonomy data";
    case KU_ENOINPUT: return "cannot open input";
    case KU_ENOMEM: return "out of memory";
    case KU_EHIP: return "HIP runtime error or no usable gfx950 device";
    case KU_ESTATE: return "call order violated";
    case KU_EUNSUP: return "not supported by this build";
    default: return "unknown status";
  }
}
extern "C" const char *ku_last_error(void) { return g_last_error.c_str(); }
extern "C" int ku_abi_version(void) { return KU_ABI_VERSION; }
extern "C" int ku_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

extern "C" int ku_host_alloc(size_t bytes, void **out) {
  if (!out) return fail(KU_EINVAL, "ku_host_alloc: null argument");
  *out = nullptr;
  hipError_t e = hipHostMalloc(out, bytes ? bytes : 1, hipHostMallocDefault);
  if (e != hipSuccess) return fail(e == hipErrorOutOfMemory ? KU_ENOMEM : KU_EHIP, std::string("hipHostMalloc: ") + hipGetErrorString(e));
  return KU_OK;
}
extern "C" void ku_host_free(void *p) {
  if (p) (void)hipHostFree(p);
}

// ---------------------------------------------------------------------------- ku_db
struct ku_db {
  const uint8_t *pairs = nullptr;
  const uint64_t *offsets = nullptr;
  ku_db_info info{};
  void *map_kdb = nullptr, *map_idx = nullptr;
  size_t map_kdb_sz = 0, map_idx_sz = 0;
  // ku_db_values: the distinct values, scanned once (callers ask for the count first and the list second)
  mutable std::mutex values_mu;
  mutable std::vector<uint32_t> values;
  mutable bool values_ready = false;
};

static void *map_file(const char *path, size_t *sz) {
  int fd = open(path, O_RDONLY);
  if (fd < 0) return nullptr;
  struct stat st;
  if (fstat(fd, &st) != 0) { close(fd); return nullptr; }
  void *p = st.st_size ? mmap(nullptr, (size_t)st.st_size, PROT_READ, MAP_PRIVATE, fd, 0) : MAP_FAILED;
  close(fd);
  if (p == MAP_FAILED) return nullptr;
  *sz = (size_t)st.st_size;
  return p;
}

static int db_validate(ku_db *db) {
  const ku_db_info &i = db->info;
  if (i.k < 1 || i.k > 31) return fail(KU_EDATA, "k must be in [1,31]");
  // KrakenDB::bin_key computes its mask in 32-bit int (krakendb.cpp:204): nt <= 15
  if (i.nt < 1 || i.nt > 15 || i.nt > i.k) return fail(KU_EDATA, "minimizer length must be in [1,min(15,k)]");
  if (i.idx_type != 1 && i.idx_type != 2) return fail(KU_EDATA, "illegal Kraken DB index format");
  return KU_OK;
}

extern "C" int ku_db_open(const char *kdb_path, const char *idx_path, ku_db **out) {
  if (!kdb_path || !idx_path || !out) return fail(KU_EINVAL, "ku_db_open: null argument");
  *out = nullptr;
  ku_db *db = new ku_db();
  db->map_kdb = map_file(kdb_path, &db->map_kdb_sz);
  if (!db->map_kdb) { delete db; return fail(KU_ENOINPUT, std::string("can't open ") + kdb_path); }
  db->map_idx = map_file(idx_path, &db->map_idx_sz);
  if (!db->map_idx) { ku_db_close(db); return fail(KU_ENOINPUT, std::string("can't open ") + idx_path); }
  const uint8_t *kp = (const uint8_t *)db->map_kdb, *ip = (const uint8_t *)db->map_idx;
  // krakendb.cpp:67-77
  if (db->map_kdb_sz < 72 || memcmp(kp, "JFLISTDN", 8) != 0) {
    ku_db_close(db);
    return fail(KU_EDATA, "database in improper format");
  }
  uint64_t key_bits, val_len, key_ct;
  memcpy(&key_bits, kp + 8, 8);
  memcpy(&val_len, kp + 16, 8);
  memcpy(&key_ct, kp + 48, 8);
  if (val_len != 4) { ku_db_close(db); return fail(KU_EDATA, "can only handle 4 byte DB values"); }
  if (key_bits == 0 || key_bits > 62 || (key_bits & 1)) { ku_db_close(db); return fail(KU_EDATA, "unsupported key_bits"); }
  size_t hdr = 72 + 2 * (4 + 8 * key_bits);  // krakendb.cpp:177
  db->info.k = (uint32_t)(key_bits / 2);
  db->info.key_len = (uint32_t)(key_bits / 8 + !!(key_bits % 8));
  db->info.key_ct = key_ct;
  // key_ct comes from the file: a crafted count must not wrap the product and pass the size test
  if (db->map_kdb_sz < hdr || key_ct > (db->map_kdb_sz - hdr) / (db->info.key_len + 4)) {
    ku_db_close(db);
    return fail(KU_EDATA, "database file truncated");
  }
  db->pairs = kp + hdr;
  // krakendb.cpp:534-544
  if (db->map_idx_sz < 8) { ku_db_close(db); return fail(KU_EDATA, "illegal Kraken DB index format"); }
  if (memcmp(ip, "KRAKIDX", 7) == 0) db->info.idx_type = 1;
  else if (memcmp(ip, "KRAKIX2", 7) == 0) db->info.idx_type = 2;
  else { ku_db_close(db); return fail(KU_EDATA, "illegal Kraken DB index format"); }
  db->info.nt = ip[7];
  int st = db_validate(db);
  if (st != KU_OK) { ku_db_close(db); return st; }
  db->info.n_bins = 1ull << (2 * db->info.nt);
  if (db->map_idx_sz < 8 + 8 * (db->info.n_bins + 1)) { ku_db_close(db); return fail(KU_EDATA, "index file truncated"); }
  db->offsets = (const uint64_t *)(ip + 8);
  if (db->offsets[db->info.n_bins] != key_ct) { ku_db_close(db); return fail(KU_EDATA, "index does not match database (last offset != key_ct)"); }
  *out = db;
  return KU_OK;
}

extern "C" int ku_db_wrap(const void *pairs, uint64_t key_ct, uint32_t k, const uint64_t *offsets, uint32_t nt,
                          uint32_t idx_type, ku_db **out) {
  if (!out || (!pairs && key_ct) || !offsets) return fail(KU_EINVAL, "ku_db_wrap: null argument");
  ku_db *db = new ku_db();
  db->pairs = (const uint8_t *)pairs;
  db->offsets = offsets;
  db->info.k = k; db->info.nt = nt; db->info.idx_type = idx_type;
  db->info.key_len = (2 * k + 7) / 8;
  db->info.key_ct = key_ct;
  int st = db_validate(db);
  if (st != KU_OK) { delete db; return st; }
  db->info.n_bins = 1ull << (2 * nt);
  *out = db;
  return KU_OK;
}

extern "C" void ku_db_close(ku_db *db) {
  if (!db) return;
  if (db->map_kdb) munmap(db->map_kdb, db->map_kdb_sz);
  if (db->map_idx) munmap(db->map_idx, db->map_idx_sz);
  delete db;
}

extern "C" int ku_db_get_info(const ku_db *db, ku_db_info *out) {
  if (!db || !out) return fail(KU_EINVAL, "ku_db_get_info: null argument");
  *out = db->info;
  return KU_OK;
}

extern "C" int ku_db_shard_plan(const ku_db *db, uint32_t n_shards, uint64_t *bounds) {
  if (!db || !bounds || n_shards == 0) return fail(KU_EINVAL, "ku_db_shard_plan: bad argument");
  const uint64_t nb = db->info.n_bins, ps = db->info.key_len + 4;
  auto cost = [&](uint64_t b) { return 8 * b + ps * db->offsets[b]; };  // bytes of bins [0, b)
  const uint64_t total = cost(nb);
  bounds[0] = 0;
  for (uint32_t s = 1; s < n_shards; ++s) {
    // smallest b with cost(b) >= total * s / n_shards (monotone in b)
    unsigned __int128 target = (unsigned __int128)total * s / n_shards;
    uint64_t lo = bounds[s - 1], hi = nb;
    while (lo < hi) {
      uint64_t mid = lo + (hi - lo) / 2;
      if ((unsigned __int128)cost(mid) < target) lo = mid + 1; else hi = mid;
    }
    bounds[s] = lo;
  }
  bounds[n_shards] = nb;
  return KU_OK;
}

extern "C" int ku_db_chunk_plan(const ku_db *db, uint64_t max_bytes, uint64_t *bounds, uint32_t cap,
                                uint32_t *n_chunks) {
  if (!db || !bounds || !n_chunks) return fail(KU_EINVAL, "ku_db_chunk_plan: null argument");
  const uint64_t nb = db->info.n_bins, ps = db->info.key_len + 4;
  uint64_t idx_pos = 0;
  uint32_t n = 0;
  bounds[0] = 0;
  uint64_t last_dbx = 0;
  while (idx_pos < nb) {
    // KrakenDB::upper_bound (krakendb.cpp:430-461): first bin that no longer fits
    uint64_t first = idx_pos, count = nb - idx_pos;
    const uint64_t orig = idx_pos, data0 = db->offsets[orig];
    while (count > 0) {
      uint64_t step = count / 2, it = first + step;
      uint64_t size_index = (it + 1 - orig) * 8;
      uint64_t size_data = (db->offsets[it + 1] - data0) * ps;
      if (size_index + size_data + 8 <= max_bytes) { first = it + 1; count -= step + 1; }
      else count = step;
    }
    if (first == idx_pos) return fail(KU_EINVAL, "preload size too small for the largest minimizer bin");
    idx_pos = first;
    uint64_t dbx = db->offsets[idx_pos];
    if (dbx == last_dbx) continue;  // chunk without k-mers is skipped (krakendb.cpp:497-498)
    last_dbx = dbx;
    ++n;
    if (n <= cap) bounds[n] = idx_pos;
  }
  *n_chunks = n;
  return n <= cap ? KU_OK : fail(KU_EINVAL, "ku_db_chunk_plan: bounds array too small");
}

int ku_db_raw(const ku_db *db, const uint8_t **pairs, const uint64_t **offsets) {
  if (!db || !pairs || !offsets) return KU_EINVAL;
  *pairs = db->pairs;
  *offsets = db->offsets;
  return KU_OK;
}

extern "C" int ku_db_values(const ku_db *db, uint32_t *out, uint64_t *n) {
  if (!db || !n) return fail(KU_EINVAL, "ku_db_values: null argument");
  std::lock_guard<std::mutex> lk(db->values_mu);
  if (!db->values_ready) {
    // one bit per value below 2^26 (taxon ids in practice), set by a team of scanning threads (the 4-byte value sits
    // behind every key); the rare larger ones go to a per-thread list
    const uint64_t np = db->info.key_ct, ps = db->info.key_len + 4, kl = db->info.key_len;
    constexpr uint32_t SMALL = 1u << 26;
    std::vector<std::atomic<uint64_t>> bits(SMALL / 64);
    for (auto &w : bits) w.store(0, std::memory_order_relaxed);
    unsigned nthr = std::max(1u, std::min(16u, std::thread::hardware_concurrency()));
    if (np < (1u << 20)) nthr = 1;
    std::vector<std::vector<uint32_t>> large(nthr);
    std::vector<std::thread> team;
    for (unsigned t = 0; t < nthr; ++t)
      team.emplace_back([&, t] {
        const uint64_t lo = np * t / nthr, hi = np * (t + 1) / nthr;
        uint32_t last = 0;
        for (uint64_t i = lo; i < hi; ++i) {
          uint32_t v;
          memcpy(&v, db->pairs + i * ps + kl, 4);
          if (v == last) continue;  // values come in long runs inside a bin
          last = v;
          if (v >= SMALL) {
            large[t].push_back(v);
            if (large[t].size() >= (1u << 20)) {
              std::sort(large[t].begin(), large[t].end());
              large[t].erase(std::unique(large[t].begin(), large[t].end()), large[t].end());
            }
            continue;
          }
          std::atomic<uint64_t> &w = bits[v >> 6];
          const uint64_t m = 1ull << (v & 63);
          if (!(w.load(std::memory_order_relaxed) & m)) w.fetch_or(m, std::memory_order_relaxed);
        }
      });
    for (auto &th : team) th.join();
    std::vector<uint32_t> &vals = db->values;
    for (uint64_t wi = 0; wi < bits.size(); ++wi) {
      uint64_t w = bits[wi].load(std::memory_order_relaxed);
      if (wi == 0) w &= ~1ull;  // value 0 is "no taxon", never a slot
      while (w) {
        vals.push_back((uint32_t)(wi * 64 + (uint64_t)__builtin_ctzll(w)));
        w &= w - 1;
      }
    }
    const size_t n_small = vals.size();
    for (auto &l : large) vals.insert(vals.end(), l.begin(), l.end());
    std::sort(vals.begin() + n_small, vals.end());
    vals.erase(std::unique(vals.begin() + n_small, vals.end()), vals.end());
    db->values_ready = true;
  }
  const uint64_t count = db->values.size();
  if (out) {
    if (count > *n) return fail(KU_EINVAL, "ku_db_values: output array too small");
    memcpy(out, db->values.data(), count * 4);
  }
  *n = count;
  return KU_OK;
}

// ---------------------------------------------------------------------------- ku_tax
extern "C" int ku_tax_open(const char *path, ku_tax **out) {
  if (!path || !out) return fail(KU_EINVAL, "ku_tax_open: null argument");
  *out = nullptr;
  FILE *f = fopen(path, "r");
  if (!f) return fail(KU_ENOINPUT, std::string("unable to open taxonomy index file ") + path);
  ku_tax *t = new ku_tax();
  // taxdb.hpp:581-597: "id <ws> parent <tab> name <tab> rank-to-end-of-line"
  char *line = nullptr;
  size_t lcap = 0;
  ssize_t ll;
  while ((ll = getline(&line, &lcap, f)) > 0) {
    if (line[ll - 1] == '\n') line[--ll] = 0;
    if (ll == 0) continue;
    char *p = line, *end;
    unsigned long id = strtoul(p, &end, 10);
    if (end == p) continue;
    p = end;
    unsigned long par = strtoul(p, &end, 10);
    if (end == p) continue;
    p = end;
    if (*p) ++p;
    char *tab = strchr(p, '\t');
    std::string name, rank;
    if (tab) { name.assign(p, tab - p); rank.assign(tab + 1); } else name.assign(p);
    if ((uint32_t)id > 1 && id == par) {  // taxdb.hpp:583-586: fatal in the reference
      free(line); fclose(f); delete t;
      return fail(KU_EDATA, "taxDB: the parent of " + std::to_string(id) + " is itself");
    }
    t->add((uint32_t)id, (uint32_t)par, name, rank);
  }
  free(line);
  fclose(f);
  t->add(0, 0, "unclassified", "no rank");  // taxdb.hpp:599
  t->finish();
  *out = t;
  return KU_OK;
}

extern "C" int ku_tax_from_arrays(const uint32_t *ids, const uint32_t *parents, uint64_t n, ku_tax **out) {
  if (!out || (n && (!ids || !parents))) return fail(KU_EINVAL, "ku_tax_from_arrays: null argument");
  ku_tax *t = new ku_tax();
  for (uint64_t i = 0; i < n; ++i) t->add(ids[i], parents[i], "", "");
  t->add(0, 0, "unclassified", "no rank");
  t->finish();
  *out = t;
  return KU_OK;
}
extern "C" void ku_tax_close(ku_tax *t) { delete t; }
extern "C" uint64_t ku_tax_size(const ku_tax *t) { return t ? t->ids.size() : 0; }
extern "C" int ku_tax_ids(const ku_tax *t, uint32_t *ids) {
  if (!t || (!ids && !t->ids.empty())) return fail(KU_EINVAL, "ku_tax_ids: null argument");
  if (!t->ids.empty()) memcpy(ids, t->ids.data(), t->ids.size() * 4);
  return KU_OK;
}
extern "C" uint32_t ku_tax_parent(const ku_tax *t, uint32_t taxid) {
  if (!t || taxid == 0) return KU_AMBIG;  // getParentMap skips key 0 (taxdb.hpp:388-389)
  auto it = t->row.find(taxid);
  return it == t->row.end() ? KU_AMBIG : t->parent_map[it->second];
}

// ---------------------------------------------------------------------------- ku_ctx
struct DevBuf {
  void *p = nullptr;
  size_t cap = 0;
  int reserve(size_t bytes) {
    if (bytes <= cap) return KU_OK;
    if (p) (void)hipFree(p);
    p = nullptr; cap = 0;
    size_t want = bytes + bytes / 4 + 256;
    if (hipMalloc(&p, want) != hipSuccess) {
      if (hipMalloc(&p, bytes) != hipSuccess) { p = nullptr; return KU_ENOMEM; }
      want = bytes;
    }
    cap = want;
    return KU_OK;
  }
  void release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
};

// page-locked host scratch that grows on demand (sources and targets of asynchronous copies must outlive the call that
// enqueues them and must be page-locked for the copy to be asynchronous at all)
struct PinBuf {
  void *p = nullptr;
  size_t cap = 0;
  int reserve(size_t bytes) {
    if (bytes <= cap) return KU_OK;
    if (p) (void)hipHostFree(p);
    p = nullptr; cap = 0;
    const size_t want = bytes + bytes / 4 + 256;
    if (hipHostMalloc(&p, want, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); p = nullptr; return KU_ENOMEM; }
    cap = want;
    return KU_OK;
  }
  void release() { if (p) (void)hipHostFree(p); p = nullptr; cap = 0; }
};

// One batch on its way through the fused kernel with run-length encoded output (ku_classify_batch_rle_enqueue / _finish,
// round 5): its own device buffers and page-locked scratch -- KU_RLE_MAX_IN_FLIGHT of them take turns, so that the uploads of
// the next batches and the copies back of the previous ones run under the kernel of batch b and the host waits for ONE event
// per batch -- and what
// _finish needs to know about the batch.
struct RleJob {
  DevBuf seqs, off, len, calls, runs, roff, rcnt, ws, unit, u_cnt, u_flag;
  PinBuf pin;       // [0] extent of the run array, [1] entries of the run-wide set, [2] the emulation's error word; byte 64 on: unit flags
  PinBuf pin_unit;  // work-unit number of every read (source of an asynchronous upload)
  unsigned long long *d_counter = nullptr;  // the kernel's bump counter of the run array (2 dwords of the context's scalars)
  hipEvent_t kernels_done = nullptr, done = nullptr;
  hipEvent_t t_k0 = nullptr, t_k1 = nullptr;  // KU_RLE_TIMES: around the batch's kernels on the stream they run on
  std::vector<hipEvent_t> seg_events;
  bool busy = false;
  bool settled = false;         // classified by a one-step path inside _enqueue: _finish only hands the totals over
  bool runs_in_ctx = false;     // the runs lie in the context's own run buffer (one-step paths, the overflow redo)
  uint64_t runs_copied = 0;     // entries of the run array already copied to the caller's buffer (0: ku_fetch_runs does it)
  uint64_t n_runs = 0;
  // the batch
  uint64_t n_bytes = 0, n_reads = 0, runs_cap = 0;
  uint32_t max_n = 0;
  ku_opts o{};
  const uint32_t *h_len = nullptr;
  uint32_t *h_calls = nullptr, *h_hits = nullptr, *h_rcnt = nullptr;
  uint64_t *h_roff = nullptr;
  // sparse-sketch emulation, fast path: the batch's work units
  bool sparse = false;
  bool cont_carry = false;      // unit 0 continues a unit whose state sits in the carry buffers (L / U entries; the staged form)
  bool cont_tail = false;       // unit 0 continues a unit kept as its reads + insert counts (the fast path's own form)
  bool open_after = false;      // the last unit is still open behind this batch
  uint32_t n_units = 0;
  uint64_t kmers = 0;           // upper bound of what the kernel may add to the run-wide set
  uint64_t acc_after = 0;
  std::vector<uint64_t> unit_first_read;
  std::vector<char> tail_text;      // cont_tail: the reads of the open unit BEFORE this batch (bases, each read followed by '\n')
  std::vector<uint32_t> tail_len;
  void release() {
    for (DevBuf *b : {&seqs, &off, &len, &calls, &runs, &roff, &rcnt, &ws, &unit, &u_cnt, &u_flag}) b->release();
    pin.release();
    pin_unit.release();
    if (kernels_done) (void)hipEventDestroy(kernels_done);
    if (done) (void)hipEventDestroy(done);
    if (t_k0) (void)hipEventDestroy(t_k0);
    if (t_k1) (void)hipEventDestroy(t_k1);
    for (hipEvent_t e : seg_events) (void)hipEventDestroy(e);
    kernels_done = done = t_k0 = t_k1 = nullptr;
    seg_events.clear();
  }
};

// one resident database (shard): the 12-byte pairs until the taxonomy is set, the probe table afterwards
struct DbStore {
  bool db_owned = false, offsets_owned = false;
  bool hash_layout = true;
  bool seen_dirty = false;  // SEEN marks of the probe table may be set (ku_device.h; the sparse-sketch emulation's fast path)
  void *d_table = nullptr;
  uint64_t n_dup = 0;
  uint64_t table_lines = 0;
  uint32_t *d_pairs = nullptr;
  uint64_t *d_offsets = nullptr;
  KuDbDev db{};
  std::vector<uint32_t> values;  // ascending distinct non-zero raw taxids of the shard
};

struct ku_ctx {
  int device = 0;
  int n_cu = 256;
  hipStream_t stream = nullptr;
  bool db_loaded = false, tax_set = false;
  bool hash_layout = true;   // KU_LAYOUT=sorted keeps the on-disk order + binary search (A/B and fallback for HBM-tight shards)
  double load_factor = 0.2;  // keys per bucket slot (8 slots per 128-byte line); KU_LOAD_FACTOR fixes it
  bool load_factor_set = false;
  DbStore m;                   // the (first) database: the only one that may be a strict minimizer-range shard
  std::vector<DbStore> extra;  // further whole databases of a hierarchical run, searched in order after `m`
  // taxonomy tables
  std::vector<uint32_t> h_node_taxid, h_slot_taxid;
  uint32_t *d_node_parent = nullptr, *d_node_slot = nullptr, *d_node_taxid = nullptr, *d_slot_node = nullptr,
           *d_slot_taxid = nullptr, *d_slot_anc_off = nullptr, *d_slot_anc = nullptr;
  KuTaxDev tax{};
  // run state
  KuCountsDev cnt{};
  // scratch for the host-buffer entry point
  DevBuf b_seqs, b_off, b_len, b_calls, b_taxa, b_hits, b_ws, b_runs, b_roff, b_rcnt;
  // ku_classify_batch_rle through the fused kernel: the batch goes up in segments on a stream of its own while the
  // segments before are classified (one event per segment)
  hipStream_t h2d_stream = nullptr, d2h_stream = nullptr, fetch_stream = nullptr;
  // Round 6: the kernels of consecutive batches in flight run on TWO streams in turn, so that the tail of one batch's launch --
  // its last waves, their counter flushes -- lies under the start of the next one's: launches of ~120 k reads then cost what the
  // bench's 10 M-read launch costs per read (scripts/launch_shape_probe.py: 30.3 -> 19.5 ms per 10 M reads; 19.8 in one launch).
  // What orders the batches: main_ev (work queued on the context's own stream before the batch), tail_ready (the open work unit's
  // insert counts travel from batch to batch), and the host, which waits for a batch's event before it settles it.
  hipStream_t k_streams[2] = {nullptr, nullptr};
  hipEvent_t main_ev = nullptr, tail_ready = nullptr;
  bool tail_ready_set = false;
  std::vector<hipEvent_t> seg_events;
  uint32_t *d_scalar = nullptr;
  // ku_classify_batch_rle in two steps: up to two batches in flight (FIFO: rle_head is the oldest)
  RleJob rle[KU_RLE_MAX_IN_FLIGHT];
  int rle_head = 0, rle_in_flight = 0;
  const void *fetch_runs_src = nullptr;  // where the runs of the batch finished last lie (ku_fetch_runs)
  const void *last_calls_dev = nullptr;  // ... and its calls on the device (ku_ctx_replace_calls)
  uint64_t last_runs_copied = 0;         // ... and how many of its runs are in the caller's buffer already
  // ku_ctx_count_taxons of the store it was computed for (identified by its buffers)
  std::vector<unsigned long long> count_cache;
  const void *count_cache_store = nullptr, *count_cache_pairs = nullptr;
  uint64_t count_cache_lines = 0;
  uint64_t n_runs = 0;  // runs of the last ku_classify_batch_rle, still in b_runs
  // exact distinct counting (classifyExact): one global set of canonical k-mers + first-insertion counters per slot
  unsigned long long *d_exact_set = nullptr, *d_exact_unique = nullptr;
  uint64_t exact_mask = 0;
  // out-of-core runs: the NEXT chunk, uploaded and laid out on its own stream while the resident one is searched
  struct Prefetch {
    bool valid = false;
    const ku_db *db = nullptr;
    uint64_t bin_lo = 0, bin_hi = 0;
    DbStore store;
    hipStream_t stream = nullptr;
    uint32_t *d_scalar = nullptr;
  } pf;
  // HyperLogLog++ sparse-mode emulation (ku_sparse.hip)
  struct Sparse {
    bool on = false;
    uint64_t unit_nt = 500000;  // Work_unit_size (classify.cpp:38); 0 = the whole run is one unit (-x mode)
    uint64_t acc_nt = 0;        // nt of the unit that is still open
    bool open = false;          // ... whose encodings and statistics sit in the carry buffers
    KuSparseDev dev{};
    unsigned long long *d_counters = nullptr;  // [0] size of G, [1..2] carry sizes, [3] export size
    DevBuf unit, carry_l, carry_u, out;
    DevBuf u_cnt, u_flag, list;  // fast path: inserts per (unit, slot), per-unit flags, the reads of the flagged units
    uint64_t n_carry_l = 0, n_carry_u = 0, cap_carry_l = 0, cap_carry_u = 0;
    uint64_t g_count = 0;       // entries of the global set after the last pass (host copy of d_counters[0])
    bool gave_up = false;       // the emulation ran out of device memory during the run and was switched off
    // The open unit in TAIL form (round 5; the fast path's own): a host copy of its reads so far and its insert counts per
    // slot.  A unit can only turn a sketch dense when it gave it >= 1025 inserts (hyperloglogplus.cpp:496-498) -- known from
    // the counts once the unit closes, whichever batches it straddled; only then, and only for such a unit, does the exact
    // evaluation (L / U tables) run, over these reads + the closing batch's.  Rounds 3-4 ran it for the first and the last
    // unit of EVERY batch to carry their L / U entries along: two passes, ten launches, two host round trips per batch.
    // (acc_nt, open, tail_open and the tail describe the state behind the newest ENQUEUED batch.)
    bool tail_open = false;
    std::vector<char> tail_text;   // bases of the unit's reads, each read followed by '\n'
    std::vector<uint32_t> tail_len;
    DevBuf tail_row;               // inserts of the open unit so far, per slot
    DevBuf t_seqs, t_off, t_len, t_taxa, t_unit;  // the tail on the device, when it is evaluated
  } sp;
};

hipStream_t ku_ctx_stream_of(ku_ctx *ctx) { return ctx->stream; }
unsigned long long *ku_ctx_exact_unique_of(ku_ctx *ctx) { return ctx->d_exact_unique; }
int ku_ctx_device_of(const ku_ctx *ctx) { return ctx->device; }
int ku_ctx_cus_of(const ku_ctx *ctx) { return ctx->n_cu; }
uint32_t ku_ctx_k_of(const ku_ctx *ctx) { return ctx->m.db.k; }

static int ctx_activate(ku_ctx *ctx) {
  if (hipSetDevice(ctx->device) != hipSuccess) return fail(KU_EHIP, "hipSetDevice failed");
  return KU_OK;
}

extern "C" int ku_ctx_create(int device, ku_ctx **out) {
  if (!out) return fail(KU_EINVAL, "ku_ctx_create: null argument");
  *out = nullptr;
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return fail(KU_EHIP, "no HIP device available (this library has no CPU fallback)");
  if (device < 0 || device >= n) return fail(KU_EINVAL, "device index out of range");
  hipDeviceProp_t prop;
  HIP_TRY(hipGetDeviceProperties(&prop, device));
  if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
    return fail(KU_EHIP, std::string("device is ") + prop.gcnArchName + ", this library is built for gfx950 only");
  ku_ctx *ctx = new ku_ctx();
  ctx->device = device;
  ctx->n_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  HIP_TRY(hipSetDevice(device));
  HIP_TRY(hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking));
  HIP_TRY(hipMalloc((void **)&ctx->d_scalar, 128));
  if (const char *e = getenv("KU_LAYOUT")) ctx->hash_layout = strcmp(e, "sorted") != 0;
  if (const char *e = getenv("KU_LOAD_FACTOR")) {
    double f = atof(e);
    if (f >= 0.05 && f <= 0.9) { ctx->load_factor = f; ctx->load_factor_set = true; }
  }
  *out = ctx;
  return KU_OK;
}

static void store_free(DbStore &d) {  // (callers that free a context's store also drop its count_taxons cache: ctx_drop_count_cache)
  if (d.d_table) (void)hipFree(d.d_table);
  if (d.db_owned && d.d_pairs) (void)hipFree(d.d_pairs);
  if (d.offsets_owned && d.d_offsets) (void)hipFree(d.d_offsets);
  d = DbStore{};
}
static void ctx_drop_count_cache(ku_ctx *ctx) {
  ctx->count_cache.clear();
  ctx->count_cache_store = ctx->count_cache_pairs = nullptr;
  ctx->count_cache_lines = 0;
}
static void ctx_free_db(ku_ctx *ctx) {
  ctx_drop_count_cache(ctx);
  if (ctx->pf.valid) store_free(ctx->pf.store);
  ctx->pf.valid = false;
  store_free(ctx->m);
  for (DbStore &e : ctx->extra) store_free(e);
  ctx->extra.clear();
  ctx->db_loaded = false;
}
static void ctx_free_sparse(ku_ctx *ctx) {
  KuSparseDev &d = ctx->sp.dev;
  for (void *p : {(void *)d.l_key, (void *)d.l_first, (void *)d.u_key, (void *)d.u_distinct, (void *)d.u_last, (void *)d.u_maxfirst,
                  (void *)d.g_key, (void *)d.dense, (void *)d.err, (void *)ctx->sp.d_counters})
    if (p) (void)hipFree(p);
  for (DevBuf *b : {&ctx->sp.unit, &ctx->sp.carry_l, &ctx->sp.carry_u, &ctx->sp.out, &ctx->sp.u_cnt, &ctx->sp.u_flag, &ctx->sp.list, &ctx->sp.tail_row,
                    &ctx->sp.t_seqs, &ctx->sp.t_off, &ctx->sp.t_len, &ctx->sp.t_taxa, &ctx->sp.t_unit})
    b->release();
  ctx->sp = ku_ctx::Sparse{};
}
static void ctx_free_tax(ku_ctx *ctx) {
  for (uint32_t **p : {&ctx->d_node_parent, &ctx->d_node_slot, &ctx->d_node_taxid, &ctx->d_slot_node, &ctx->d_slot_taxid,
                       &ctx->d_slot_anc_off, &ctx->d_slot_anc}) {
    if (*p) (void)hipFree(*p);
    *p = nullptr;
  }
  if (ctx->cnt.registers) (void)hipFree(ctx->cnt.registers);
  if (ctx->cnt.n_kmers) (void)hipFree(ctx->cnt.n_kmers);
  if (ctx->cnt.n_reads) (void)hipFree(ctx->cnt.n_reads);
  if (ctx->d_exact_set) (void)hipFree(ctx->d_exact_set);
  if (ctx->d_exact_unique) (void)hipFree(ctx->d_exact_unique);
  ctx->d_exact_set = ctx->d_exact_unique = nullptr;
  ctx->exact_mask = 0;
  ctx_free_sparse(ctx);
  ctx->cnt = KuCountsDev{};
  ctx->tax_set = false;
}

static void rle_times_print();
extern "C" void ku_ctx_destroy(ku_ctx *ctx) {
  if (!ctx) return;
  rle_times_print();
  (void)hipSetDevice(ctx->device);
  if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
  ctx_free_db(ctx);
  ctx_free_tax(ctx);
  for (DevBuf *b : {&ctx->b_seqs, &ctx->b_off, &ctx->b_len, &ctx->b_calls, &ctx->b_taxa, &ctx->b_hits, &ctx->b_ws, &ctx->b_runs,
                    &ctx->b_roff, &ctx->b_rcnt})
    b->release();
  if (ctx->d_scalar) (void)hipFree(ctx->d_scalar);
  if (ctx->pf.d_scalar) (void)hipFree(ctx->pf.d_scalar);
  if (ctx->pf.stream) (void)hipStreamDestroy(ctx->pf.stream);
  for (hipEvent_t e : ctx->seg_events) (void)hipEventDestroy(e);
  for (RleJob &j : ctx->rle) j.release();
  if (ctx->h2d_stream) (void)hipStreamDestroy(ctx->h2d_stream);
  if (ctx->d2h_stream) (void)hipStreamDestroy(ctx->d2h_stream);
  for (hipStream_t &ks : ctx->k_streams) if (ks) { (void)hipStreamDestroy(ks); ks = nullptr; }
  if (ctx->main_ev) (void)hipEventDestroy(ctx->main_ev);
  if (ctx->tail_ready) (void)hipEventDestroy(ctx->tail_ready);
  if (ctx->fetch_stream) (void)hipStreamDestroy(ctx->fetch_stream);
  if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
  delete ctx;
}

// distinct values of a resident shard via a 2^32-bit bitmap (512 MiB scratch)
static int store_scan_values(ku_ctx *ctx, DbStore &d) {
  d.values.clear();
  uint32_t *d_bitmap = nullptr, *d_list = nullptr, *d_count = ctx->d_scalar;
  const uint32_t cap = 1u << 24;
  HIP_TRY(hipMalloc((void **)&d_bitmap, 1ull << 29));
  if (hipMalloc((void **)&d_list, (size_t)cap * 4) != hipSuccess) { (void)hipFree(d_bitmap); return fail(KU_ENOMEM, "hipMalloc values list"); }
  int st = KU_OK;
  uint32_t count = 0;
  if (hipMemsetAsync(d_bitmap, 0, 1ull << 29, ctx->stream) != hipSuccess ||
      hipMemsetAsync(d_count, 0, 4, ctx->stream) != hipSuccess)
    st = fail(KU_EHIP, "memset failed");
  if (st == KU_OK) st = ku_launch_mark_values(d.d_pairs, d.db.n_pairs, d_bitmap, ctx->stream);
  if (st == KU_OK) st = ku_launch_collect_values(d_bitmap, d_list, cap, d_count, ctx->stream);
  if (st == KU_OK && hipMemcpyAsync(&count, d_count, 4, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess) st = fail(KU_EHIP, "memcpy failed");
  if (st == KU_OK && hipStreamSynchronize(ctx->stream) != hipSuccess) st = fail(KU_EHIP, "value scan kernel failed");
  if (st == KU_OK && count > cap) st = fail(KU_EUNSUP, "more than 2^24 distinct taxids in the database");
  if (st == KU_OK) {
    d.values.resize(count);
    if (count && hipMemcpy(d.values.data(), d_list, (size_t)count * 4, hipMemcpyDeviceToHost) != hipSuccess) st = fail(KU_EHIP, "memcpy failed");
    std::sort(d.values.begin(), d.values.end());
    if (!d.values.empty() && d.values[0] == 0) d.values.erase(d.values.begin());
  }
  (void)hipFree(d_bitmap);
  (void)hipFree(d_list);
  return st;
}

static void fill_db_dev(DbStore &d, uint64_t n_pairs, uint64_t pair_base, uint32_t k, uint32_t nt, uint32_t idx_type,
                        uint64_t bin_lo, uint64_t bin_hi) {
  d.db.pairs = d.d_pairs;
  d.db.table = nullptr;
  d.db.n_lines = 0;
  d.db.offsets = d.d_offsets;
  d.db.pair_base = pair_base;
  d.db.n_pairs = n_pairs;
  d.db.bin_lo = bin_lo;
  d.db.bin_hi = bin_hi;
  d.db.k = k;
  d.db.nt = nt;
  const uint64_t INDEX2_XOR_MASK = 0xe37e28c4271b5a2dULL;  // krakendb.cpp:45
  d.db.xor_mask = idx_type == 1 ? 0u : (uint32_t)(INDEX2_XOR_MASK & ((1ull << (2 * nt)) - 1));
}

// host KrakenDB bins [bin_lo, bin_hi) -> device pairs (12-byte form) + offsets slice
static int store_upload(ku_ctx *ctx, DbStore &d, const ku_db *db, uint64_t bin_lo, uint64_t bin_hi, bool scan_values = true,
                        hipStream_t stream = nullptr) {
  if (!stream) stream = ctx->stream;
  const uint64_t p0 = db->offsets[bin_lo], p1 = db->offsets[bin_hi], np = p1 - p0;
  const uint32_t kl = db->info.key_len, ps = kl + 4;
  HIP_TRY(hipMalloc((void **)&d.d_pairs, std::max<uint64_t>(np, 1) * 12));
  d.db_owned = true;
  if (kl == 8) {
    if (np) HIP_TRY(hipMemcpy(d.d_pairs, db->pairs + p0 * 12, np * 12, hipMemcpyHostToDevice));
  } else if (np) {
    void *d_raw = nullptr;
    HIP_TRY(hipMalloc(&d_raw, np * ps));
    hipError_t e = hipMemcpy(d_raw, db->pairs + p0 * ps, np * ps, hipMemcpyHostToDevice);
    int st = e == hipSuccess ? ku_launch_repack((const uint8_t *)d_raw, np, kl, d.d_pairs, stream) : KU_EHIP;
    if (st == KU_OK && hipStreamSynchronize(stream) != hipSuccess) st = KU_EHIP;
    (void)hipFree(d_raw);
    if (st != KU_OK) return fail(st, "pair repack failed");
  }
  const uint64_t no = bin_hi - bin_lo + 1;
  HIP_TRY(hipMalloc((void **)&d.d_offsets, no * 8));
  d.offsets_owned = true;
  HIP_TRY(hipMemcpy(d.d_offsets, db->offsets + bin_lo, no * 8, hipMemcpyHostToDevice));
  fill_db_dev(d, np, p0, db->info.k, db->info.nt, db->info.idx_type, bin_lo, bin_hi);
  return scan_values ? store_scan_values(ctx, d) : KU_OK;
}

extern "C" int ku_ctx_load_db(ku_ctx *ctx, const ku_db *db, uint64_t bin_lo, uint64_t bin_hi) {
  if (!ctx || !db) return fail(KU_EINVAL, "ku_ctx_load_db: null argument");
  if (bin_lo > bin_hi || bin_hi > db->info.n_bins) return fail(KU_EINVAL, "bin range out of bounds");
  KU_TRY(ctx_activate(ctx));
  ctx_free_tax(ctx);
  ctx_free_db(ctx);
  KU_TRY(store_upload(ctx, ctx->m, db, bin_lo, bin_hi));
  ctx->db_loaded = true;
  return KU_OK;
}

static bool store_whole(const DbStore &d) { return d.db.bin_lo == 0 && d.db.bin_hi == (1ull << (2 * d.db.nt)); }

extern "C" int ku_ctx_add_db(ku_ctx *ctx, const ku_db *db) {
  if (!ctx || !db) return fail(KU_EINVAL, "ku_ctx_add_db: null argument");
  if (!ctx->db_loaded) return fail(KU_ESTATE, "load the first database before adding further ones");
  if (ctx->tax_set) return fail(KU_ESTATE, "add every database before the taxonomy is set");
  if (db->info.k != ctx->m.db.k)  // classify.cpp:199-208 "Different k-mer sizes in databases"
    return fail(KU_EINVAL, "different k-mer sizes in the databases: " + std::to_string(ctx->m.db.k) + " vs " + std::to_string(db->info.k));
  if (!store_whole(ctx->m)) return fail(KU_EUNSUP, "hierarchical multi-database runs need the first database resident as a whole");
  if (ctx->extra.size() >= 7) return fail(KU_EUNSUP, "at most 8 databases");
  KU_TRY(ctx_activate(ctx));
  ctx->extra.emplace_back();
  int st = store_upload(ctx, ctx->extra.back(), db, 0, db->info.n_bins);
  if (st != KU_OK) { store_free(ctx->extra.back()); ctx->extra.pop_back(); }
  return st;
}

extern "C" int ku_ctx_adopt_db(ku_ctx *ctx, void *d_pairs, uint64_t n_pairs, const uint64_t *d_offsets, uint32_t k,
                               uint32_t nt, uint32_t idx_type, uint64_t bin_lo, uint64_t bin_hi) {
  if (!ctx || (!d_pairs && n_pairs) || !d_offsets) return fail(KU_EINVAL, "ku_ctx_adopt_db: null argument");
  if (k < 1 || k > 31 || nt < 1 || nt > 15 || nt > k || (idx_type != 1 && idx_type != 2) || bin_lo > bin_hi ||
      bin_hi > (1ull << (2 * nt)))
    return fail(KU_EINVAL, "ku_ctx_adopt_db: bad geometry");
  KU_TRY(ctx_activate(ctx));
  ctx_free_tax(ctx);
  ctx_free_db(ctx);
  ctx->m.d_pairs = (uint32_t *)d_pairs;
  ctx->m.d_offsets = const_cast<uint64_t *>(d_offsets);
  uint64_t pair_base = 0;
  HIP_TRY(hipMemcpy(&pair_base, d_offsets, 8, hipMemcpyDeviceToHost));
  fill_db_dev(ctx->m, n_pairs, pair_base, k, nt, idx_type, bin_lo, bin_hi);
  ctx->db_loaded = true;
  return store_scan_values(ctx, ctx->m);
}

extern "C" int ku_ctx_db_layout(ku_ctx *ctx, uint32_t *is_hash, uint64_t *resident_bytes) {
  if (!ctx) return fail(KU_EINVAL, "null context");
  if (!ctx->db_loaded) return fail(KU_ESTATE, "no database loaded");
  const bool hash = ctx->m.db.table != nullptr;
  if (is_hash) *is_hash = hash ? 1u : 0u;
  if (resident_bytes)
    *resident_bytes = (hash ? ctx->m.db.n_lines * 128 : ctx->m.db.n_pairs * 12) + (ctx->m.db.bin_hi - ctx->m.db.bin_lo + 1) * 8;
  return KU_OK;
}

// ascending distinct non-zero taxids over every resident database
static std::vector<uint32_t> ctx_all_values(const ku_ctx *ctx) {
  std::vector<uint32_t> v(ctx->m.values);
  for (const DbStore &e : ctx->extra) v.insert(v.end(), e.values.begin(), e.values.end());
  std::sort(v.begin(), v.end());
  v.erase(std::unique(v.begin(), v.end()), v.end());
  return v;
}

extern "C" int ku_ctx_db_values(ku_ctx *ctx, uint32_t *out, uint64_t *n) {
  if (!ctx || !n) return fail(KU_EINVAL, "ku_ctx_db_values: null argument");
  if (!ctx->db_loaded) return fail(KU_ESTATE, "no database loaded");
  const std::vector<uint32_t> v = ctx_all_values(ctx);
  if (out) {
    if (*n < v.size()) return fail(KU_EINVAL, "output array too small");
    if (!v.empty()) memcpy(out, v.data(), v.size() * 4);
  }
  *n = v.size();
  return KU_OK;
}

template <typename T> static int upload(T **dst, const std::vector<T> &src) {
  HIP_TRY(hipMalloc((void **)dst, std::max<size_t>(src.size(), 1) * sizeof(T)));
  if (!src.empty()) HIP_TRY(hipMemcpy(*dst, src.data(), src.size() * sizeof(T), hipMemcpyHostToDevice));
  return KU_OK;
}

// raw taxids -> slot ids in place, then the probe-table layout (needs ctx->tax / d_slot_taxid)
static int store_finalize(ku_ctx *ctx, DbStore &d, hipStream_t stream = nullptr, uint32_t *d_scalar = nullptr) {
  if (!stream) stream = ctx->stream;
  if (!d_scalar) d_scalar = ctx->d_scalar;
  HIP_TRY(hipMemsetAsync(d_scalar, 0, 4, stream));
  KU_TRY(ku_launch_remap_values(d.d_pairs, d.db.n_pairs, ctx->d_slot_taxid, ctx->tax.n_slots, d_scalar, stream));
  uint32_t err = 0;
  HIP_TRY(hipMemcpyAsync(&err, d_scalar, 4, hipMemcpyDeviceToHost, stream));
  HIP_TRY(hipStreamSynchronize(stream));
  if (err) return fail(KU_EDATA, "internal: " + std::to_string(err) + " DB values missing from the slot table");
  d.hash_layout = ctx->hash_layout;
  if (d.hash_layout) {
    // re-lay the shard out as the open-addressing table the lookup kernel probes; the 12-byte
    // pairs are only the build input: owned copies are released, adopted buffers go back to the caller.
    // preferred load factor first (fewest spilled buckets = fewest dependent round trips); denser tables when
    // HBM is short; the sorted on-disk layout (no extra memory) as the last resort
    uint64_t n_lines = 0;
    size_t free_b = 0, total_b = 0;
    (void)hipMemGetInfo(&free_b, &total_b);
    // load factor 0.2 probes fastest (21.2 ms per 10 M reads vs 22.8 at 0.3, 24.2 at 0.4, 26.8 at 0.5) at 72 bytes of
    // HBM per pair: taken when that is at most 40 % of the free memory; then denser tables; KU_LOAD_FACTOR fixes the
    // first choice
    std::vector<double> chain;
    if (ctx->load_factor_set) chain.push_back(ctx->load_factor);
    else if ((double)d.db.n_pairs / 0.2 / 8.0 * 128.0 <= 0.4 * (double)free_b) chain.push_back(0.2);
    for (double lf : {0.3, 0.45, 0.6, 0.8})
      if (chain.empty() || lf > chain.back()) chain.push_back(lf);
    for (double lf : chain) {
      n_lines = (uint64_t)((double)d.db.n_pairs / lf / 8.0) + 1;  // 8 entries per line: load factors up to 0.9 leave free slots
      if (n_lines >= (1ull << 32)) { n_lines = 0; continue; }  // ku_locus_line() reduces to 32 bits
      if (hipMalloc(&d.d_table, n_lines * 128) == hipSuccess) { d.table_lines = n_lines; break; }
      (void)hipGetLastError();
      d.d_table = nullptr;
      n_lines = 0;
    }
    if (!d.d_table) d.hash_layout = false;  // keep the sorted pairs resident and binary-search them
  }
  if (d.hash_layout) {
    const uint64_t n_lines = d.table_lines;
    unsigned long long *d_dup = (unsigned long long *)(d_scalar + 2);
    HIP_TRY(hipMemsetAsync(d_dup, 0, 8, stream));
    KU_TRY(ku_launch_build_table(d.d_pairs, d.db.n_pairs, d.d_table, n_lines, d.db.k, d.db.nt, d.db.xor_mask, d_dup,
                                 stream));
    unsigned long long dup = 0;
    HIP_TRY(hipMemcpyAsync(&dup, d_dup, 8, hipMemcpyDeviceToHost, stream));
    HIP_TRY(hipStreamSynchronize(stream));
    d.n_dup = dup;
    d.db.table = (const uint4 *)d.d_table;
    d.db.n_lines = n_lines;
    if (d.db_owned) (void)hipFree(d.d_pairs);
    d.d_pairs = nullptr;
    d.db_owned = false;
    d.db.pairs = nullptr;
  }
  return KU_OK;
}

extern "C" int ku_ctx_set_taxonomy(ku_ctx *ctx, const ku_tax *tax, const uint32_t *all_values, uint64_t n_values) {
  if (!ctx || !tax) return fail(KU_EINVAL, "ku_ctx_set_taxonomy: null argument");
  if (!ctx->db_loaded) return fail(KU_ESTATE, "load a database before the taxonomy");
  if (ctx->tax_set) return fail(KU_ESTATE, "taxonomy already set for this shard (values are remapped once)");
  KU_TRY(ctx_activate(ctx));
  // slot table: 0 + ascending distinct DB values (over all shards when given)
  std::vector<uint32_t> slots;
  slots.push_back(0);
  if (all_values) {
    for (uint64_t i = 0; i < n_values; ++i) if (all_values[i]) slots.push_back(all_values[i]);
    std::sort(slots.begin() + 1, slots.end());
    slots.erase(std::unique(slots.begin(), slots.end()), slots.end());
    const std::vector<uint32_t> mine = ctx_all_values(ctx);
    if (!std::includes(slots.begin(), slots.end(), mine.begin(), mine.end()))
      return fail(KU_EINVAL, "all_values does not cover this shard's values");
  } else {
    const std::vector<uint32_t> mine = ctx_all_values(ctx);
    slots.insert(slots.end(), mine.begin(), mine.end());
  }
  // node universe: taxDB ids U DB values U {0, 1}, ascending => node 0 = taxid 0, node 1 = taxid 1
  std::vector<uint32_t> nodes(tax->ids);
  nodes.insert(nodes.end(), slots.begin(), slots.end());
  nodes.push_back(0);
  nodes.push_back(1);
  std::sort(nodes.begin(), nodes.end());
  nodes.erase(std::unique(nodes.begin(), nodes.end()), nodes.end());
  auto node_of = [&](uint32_t taxid) { return (uint32_t)(std::lower_bound(nodes.begin(), nodes.end(), taxid) - nodes.begin()); };
  std::vector<uint32_t> node_parent(nodes.size(), 0), node_slot(nodes.size(), 0), slot_node(slots.size(), 0);
  for (size_t i = 0; i < tax->ids.size(); ++i) {
    uint32_t p = tax->parent_map[i];  // Parent_map semantics: 0 = none
    if (tax->ids[i] != 0 && p != 0) node_parent[node_of(tax->ids[i])] = node_of(p);
  }
  for (size_t s = 1; s < slots.size(); ++s) {
    uint32_t nd = node_of(slots[s]);
    slot_node[s] = nd;
    node_slot[nd] = (uint32_t)s;
  }
  // root paths in slot space (same walk, same 4096-step guard as the kernels' parent chase had)
  std::vector<uint32_t> anc_off(slots.size() + 1, 0), anc;
  for (size_t s = 0; s < slots.size(); ++s) {
    anc_off[s] = (uint32_t)anc.size();
    if (s == 0) continue;
    uint32_t node = slot_node[s];
    for (uint32_t guard = 0; node > 0 && guard < 4096; ++guard) {
      if (node_slot[node]) anc.push_back(node_slot[node]);
      node = node_parent[node];
    }
  }
  anc_off[slots.size()] = (uint32_t)anc.size();
  ctx_free_tax(ctx);
  ctx->h_node_taxid = nodes;
  ctx->h_slot_taxid = slots;
  KU_TRY(upload(&ctx->d_node_parent, node_parent));
  KU_TRY(upload(&ctx->d_node_slot, node_slot));
  KU_TRY(upload(&ctx->d_node_taxid, nodes));
  KU_TRY(upload(&ctx->d_slot_node, slot_node));
  KU_TRY(upload(&ctx->d_slot_taxid, slots));
  KU_TRY(upload(&ctx->d_slot_anc_off, anc_off));
  KU_TRY(upload(&ctx->d_slot_anc, anc));
  ctx->tax.node_parent = ctx->d_node_parent;
  ctx->tax.node_slot = ctx->d_node_slot;
  ctx->tax.node_taxid = ctx->d_node_taxid;
  ctx->tax.slot_node = ctx->d_slot_node;
  ctx->tax.slot_taxid = ctx->d_slot_taxid;
  ctx->tax.slot_anc_off = ctx->d_slot_anc_off;
  ctx->tax.slot_anc = ctx->d_slot_anc;
  ctx->tax.n_nodes = (uint32_t)nodes.size();
  ctx->tax.n_slots = (uint32_t)slots.size();
  ctx->tax.node_one = 1;
  // per-taxon state before the stores are finalized: store_finalize remaps the values in place and drops the raw
  // pairs, so nothing that can fail for lack of memory may come after it
  if (hipMalloc((void **)&ctx->cnt.registers, (size_t)slots.size() * KU_HLL_M) != hipSuccess ||
      hipMalloc((void **)&ctx->cnt.n_kmers, slots.size() * 8) != hipSuccess ||
      hipMalloc((void **)&ctx->cnt.n_reads, nodes.size() * 8) != hipSuccess) {
    (void)hipGetLastError();
    ctx_free_tax(ctx);
    return fail(KU_ENOMEM, "device memory for the per-taxon counters");
  }
  int st = store_finalize(ctx, ctx->m);
  for (size_t e = 0; st == KU_OK && e < ctx->extra.size(); ++e) st = store_finalize(ctx, ctx->extra[e]);
  if (st != KU_OK) {  // a store may be half remapped: the context needs its database loaded again
    ctx_free_tax(ctx);
    ctx->db_loaded = false;
    return st;
  }
  ctx->tax_set = true;
  return ku_ctx_reset_counts(ctx);
}

extern "C" int ku_ctx_reset_counts(ku_ctx *ctx) {
  if (!ctx || !ctx->tax_set) return fail(KU_ESTATE, "taxonomy not set");
  if (ctx->rle_in_flight) return fail(KU_ESTATE, "ku_ctx_reset_counts: batches are in flight (ku_classify_batch_rle_finish first)");
  KU_TRY(ctx_activate(ctx));
  HIP_TRY(hipMemsetAsync(ctx->cnt.registers, 0, (size_t)ctx->tax.n_slots * KU_HLL_M, ctx->stream));
  HIP_TRY(hipMemsetAsync(ctx->cnt.n_kmers, 0, (size_t)ctx->tax.n_slots * 8, ctx->stream));
  HIP_TRY(hipMemsetAsync(ctx->cnt.n_reads, 0, (size_t)ctx->tax.n_nodes * 8, ctx->stream));
  if (ctx->d_exact_set) {
    HIP_TRY(hipMemsetAsync(ctx->d_exact_set, 0, (ctx->exact_mask + 1) * 8, ctx->stream));
    HIP_TRY(hipMemsetAsync(ctx->d_exact_unique, 0, (size_t)ctx->tax.n_slots * 8, ctx->stream));
    HIP_TRY(hipMemsetAsync(ctx->d_scalar + 6, 0, 4, ctx->stream));
  }
  if (ctx->m.seen_dirty && ctx->m.d_table) {  // the marks of the run before (sparse-sketch emulation, ku_device.h)
    KU_TRY(ku_launch_seen(2, ctx->m.d_table, ctx->m.db.n_lines, KuSparseDev{}, nullptr, ctx->stream));
    ctx->m.seen_dirty = false;
  }
  if (ctx->sp.on) {
    KuSparseDev &d = ctx->sp.dev;
    HIP_TRY(hipMemsetAsync(d.g_key, 0, (d.g_mask + 1) * 8, ctx->stream));
    HIP_TRY(hipMemsetAsync(d.dense, 0, (size_t)ctx->tax.n_slots * 4, ctx->stream));
    HIP_TRY(hipMemsetAsync(d.err, 0, 4, ctx->stream));
    HIP_TRY(hipMemsetAsync(ctx->sp.d_counters, 0, 32, ctx->stream));
    ctx->sp.acc_nt = 0;
    ctx->sp.open = false;
    ctx->sp.n_carry_l = ctx->sp.n_carry_u = 0;
    ctx->sp.g_count = 0;
    ctx->sp.tail_open = false;
    ctx->sp.tail_text.clear();
    ctx->sp.tail_len.clear();
  }
  HIP_TRY(hipStreamSynchronize(ctx->stream));
  return KU_OK;
}

static int check_ready(ku_ctx *ctx);
static int rle_idle(const ku_ctx *ctx, const char *who);
static int sparse_tail_close(ku_ctx *ctx);
static int sparse_tail_to_carry(ku_ctx *ctx);
static int sparse_pass(ku_ctx *ctx, const void *d_seqs, const uint64_t *d_off, const uint32_t *d_len, const uint64_t *h_off,
                       const uint32_t *h_len, uint64_t n_reads, uint64_t n_bytes, const uint32_t *d_taxa, uint32_t quick_min_hits,
                       hipStream_t s);

extern "C" int ku_ctx_enable_exact(ku_ctx *ctx, uint32_t capacity_log2) {
  KU_TRY(check_ready(ctx));
  if (capacity_log2 < 10 || capacity_log2 > 36) return fail(KU_EINVAL, "ku_ctx_enable_exact: capacity_log2 out of range (10..36)");
  if (ctx->d_exact_set) { (void)hipFree(ctx->d_exact_set); ctx->d_exact_set = nullptr; }
  if (ctx->d_exact_unique) { (void)hipFree(ctx->d_exact_unique); ctx->d_exact_unique = nullptr; }
  ctx->exact_mask = 0;
  const uint64_t cells = 1ull << capacity_log2;
  if (hipMalloc((void **)&ctx->d_exact_set, cells * 8) != hipSuccess) { ctx->d_exact_set = nullptr; return fail(KU_ENOMEM, "device memory for the exact k-mer set"); }
  if (hipMalloc((void **)&ctx->d_exact_unique, (size_t)ctx->tax.n_slots * 8) != hipSuccess) {
    (void)hipFree(ctx->d_exact_set);
    ctx->d_exact_set = ctx->d_exact_unique = nullptr;
    return fail(KU_ENOMEM, "device memory for the exact counters");
  }
  ctx->exact_mask = cells - 1;
  return ku_ctx_reset_counts(ctx);
}

extern "C" int ku_counts_export_exact(ku_ctx *ctx, uint64_t *unique_kmers) {
  KU_TRY(check_ready(ctx));
  if (!ctx->d_exact_set) return fail(KU_ESTATE, "exact counting is not enabled (ku_ctx_enable_exact)");
  if (!unique_kmers) return fail(KU_EINVAL, "ku_counts_export_exact: null buffer");
  HIP_TRY(hipStreamSynchronize(ctx->stream));
  uint32_t overflow = 0;
  HIP_TRY(hipMemcpy(&overflow, ctx->d_scalar + 6, 4, hipMemcpyDeviceToHost));
  if (overflow) return fail(KU_ENOMEM, "the exact k-mer set is full: enable it with a larger capacity");
  HIP_TRY(hipMemcpy(unique_kmers, ctx->d_exact_unique, (size_t)ctx->tax.n_slots * 8, hipMemcpyDeviceToHost));
  return KU_OK;
}

extern "C" int ku_ctx_count_taxons(ku_ctx *ctx, uint32_t *taxids, uint64_t *counts, uint64_t *n) {
  return ku_ctx_count_taxons_db(ctx, 0, taxids, counts, n);
}

extern "C" int ku_ctx_count_taxons_db(ku_ctx *ctx, uint32_t db_index, uint32_t *taxids, uint64_t *counts, uint64_t *n) {
  if (!ctx || !n) return fail(KU_EINVAL, "ku_ctx_count_taxons: null argument");
  if (!ctx->tax_set) return fail(KU_ESTATE, "taxonomy not set");
  if (db_index > ctx->extra.size()) return fail(KU_EINVAL, "database index out of range");
  const DbStore &d = db_index ? ctx->extra[db_index - 1] : ctx->m;
  KU_TRY(ctx_activate(ctx));
  const uint32_t ns = ctx->tax.n_slots;
  // callers ask twice (sizes, then values): the table is scanned once per resident store
  if (ctx->count_cache_store == (const void *)d.d_table && ctx->count_cache_pairs == (const void *)d.d_pairs && ctx->count_cache.size() == ns &&
      ctx->count_cache_lines == d.db.n_lines && d.db.n_lines + d.db.n_pairs) {
    const std::vector<unsigned long long> &h = ctx->count_cache;
    uint64_t m = 0;
    for (uint32_t s = 0; s < ns; ++s) if (h[s]) ++m;
    if (taxids && counts) {
      if (*n < m) return fail(KU_EINVAL, "output arrays too small");
      uint64_t j = 0;
      for (uint32_t s = 0; s < ns; ++s) if (h[s]) { taxids[j] = ctx->h_slot_taxid[s]; counts[j] = h[s]; ++j; }
    }
    *n = m;
    return KU_OK;
  }
  unsigned long long *d_c = nullptr;
  HIP_TRY(hipMalloc((void **)&d_c, (size_t)ns * 8));
  std::vector<unsigned long long> h(ns);
  int st = KU_OK;
  if (hipMemsetAsync(d_c, 0, (size_t)ns * 8, ctx->stream) != hipSuccess) st = KU_EHIP;
  if (st == KU_OK)
    st = d.db.table ? ku_launch_count_table(d.d_table, d.db.n_lines, d_c, ctx->stream)
                    : ku_launch_count_slots(d.d_pairs, d.db.n_pairs, d_c, ns, ctx->stream);
  if (st == KU_OK && hipMemcpyAsync(h.data(), d_c, (size_t)ns * 8, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess) st = KU_EHIP;
  if (st == KU_OK && hipStreamSynchronize(ctx->stream) != hipSuccess) st = KU_EHIP;
  (void)hipFree(d_c);
  if (st != KU_OK) return fail(st, "count_taxons kernel failed");
  ctx->count_cache = h;
  ctx->count_cache_store = d.d_table;
  ctx->count_cache_pairs = d.d_pairs;
  ctx->count_cache_lines = d.db.n_lines;
  uint64_t m = 0;
  for (uint32_t s = 0; s < ns; ++s) if (h[s]) ++m;
  if (taxids && counts) {
    if (*n < m) return fail(KU_EINVAL, "output arrays too small");
    uint64_t j = 0;
    for (uint32_t s = 0; s < ns; ++s) if (h[s]) { taxids[j] = ctx->h_slot_taxid[s]; counts[j] = h[s]; ++j; }
  }
  *n = m;
  return KU_OK;
}

// ---------------------------------------------------------------------------- HLL sparse-mode emulation
extern "C" int ku_ctx_enable_sparse(ku_ctx *ctx, uint64_t work_unit_nt, uint32_t global_log2) {
  KU_TRY(check_ready(ctx));
  KU_TRY(rle_idle(ctx, "ku_ctx_enable_sparse"));
  if (ctx->tax.n_slots > KU_SPARSE_MAX_SLOTS) return fail(KU_EUNSUP, "sparse-mode emulation handles up to 2^18 distinct database taxids");
  if (global_log2 == 0) global_log2 = 26;
  if (global_log2 < 10 || global_log2 > 34) return fail(KU_EINVAL, "ku_ctx_enable_sparse: global_log2 out of range (10..34)");
  if (ctx->sp.on) {  // a second call starts afresh with the new work-unit size
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    ctx_free_sparse(ctx);
  }
  ku_ctx::Sparse &sp = ctx->sp;
  KuSparseDev &d = sp.dev;
  // One pass of the emulation covers at most 2^25 bases (sparse_pass), the carried open unit at most 2^25 entries more: L
  // (distinct (unit, slot, encoding) of a pass) at 2^27 cells and U ((unit, slot) pairs: at most one per k-mer, and at
  // most n_slots per unit) sized from the slot count never run above half full, whatever the sample looks like
  const uint32_t l_log2 = 27;
  uint32_t u_log2 = 22;
  while (u_log2 < 27 && (1ull << u_log2) < 128ull * ctx->tax.n_slots) ++u_log2;
  d.l_mask = (1ull << l_log2) - 1;
  d.u_mask = (1ull << u_log2) - 1;
  d.g_mask = (1ull << global_log2) - 1;
  sp.cap_carry_l = std::min<uint64_t>(1024ull * ctx->tax.n_slots, 1ull << 25);
  sp.cap_carry_u = ctx->tax.n_slots;
  bool ok = hipMalloc((void **)&d.l_key, (d.l_mask + 1) * 8) == hipSuccess && hipMalloc((void **)&d.l_first, (d.l_mask + 1) * 4) == hipSuccess &&
            hipMalloc((void **)&d.u_key, (d.u_mask + 1) * 8) == hipSuccess && hipMalloc((void **)&d.u_distinct, (d.u_mask + 1) * 4) == hipSuccess &&
            hipMalloc((void **)&d.u_last, (d.u_mask + 1) * 4) == hipSuccess && hipMalloc((void **)&d.u_maxfirst, (d.u_mask + 1) * 4) == hipSuccess &&
            hipMalloc((void **)&d.g_key, (d.g_mask + 1) * 8) == hipSuccess && hipMalloc((void **)&d.dense, (size_t)ctx->tax.n_slots * 4) == hipSuccess &&
            hipMalloc((void **)&d.err, 4) == hipSuccess && hipMalloc((void **)&sp.d_counters, 32) == hipSuccess &&
            sp.carry_l.reserve(sp.cap_carry_l * 8) == KU_OK && sp.carry_u.reserve(sp.cap_carry_u * 12) == KU_OK;
  if (!ok) {
    for (void *p : {(void *)d.l_key, (void *)d.l_first, (void *)d.u_key, (void *)d.u_distinct, (void *)d.u_last, (void *)d.u_maxfirst,
                    (void *)d.g_key, (void *)d.dense, (void *)d.err, (void *)sp.d_counters})
      if (p) (void)hipFree(p);
    sp.carry_l.release();
    sp.carry_u.release();
    sp = ku_ctx::Sparse{};
    return fail(KU_ENOMEM, "device memory for the sparse-mode emulation");
  }
  d.g_count = sp.d_counters;
  sp.unit_nt = work_unit_nt;
  sp.on = true;
  return ku_ctx_reset_counts(ctx);
}

// room in the run-wide (slot, encoding) set for `incoming` more entries at load <= 1/2: a larger table takes over when
// the current one could fill (the set only grows with the distinct k-mers of the taxa that stay sparse -- on a run of
// many taxa that is most of what the reads hold)
// the kernels of every batch in flight are through (they run on streams of their own: what is about to replace a table they
// write to -- the run-wide set growing -- waits for them on the host; rare)
static int rle_drain_kernels(ku_ctx *ctx);

static int sparse_reserve_global(ku_ctx *ctx, uint64_t incoming, hipStream_t s) {
  ku_ctx::Sparse &sp = ctx->sp;
  KuSparseDev &d = sp.dev;
  const uint64_t need = 2 * (sp.g_count + incoming);
  if (need <= d.g_mask + 1) return KU_OK;
  KU_TRY(rle_drain_kernels(ctx));
  uint64_t cells = (d.g_mask + 1) * 2;
  while (cells < need) cells *= 2;
  const char *cap_env = getenv("KU_SPARSE_MAX_LOG2");  // test hook: a small ceiling stands in for a full device
  if (cells > (1ull << (cap_env ? atoi(cap_env) : 36))) return fail(KU_ENOMEM, "sparse-mode emulation: the run-wide set outgrew its ceiling");
  unsigned long long *nk = nullptr;
  if (hipMalloc((void **)&nk, cells * 8) != hipSuccess) {
    (void)hipGetLastError();
    return fail(KU_ENOMEM, "sparse-mode emulation: device memory for the run-wide set of encoded hashes (" + std::to_string(cells >> 17) + " MiB)");
  }
  unsigned long long *old = d.g_key;
  const uint64_t old_cells = d.g_mask + 1;
  HIP_TRY(hipMemsetAsync(nk, 0, cells * 8, s));
  HIP_TRY(hipMemsetAsync(d.g_count, 0, 8, s));
  d.g_key = nk;
  d.g_mask = cells - 1;
  KU_TRY(ku_launch_sparse_rehash(d, old, old_cells, s));
  unsigned long long n = 0;
  HIP_TRY(hipMemcpyAsync(&n, d.g_count, 8, hipMemcpyDeviceToHost, s));
  HIP_TRY(hipStreamSynchronize(s));
  (void)hipFree(old);
  sp.g_count = n;
  return KU_OK;
}

// The per-pass tables L / U of the exact evaluation, sized for what the pass holds: a prefix of the allocated arrays (the
// kernels that close a unit scan whole tables -- with 2^26 cells for a few hundred thousand entries those scans and the
// memsets were most of the emulation's cost in a `classify -r` run).  Returns the device view to hand to the kernels.
static int sparse_pass_tables(ku_ctx *ctx, uint64_t n_entries, KuSparseDev *view, hipStream_t s) {
  const KuSparseDev &d = ctx->sp.dev;
  uint64_t l_cells = 1ull << 14, u_cells = 1ull << 12;
  while (l_cells < 4 * n_entries && l_cells < d.l_mask + 1) l_cells <<= 1;
  while (u_cells < 2 * n_entries && u_cells < d.u_mask + 1) u_cells <<= 1;
  *view = d;
  view->l_mask = l_cells - 1;
  view->u_mask = u_cells - 1;
  return ku_launch_sparse_clear(*view, s);
}

// the reads [r0, r1) of a batch whose taxa[] holds slot ids: one pass of the emulation (at most KU_SPARSE_MAX_UNITS
// work units and 2^25 bases at a time; a unit that is still open at the end is carried into the next pass)
static int sparse_pass(ku_ctx *ctx, const void *d_seqs, const uint64_t *d_off, const uint32_t *d_len, const uint64_t *h_off,
                       const uint32_t *h_len, uint64_t n_reads, uint64_t n_bytes, const uint32_t *d_taxa, uint32_t quick_min_hits,
                       hipStream_t s) {
  ku_ctx::Sparse &sp = ctx->sp;
  KuSparseDev &d = sp.dev;
  KU_TRY(sparse_tail_to_carry(ctx));  // (an open unit the fast path left: into the form these passes carry along)
  if (n_bytes + 2 >= (1ull << 32)) return fail(KU_EUNSUP, "sparse-mode emulation: batches of at most 4 G bases");
  for (uint64_t i = 1; i < n_reads; ++i)
    if (h_off[i] < h_off[i - 1]) return fail(KU_EINVAL, "sparse-mode emulation: the reads of a batch must be in buffer order");
  std::vector<uint32_t> unit(n_reads);
  if (sp.unit.reserve(std::max<uint64_t>(n_reads, 1) * 4) != KU_OK) return fail(KU_ENOMEM, "device memory for the work-unit ids");
  uint64_t r0 = 0;
  while (r0 < n_reads) {
    // cut: units and bases of this pass
    uint32_t cur = 0;
    uint64_t acc = sp.acc_nt, bases = 0, r1 = r0;
    while (r1 < n_reads && cur < KU_SPARSE_MAX_UNITS && bases < (1ull << 25)) {
      unit[r1] = cur;
      acc += h_len[r1];
      bases += h_len[r1];
      ++r1;
      if (sp.unit_nt && acc >= sp.unit_nt) { ++cur; acc = 0; }  // the unit closes behind the read that fills it (classify.cpp:510-521)
    }
    const bool open_after = acc > 0 || (sp.unit_nt == 0 && (sp.open || r1 > r0));
    const uint32_t n_closed = cur;  // units 0 .. cur-1 are complete; unit `cur` (if any read fell into it) stays open
    KU_TRY(sparse_reserve_global(ctx, bases + sp.n_carry_l, s));
    KU_TRY(ku_launch_sparse_clear(d, s));
    if (sp.open) KU_TRY(ku_launch_sparse_carry_in(d, (const unsigned long long *)sp.carry_l.p, sp.n_carry_l, (const uint32_t *)sp.carry_u.p,
                                                   sp.n_carry_u, s));
    HIP_TRY(hipMemcpyAsync((uint32_t *)sp.unit.p + r0, unit.data() + r0, (r1 - r0) * 4, hipMemcpyHostToDevice, s));
    KU_TRY(ku_launch_sparse_insert(d, ctx->m.db.k, (const uint8_t *)d_seqs, d_off + r0, d_len + r0, (const uint32_t *)sp.unit.p + r0, r1 - r0,
                                   d_taxa, quick_min_hits, ctx->n_cu, s));
    KU_TRY(ku_launch_sparse_close(d, n_closed, s));
    sp.n_carry_l = sp.n_carry_u = 0;
    if (open_after) {
      HIP_TRY(hipMemsetAsync(sp.d_counters + 1, 0, 16, s));
      KU_TRY(ku_launch_sparse_carry_out(d, cur, (unsigned long long *)sp.carry_l.p, (uint32_t *)sp.carry_u.p, sp.d_counters + 1, sp.cap_carry_l,
                                        sp.cap_carry_u, s));
      unsigned long long c[3] = {0, 0, 0};
      HIP_TRY(hipMemcpyAsync(c, sp.d_counters, 24, hipMemcpyDeviceToHost, s));
      HIP_TRY(hipStreamSynchronize(s));
      sp.g_count = c[0];
      sp.n_carry_l = std::min<uint64_t>(c[1], sp.cap_carry_l);
      sp.n_carry_u = std::min<uint64_t>(c[2], sp.cap_carry_u);
    } else {
      unsigned long long c = 0;
      HIP_TRY(hipMemcpyAsync(&c, sp.d_counters, 8, hipMemcpyDeviceToHost, s));
      HIP_TRY(hipStreamSynchronize(s));  // `unit` is reused by the next pass
      sp.g_count = c;
    }
    sp.open = open_after;
    sp.acc_nt = acc;
    r0 = r1;
  }
  uint32_t err = 0;
  HIP_TRY(hipMemcpyAsync(&err, d.err, 4, hipMemcpyDeviceToHost, s));
  HIP_TRY(hipStreamSynchronize(s));
  if (err) return fail(KU_ENOMEM, std::string("sparse-mode emulation: a device table is full (") + ((err & 1) ? "L " : "") + ((err & 2) ? "U " : "") +
                                      ((err & 4) ? "G" : "") + ")");
  return KU_OK;
}

// the unit that is still open ends here (end of an input file / of the run): evaluate and commit what was carried
static int sparse_close_open_unit(ku_ctx *ctx) {
  ku_ctx::Sparse &sp = ctx->sp;
  hipStream_t s = ctx->stream;
  KU_TRY(sparse_tail_close(ctx));  // (a unit the fast path kept as its reads + insert counts)
  if (sp.open) {
    KU_TRY(sparse_reserve_global(ctx, sp.n_carry_l, s));
    KuSparseDev d;
    KU_TRY(sparse_pass_tables(ctx, sp.n_carry_l + sp.n_carry_u, &d, s));
    KU_TRY(ku_launch_sparse_carry_in(d, (const unsigned long long *)sp.carry_l.p, sp.n_carry_l, (const uint32_t *)sp.carry_u.p, sp.n_carry_u, s));
    KU_TRY(ku_launch_sparse_close(d, 1, s));
    unsigned long long c = 0;
    HIP_TRY(hipMemcpyAsync(&c, sp.d_counters, 8, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    sp.g_count = c;
  }
  sp.open = false;
  sp.acc_nt = 0;
  sp.n_carry_l = sp.n_carry_u = 0;
  return KU_OK;
}

// The entries the fused kernel's fast path marked in the probe table (ku_device.h: SEEN bytes) join the run-wide set: for whoever
// needs the set as such (ku_sparse_export, the union of several ranks' sets, a table that is about to go).  The marks stay.
static int ctx_seen_harvest(ku_ctx *ctx) {
  if (!ctx->sp.on || !ctx->m.seen_dirty || !ctx->m.d_table) return KU_OK;
  hipStream_t s = ctx->stream;
  unsigned long long *d_n = ctx->sp.d_counters + 3, n = 0;
  HIP_TRY(hipMemsetAsync(d_n, 0, 8, s));
  KU_TRY(ku_launch_seen(0, ctx->m.d_table, ctx->m.db.n_lines, ctx->sp.dev, d_n, s));
  HIP_TRY(hipMemcpyAsync(&n, d_n, 8, hipMemcpyDeviceToHost, s));
  HIP_TRY(hipStreamSynchronize(s));
  if (n == 0) return KU_OK;
  KU_TRY(sparse_reserve_global(ctx, n, s));
  KU_TRY(ku_launch_seen(1, ctx->m.d_table, ctx->m.db.n_lines, ctx->sp.dev, ctx->sp.dev.g_count, s));
  unsigned long long c = 0;
  uint32_t err = 0;
  HIP_TRY(hipMemcpyAsync(&c, ctx->sp.dev.g_count, 8, hipMemcpyDeviceToHost, s));
  HIP_TRY(hipMemcpyAsync(&err, ctx->sp.dev.err, 4, hipMemcpyDeviceToHost, s));
  HIP_TRY(hipStreamSynchronize(s));
  ctx->sp.g_count = c;
  if (err) return fail(KU_ENOMEM, "sparse-mode emulation: the run-wide set is full");
  return KU_OK;
}

extern "C" int ku_ctx_disable_sparse(ku_ctx *ctx) {
  if (!ctx) return fail(KU_EINVAL, "null context");
  KU_TRY(rle_idle(ctx, "ku_ctx_disable_sparse"));
  KU_TRY(ctx_activate(ctx));
  HIP_TRY(hipStreamSynchronize(ctx->stream));
  ctx_free_sparse(ctx);
  return KU_OK;
}

extern "C" int ku_ctx_sparse_state(const ku_ctx *ctx) { return !ctx ? 0 : (ctx->sp.on ? 1 : (ctx->sp.gave_up ? 2 : 0)); }

extern "C" int ku_sparse_close_unit(ku_ctx *ctx) {
  KU_TRY(check_ready(ctx));
  KU_TRY(rle_idle(ctx, "ku_sparse_close_unit"));
  if (!ctx->sp.on) return KU_OK;
  if (ctx->sp.unit_nt == 0) return KU_OK;  // one unit for the whole run
  return sparse_close_open_unit(ctx);
}

extern "C" int ku_sparse_export(ku_ctx *ctx, uint8_t *slot_is_sparse, uint64_t *pairs, uint64_t *n_pairs) {
  KU_TRY(check_ready(ctx));
  KU_TRY(rle_idle(ctx, "ku_sparse_export"));
  if (!ctx->sp.on) return fail(KU_ESTATE, "sparse-mode emulation is not enabled (ku_ctx_enable_sparse)");
  if (!n_pairs) return fail(KU_EINVAL, "ku_sparse_export: null argument");
  KU_TRY(sparse_close_open_unit(ctx));
  KU_TRY(ctx_seen_harvest(ctx));
  ku_ctx::Sparse &sp = ctx->sp;
  KuSparseDev &d = sp.dev;
  hipStream_t s = ctx->stream;  // end of the run: the last, partial work unit closes (classify.cpp:522-523)
  unsigned long long total = 0;
  uint32_t err = 0;
  HIP_TRY(hipMemcpyAsync(&total, d.g_count, 8, hipMemcpyDeviceToHost, s));
  HIP_TRY(hipMemcpyAsync(&err, d.err, 4, hipMemcpyDeviceToHost, s));
  HIP_TRY(hipStreamSynchronize(s));
  if (err) return fail(KU_ENOMEM, "sparse-mode emulation: a device table overflowed");
  if (sp.out.reserve(std::max<uint64_t>(total, 1) * 8) != KU_OK) return fail(KU_ENOMEM, "device memory for the sparse export");
  HIP_TRY(hipMemsetAsync(sp.d_counters + 3, 0, 8, s));
  KU_TRY(ku_launch_sparse_export(d, (unsigned long long *)sp.out.p, total, sp.d_counters + 3, s));
  unsigned long long n = 0;
  HIP_TRY(hipMemcpyAsync(&n, sp.d_counters + 3, 8, hipMemcpyDeviceToHost, s));
  HIP_TRY(hipStreamSynchronize(s));
  if (slot_is_sparse) {
    std::vector<uint32_t> dense(ctx->tax.n_slots);
    HIP_TRY(hipMemcpy(dense.data(), d.dense, (size_t)ctx->tax.n_slots * 4, hipMemcpyDeviceToHost));
    for (uint32_t i = 0; i < ctx->tax.n_slots; ++i) slot_is_sparse[i] = dense[i] ? 0 : 1;
  }
  if (pairs) {
    if (*n_pairs < n) return fail(KU_EINVAL, "ku_sparse_export: output array too small");
    if (n) HIP_TRY(hipMemcpy(pairs, sp.out.p, n * 8, hipMemcpyDeviceToHost));
  }
  *n_pairs = n;
  return KU_OK;
}

// ---- the emulation over several GPUs (ku_mgpu.cpp): every rank runs it on whole work units of the read stream, the open
// unit travels to the rank that classifies the next reads, and the ranks' states are folded into one at the end of the run
// (a taxon's global sketch is dense iff some unit made it dense, on whichever rank; else it holds every encoding of the
// run: the union of the ranks' sets)
int ku_ctx_sparse_on(const ku_ctx *ctx) { return ctx && ctx->sp.on ? 1 : 0; }
uint64_t ku_ctx_sparse_unit_nt(const ku_ctx *ctx) { return ctx ? ctx->sp.unit_nt : 0; }
// one pass of the emulation over reads whose per-k-mer array holds slot ids (the sharded path, between the exchange and
// the resolve stage); KU_ENOMEM switches the emulation off on this context like the single-GPU path does
int ku_ctx_sparse_pass_slots(ku_ctx *ctx, const void *d_seqs, const uint64_t *d_off, const uint32_t *d_len, const uint64_t *h_off,
                             const uint32_t *h_len, uint64_t n_reads, uint64_t n_bytes, const uint32_t *d_taxa, uint32_t quick_min_hits,
                             hipStream_t s) {
  KU_TRY(check_ready(ctx));
  if (!ctx->sp.on || n_reads == 0) return KU_OK;
  int st = sparse_pass(ctx, d_seqs, d_off, d_len, h_off, h_len, n_reads, n_bytes, d_taxa, quick_min_hits, s);
  if (st == KU_ENOMEM) {
    (void)hipStreamSynchronize(s);
    (void)hipGetLastError();
    ctx_free_sparse(ctx);
    ctx->sp.gave_up = true;
    return KU_OK;
  }
  return st;
}
// the unit that is still open on `src` continues on `dst` (same process; the contexts may sit on different devices)
int ku_ctx_sparse_move_open_unit(ku_ctx *src, ku_ctx *dst) {
  if (!src || !dst || !src->sp.on || !dst->sp.on) return fail(KU_ESTATE, "sparse-mode emulation is not enabled on both contexts");
  if (src == dst || !(src->sp.open || src->sp.tail_open)) return KU_OK;
  if (dst->sp.open || dst->sp.tail_open) return fail(KU_ESTATE, "the destination context holds an open work unit of its own");
  KU_TRY(ctx_activate(src));
  KU_TRY(sparse_tail_to_carry(src));  // (what travels is the staged form: L / U entries)
  HIP_TRY(hipStreamSynchronize(src->stream));
  KU_TRY(ctx_activate(dst));
  HIP_TRY(hipStreamSynchronize(dst->stream));
  if (src->sp.n_carry_l > dst->sp.cap_carry_l || src->sp.n_carry_u > dst->sp.cap_carry_u) return fail(KU_ESTATE, "carry buffers differ between the contexts");
  if (src->sp.n_carry_l) HIP_TRY(hipMemcpy(dst->sp.carry_l.p, src->sp.carry_l.p, src->sp.n_carry_l * 8, hipMemcpyDefault));
  if (src->sp.n_carry_u) HIP_TRY(hipMemcpy(dst->sp.carry_u.p, src->sp.carry_u.p, src->sp.n_carry_u * 12, hipMemcpyDefault));
  dst->sp.n_carry_l = src->sp.n_carry_l;
  dst->sp.n_carry_u = src->sp.n_carry_u;
  dst->sp.open = true;
  dst->sp.acc_nt = src->sp.acc_nt;
  src->sp.open = false;
  src->sp.acc_nt = 0;
  src->sp.n_carry_l = src->sp.n_carry_u = 0;
  return KU_OK;
}
// end of the run on this rank: the last, partial unit closes; dense flags out (host, one per slot)
int ku_ctx_sparse_finish(ku_ctx *ctx, uint32_t *h_dense) {
  KU_TRY(check_ready(ctx));
  if (!ctx->sp.on) return fail(KU_ESTATE, "sparse-mode emulation is not enabled");
  KU_TRY(sparse_close_open_unit(ctx));
  uint32_t err = 0;
  HIP_TRY(hipMemcpyAsync(&err, ctx->sp.dev.err, 4, hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(hipMemcpyAsync(h_dense, ctx->sp.dev.dense, (size_t)ctx->tax.n_slots * 4, hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(hipStreamSynchronize(ctx->stream));
  if (err) return fail(KU_ENOMEM, "sparse-mode emulation: a device table overflowed");
  return KU_OK;
}
int ku_ctx_sparse_set_dense(ku_ctx *ctx, const uint32_t *h_dense) {
  KU_TRY(check_ready(ctx));
  if (!ctx->sp.on) return fail(KU_ESTATE, "sparse-mode emulation is not enabled");
  HIP_TRY(hipMemcpyAsync(ctx->sp.dev.dense, h_dense, (size_t)ctx->tax.n_slots * 4, hipMemcpyHostToDevice, ctx->stream));
  HIP_TRY(hipStreamSynchronize(ctx->stream));
  return KU_OK;
}
// the run-wide set of `src` joins that of `dst` (entries of slots that are dense by now are dropped on the way)
int ku_ctx_sparse_absorb(ku_ctx *dst, ku_ctx *src) {
  if (!dst || !src || !dst->sp.on || !src->sp.on) return fail(KU_ESTATE, "sparse-mode emulation is not enabled on both contexts");
  KU_TRY(ctx_activate(src));
  KU_TRY(ctx_seen_harvest(src));  // (what src's fused kernel marked in its probe table; after the group's dense flags were set)
  HIP_TRY(hipStreamSynchronize(src->stream));
  unsigned long long n_src = 0;
  HIP_TRY(hipMemcpy(&n_src, src->sp.dev.g_count, 8, hipMemcpyDeviceToHost));
  KU_TRY(ctx_activate(dst));
  hipStream_t s = dst->stream;
  KU_TRY(sparse_reserve_global(dst, n_src, s));
  const uint64_t cells = src->sp.dev.g_mask + 1, step = 1ull << 23;  // 64 MB of cells at a time
  if (dst->sp.out.reserve(std::min(cells, step) * 8) != KU_OK) return fail(KU_ENOMEM, "device memory for the merge of the sparse sets");
  for (uint64_t c0 = 0; c0 < cells; c0 += step) {
    const uint64_t n = std::min(step, cells - c0);
    HIP_TRY(hipMemcpyAsync(dst->sp.out.p, src->sp.dev.g_key + c0, n * 8, hipMemcpyDefault, s));
    KU_TRY(ku_launch_sparse_absorb(dst->sp.dev, (const unsigned long long *)dst->sp.out.p, n, s));
  }
  unsigned long long c = 0;
  uint32_t err = 0;
  HIP_TRY(hipMemcpyAsync(&c, dst->sp.dev.g_count, 8, hipMemcpyDeviceToHost, s));
  HIP_TRY(hipMemcpyAsync(&err, dst->sp.dev.err, 4, hipMemcpyDeviceToHost, s));
  HIP_TRY(hipStreamSynchronize(s));
  dst->sp.g_count = c;
  if (err) return fail(KU_ENOMEM, "sparse-mode emulation: the run-wide set is full");
  return KU_OK;
}

// ---------------------------------------------------------------------------- classification
static int check_ready(ku_ctx *ctx) {
  if (!ctx) return fail(KU_EINVAL, "null context");
  if (!ctx->db_loaded) return fail(KU_ESTATE, "no database loaded");
  if (!ctx->tax_set) return fail(KU_ESTATE, "taxonomy not set");
  return ctx_activate(ctx);
}

extern "C" int ku_lookup_device(ku_ctx *ctx, const void *d_seqs, uint64_t n_bytes, const ku_opts *opts,
                                uint32_t *d_taxa, void *stream) {
  KU_TRY(check_ready(ctx));
  if (n_bytes && (!d_seqs || !d_taxa)) return fail(KU_EINVAL, "ku_lookup_device: null buffer");
  const uint32_t flags = opts ? opts->flags : 0;
  // quick mode counts only the scanned prefix of each read -> accounted in the resolve stage
  const bool counts = !(flags & (KU_F_NO_COUNTS | KU_F_QUICK));
  hipStream_t s = stream ? (hipStream_t)stream : ctx->stream;
  // hierarchical run: one pass per database in command-line order; later passes only search the positions that
  // are still 0, the last one does the per-taxon accounting (classify.cpp:928-939)
  const size_t nd = 1 + ctx->extra.size();
  for (size_t i = 0; i < nd; ++i) {
    const DbStore &d = i ? ctx->extra[i - 1] : ctx->m;
    int st = ku_launch_lookup(d.db, ctx->cnt, (const uint8_t *)d_seqs, n_bytes, d_taxa, counts && i + 1 == nd, i > 0,
                              (flags & KU_F_MERGE_CHUNK) != 0, ctx->n_cu, s);
    if (st != KU_OK) return fail(st, "lookup kernel launch failed");
  }
  return KU_OK;
}

// ---- owner routing (ku_mgpu.cpp): the context's database / counters behind the three kernels
int ku_ctx_route_info(const ku_ctx *ctx, uint64_t *bin_lo, uint64_t *bin_hi, int *is_hash, int *single_db) {
  if (!ctx || !ctx->db_loaded || !ctx->tax_set) return fail(KU_ESTATE, "no database / taxonomy on this context");
  if (bin_lo) *bin_lo = ctx->m.db.bin_lo;
  if (bin_hi) *bin_hi = ctx->m.db.bin_hi;
  if (is_hash) *is_hash = ctx->m.db.table != nullptr;
  if (single_db) *single_db = ctx->extra.empty();
  return KU_OK;
}
int ku_ctx_route_scan(ku_ctx *ctx, const void *d_seqs, uint64_t n_bytes, uint32_t *d_taxa, const KuRouteDev &rt, hipStream_t s) {
  KU_TRY(check_ready(ctx));
  int st = ku_launch_route_scan(ctx->m.db, (const uint8_t *)d_seqs, n_bytes, d_taxa, rt, ctx->n_cu, s);
  return st == KU_OK ? KU_OK : fail(st, "route scan kernel launch failed");
}
// whether the resolve stage of a routed step can run as the fused kernel's ROUTE instance (KU_EUNSUP: no -- quick mode, reads
// beyond 65535 k-mers, unknown read length); reserves the windowed instance's spill workspace for any number of reads, so
// that the per-round calls below never reallocate it under a kernel of the other stream
int ku_ctx_route_resolve_prepare(ku_ctx *ctx, const ku_opts *opts, hipStream_t s) {
  KU_TRY(check_ready(ctx));
  const uint32_t flags = opts ? opts->flags : 0;
  const uint32_t max_len = opts ? opts->max_read_len : 0;
  if (max_len == 0 || (flags & (KU_F_QUICK | KU_F_KEEP_SLOTS)) || getenv("KU_NO_FUSED")) return KU_EUNSUP;  // (no message: the caller has another path)
  const uint32_t max_n = max_len >= ctx->m.db.k ? max_len - ctx->m.db.k + 1 : 0;
  if (max_n > ku_route_resolve_max_kmers()) return KU_EUNSUP;
  if (max_n > 128) {
    const uint64_t ws = ku_short_workspace_bytes(std::max(max_n, 193u), ctx->tax.n_slots, ~0ull >> 8, ctx->n_cu);
    if (ws > ctx->b_ws.cap) {
      HIP_TRY(hipStreamSynchronize(s));
      if (ctx->b_ws.reserve(ws) != KU_OK) { (void)hipGetLastError(); return KU_EUNSUP; }
    }
  }
  return KU_OK;
}
int ku_ctx_route_resolve(ku_ctx *ctx, const uint64_t *d_off, const uint32_t *d_len, uint64_t n_reads, const ku_opts *opts, uint32_t *d_calls,
                         uint32_t *d_taxa, uint32_t *d_hits, const uint32_t *d_kb, const uint32_t *d_ret, hipStream_t s) {
  const uint32_t flags = opts ? opts->flags : 0;
  const uint32_t max_len = opts ? opts->max_read_len : 0;
  const uint32_t max_n = max_len >= ctx->m.db.k ? max_len - ctx->m.db.k + 1 : 0;
  int st = ku_launch_route_resolve(ctx->m.db, ctx->tax, ctx->cnt, d_off, d_len, n_reads, max_n, flags, d_calls, d_taxa, d_hits, d_kb, d_ret,
                                   ctx->b_ws.p, ctx->b_ws.cap, ctx->n_cu, s);
  return st == KU_OK ? KU_OK : fail(st, "routed resolve kernel launch failed");
}
int ku_ctx_route_owner(ku_ctx *ctx, const void *d_rec, uint64_t n_rec, const uint32_t *d_kb, uint32_t *d_slots, bool do_counts, hipStream_t s) {
  KU_TRY(check_ready(ctx));
  int st = ku_launch_route_owner(ctx->m.db, ctx->cnt, d_rec, n_rec, d_kb, d_slots, do_counts, ctx->n_cu, s);
  return st == KU_OK ? KU_OK : fail(st, "route owner kernel launch failed");
}

int ku_exact_owned_step(ku_ctx *ctx, const void *d_seqs, const uint64_t *d_off, const uint32_t *d_len, uint64_t n_reads, uint64_t n_bytes,
                        const ku_opts *opts, uint32_t *d_taxa, hipStream_t s) {
  KU_TRY(check_ready(ctx));
  if (!ctx->d_exact_set) return fail(KU_ESTATE, "exact counting is not enabled on this context");
  if (!ctx->extra.empty()) return fail(KU_EUNSUP, "exact counting on a shard goes with one database");
  ku_opts o = opts ? *opts : ku_opts{0, 1, 0, 0};
  if (o.flags & (KU_F_QUICK | KU_F_NO_COUNTS)) return fail(KU_EUNSUP, "exact counting goes with the plain classification only");
  o.flags |= KU_F_KEEP_SLOTS | KU_F_MERGE_CHUNK;
  HIP_TRY(hipMemsetAsync(d_taxa, 0xFE, n_bytes * 4, s));
  KU_TRY(ku_lookup_device(ctx, d_seqs, n_bytes, &o, d_taxa, s));
  int st = ku_launch_exact(ctx->m.db.k, (const uint8_t *)d_seqs, d_off, d_len, n_reads, d_taxa, ctx->d_exact_set, ctx->exact_mask,
                           ctx->d_exact_unique, ctx->d_scalar + 6, ctx->n_cu, s);
  if (st != KU_OK) return fail(st, "exact counting kernel launch failed");
  return ku_launch_replace_u32(d_taxa, n_bytes, KU_FOREIGN_MARK, 0u, s);
}

extern "C" int ku_lookup_stats_device(ku_ctx *ctx, const void *d_seqs, uint64_t n_bytes, uint64_t *stats_out,
                                      void *stream) {
  KU_TRY(check_ready(ctx));
  if (!stats_out || (n_bytes && !d_seqs)) return fail(KU_EINVAL, "ku_lookup_stats_device: null argument");
  hipStream_t s = stream ? (hipStream_t)stream : ctx->stream;
  unsigned long long *d_stats = (unsigned long long *)(ctx->d_scalar + 8);  // 32 bytes at offset 32
  HIP_TRY(hipMemsetAsync(d_stats, 0, 32, s));
  int st = ku_launch_lookup_stats(ctx->m.db, (const uint8_t *)d_seqs, n_bytes, d_stats, ctx->n_cu, s);
  if (st != KU_OK) return fail(st, "stats kernel launch failed");
  HIP_TRY(hipMemcpyAsync(stats_out, d_stats, 32, hipMemcpyDeviceToHost, s));
  HIP_TRY(hipStreamSynchronize(s));
  return KU_OK;
}

extern "C" int ku_resolve_device(ku_ctx *ctx, const void *d_seqs, const uint64_t *d_seq_off, const uint32_t *d_seq_len,
                                 uint64_t n_reads, const ku_opts *opts, uint32_t *d_calls, uint32_t *d_taxa,
                                 uint32_t *d_hits, void *stream) {
  KU_TRY(check_ready(ctx));
  if (n_reads && (!d_seq_off || !d_seq_len || !d_calls || !d_taxa)) return fail(KU_EINVAL, "ku_resolve_device: null buffer");
  const uint32_t flags = opts ? opts->flags : 0;
  if ((flags & KU_F_QUICK) && !d_seqs) return fail(KU_EINVAL, "quick mode needs the sequence buffer");
  hipStream_t s = stream ? (hipStream_t)stream : ctx->stream;
  uint32_t max_len = opts ? opts->max_read_len : 0;
  if (max_len == 0 && n_reads) {
    KU_TRY(ku_launch_max_len(d_seq_len, n_reads, ctx->d_scalar + 4, s));
    HIP_TRY(hipMemcpyAsync(&max_len, ctx->d_scalar + 4, 4, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
  }
  uint64_t ws = (flags & KU_F_QUICK) ? 0 : ku_resolve_workspace_bytes(max_len, ctx->m.db.k, ctx->n_cu);
  if (ws > ctx->b_ws.cap) {
    HIP_TRY(hipStreamSynchronize(s));
    if (ctx->b_ws.reserve(ws) != KU_OK) return fail(KU_ENOMEM, "resolve workspace allocation failed");
  }
  if (ws) HIP_TRY(hipMemsetAsync(ctx->b_ws.p, 0, ws, s));
  int st = ku_launch_resolve(ctx->m.db, ctx->tax, ctx->cnt, (const uint8_t *)d_seqs, d_seq_off, d_seq_len, n_reads, flags,
                             opts ? opts->min_hits : 1, max_len, d_calls, d_taxa, d_hits, ctx->b_ws.p, ctx->b_ws.cap,
                             ctx->n_cu, s);
  return st == KU_OK ? KU_OK : fail(st, "resolve kernel launch failed");
}

// h_off / h_len: host copies of the read offsets / lengths when the caller has them (the host-buffer entry points);
// the sparse-mode emulation needs them for the work-unit plan
static int classify_device_impl(ku_ctx *ctx, const void *d_seqs, uint64_t n_bytes, const uint64_t *d_seq_off,
                                const uint32_t *d_seq_len, uint64_t n_reads, const ku_opts *opts, uint32_t *d_calls,
                                uint32_t *d_taxa, uint32_t *d_hits, void *stream, const uint64_t *h_off, const uint32_t *h_len) {
  KU_TRY(check_ready(ctx));
  const uint32_t flags = opts ? opts->flags : 0;
  // short reads against the resident probe table: one fused kernel, a wave per read (ku_short.hip)
  const bool exact = ctx->d_exact_set != nullptr;
  if (exact && !store_whole(ctx->m)) return fail(KU_EUNSUP, "exact counting on a shard runs through the multi-GPU driver (ku_mgpu_enable_exact)");
  if (exact && (flags & (KU_F_KEEP_SLOTS | KU_F_NO_COUNTS)))
    return fail(KU_EUNSUP, "exact counting goes with a whole classification (no slot output / count-less runs)");
  const bool sparse = ctx->sp.on && !(flags & KU_F_NO_COUNTS);
  if (sparse && !h_len) return fail(KU_EUNSUP, "the sparse-mode emulation runs through the host-buffer entry points (it needs the read lengths on the host)");
  if (sparse && (flags & KU_F_KEEP_SLOTS)) return fail(KU_EUNSUP, "the sparse-mode emulation does not combine with slot output");
  const uint32_t short_max = (getenv("KU_NO_FUSED") || !ctx->extra.empty() || exact || sparse) ? 0 : ku_short_max_kmers(ctx->m.db);
  if (short_max && !(flags & (KU_F_QUICK | KU_F_KEEP_SLOTS)) && n_reads) {
    if (!d_seqs || !d_seq_off || !d_seq_len || !d_calls || !d_taxa) return fail(KU_EINVAL, "ku_classify_batch_device: null buffer");
    hipStream_t s = stream ? (hipStream_t)stream : ctx->stream;
    uint32_t max_len = opts ? opts->max_read_len : 0;
    if (max_len == 0) {
      KU_TRY(ku_launch_max_len(d_seq_len, n_reads, ctx->d_scalar + 4, s));
      HIP_TRY(hipMemcpyAsync(&max_len, ctx->d_scalar + 4, 4, hipMemcpyDeviceToHost, s));
      HIP_TRY(hipStreamSynchronize(s));
    }
    const uint32_t max_n = max_len >= ctx->m.db.k ? max_len - ctx->m.db.k + 1 : 0;
    // longer reads (mate pairs, long reads up to 65535 k-mers): the same kernel in windows of 128 k-mers -- when its
    // spill workspace can be had; KU_NO_WINDOWED=1 keeps them on the flat lookup + resolve kernels
    bool fused = max_n <= short_max;
    if (!fused && max_n <= ku_short_max_kmers_windowed(ctx->m.db) && !getenv("KU_NO_WINDOWED")) {
      const uint64_t ws = ku_short_workspace_bytes(max_n, ctx->tax.n_slots, n_reads, ctx->n_cu);
      if (ws > ctx->b_ws.cap) HIP_TRY(hipStreamSynchronize(s));
      fused = ctx->b_ws.reserve(ws) == KU_OK;
      if (!fused) (void)hipGetLastError();
    }
    if (fused) {
      int st = ku_launch_classify_short(ctx->m.db, ctx->tax, ctx->cnt, (const uint8_t *)d_seqs, n_bytes, d_seq_off, d_seq_len,
                                        n_reads, max_n, flags, d_calls, d_taxa, d_hits, ctx->b_ws.p, ctx->b_ws.cap, ctx->n_cu, s);
      return st == KU_OK ? KU_OK : fail(st, "fused short-read kernel launch failed");
    }
    ku_opts o = opts ? *opts : ku_opts{0, 1, 0, 0};
    o.max_read_len = max_len;
    KU_TRY(ku_lookup_device(ctx, d_seqs, n_bytes, &o, d_taxa, stream));
    return ku_resolve_device(ctx, d_seqs, d_seq_off, d_seq_len, n_reads, &o, d_calls, d_taxa, d_hits, stream);
  }
  KU_TRY(ku_lookup_device(ctx, d_seqs, n_bytes, opts, d_taxa, stream));
  if (sparse && n_reads) {  // between the stages: d_taxa holds slot ids
    int st = sparse_pass(ctx, d_seqs, d_seq_off, d_seq_len, h_off, h_len, n_reads, n_bytes, d_taxa,
                         (flags & KU_F_QUICK) ? std::max(1u, opts ? opts->min_hits : 1u) : 0u, stream ? (hipStream_t)stream : ctx->stream);
    if (st == KU_ENOMEM) {
      // no room for the emulation's tables: the classification itself does not depend on them -- the run goes on with
      // the dense registers alone and says so (ku_ctx_sparse_state; the reports then carry their estimates)
      (void)hipStreamSynchronize(stream ? (hipStream_t)stream : ctx->stream);
      (void)hipGetLastError();
      ctx_free_sparse(ctx);
      ctx->sp.gave_up = true;
    } else if (st != KU_OK) return st;
  }
  if (exact) {  // between the stages: d_taxa holds slot ids
    if (n_reads && (!d_seqs || !d_seq_off || !d_seq_len || !d_taxa)) return fail(KU_EINVAL, "ku_classify_batch_device: null buffer");
    int st = ku_launch_exact(ctx->m.db.k, (const uint8_t *)d_seqs, d_seq_off, d_seq_len, n_reads, d_taxa, ctx->d_exact_set,
                             ctx->exact_mask, ctx->d_exact_unique, ctx->d_scalar + 6, ctx->n_cu,
                             stream ? (hipStream_t)stream : ctx->stream,
                             (flags & KU_F_QUICK) ? std::max(1u, opts ? opts->min_hits : 1u) : 0u);
    if (st != KU_OK) return fail(st, "exact counting kernel launch failed");
  }
  return ku_resolve_device(ctx, d_seqs, d_seq_off, d_seq_len, n_reads, opts, d_calls, d_taxa, d_hits, stream);
}

extern "C" int ku_classify_batch_device(ku_ctx *ctx, const void *d_seqs, uint64_t n_bytes, const uint64_t *d_seq_off,
                                        const uint32_t *d_seq_len, uint64_t n_reads, const ku_opts *opts,
                                        uint32_t *d_calls, uint32_t *d_taxa, uint32_t *d_hits, void *stream) {
  return classify_device_impl(ctx, d_seqs, n_bytes, d_seq_off, d_seq_len, n_reads, opts, d_calls, d_taxa, d_hits, stream, nullptr, nullptr);
}

extern "C" int ku_classify_batch(ku_ctx *ctx, const char *seqs, uint64_t n_bytes, const uint64_t *seq_off,
                                 const uint32_t *seq_len, uint64_t n_reads, const ku_opts *opts, uint32_t *calls,
                                 uint32_t *taxa, uint32_t *hits) {
  KU_TRY(check_ready(ctx));
  if ((n_bytes && !seqs) || (n_reads && (!seq_off || !seq_len || !calls))) return fail(KU_EINVAL, "ku_classify_batch: null buffer");
  if (n_reads == 0) return KU_OK;
  ku_opts o = opts ? *opts : ku_opts{0, 1, 0, 0};
  if (o.max_read_len == 0) for (uint64_t i = 0; i < n_reads; ++i) o.max_read_len = std::max(o.max_read_len, seq_len[i]);
  for (uint64_t i = 0; i < n_reads; ++i)
    if (seq_off[i] + seq_len[i] > n_bytes) return fail(KU_EINVAL, "read " + std::to_string(i) + " exceeds the sequence buffer");
  if (ctx->b_seqs.reserve(n_bytes + 16) || ctx->b_off.reserve(n_reads * 8) || ctx->b_len.reserve(n_reads * 4) ||
      ctx->b_calls.reserve(n_reads * 4) || ctx->b_taxa.reserve((n_bytes + 16) * 4) || ctx->b_hits.reserve(n_reads * 4))
    return fail(KU_ENOMEM, "device batch buffers");
  hipStream_t s = ctx->stream;
  HIP_TRY(hipMemcpyAsync(ctx->b_seqs.p, seqs, n_bytes, hipMemcpyHostToDevice, s));
  HIP_TRY(hipMemcpyAsync(ctx->b_off.p, seq_off, n_reads * 8, hipMemcpyHostToDevice, s));
  HIP_TRY(hipMemcpyAsync(ctx->b_len.p, seq_len, n_reads * 4, hipMemcpyHostToDevice, s));
  KU_TRY(classify_device_impl(ctx, ctx->b_seqs.p, n_bytes, (const uint64_t *)ctx->b_off.p, (const uint32_t *)ctx->b_len.p,
                              n_reads, &o, (uint32_t *)ctx->b_calls.p, (uint32_t *)ctx->b_taxa.p,
                              (uint32_t *)ctx->b_hits.p, s, seq_off, seq_len));
  HIP_TRY(hipMemcpyAsync(calls, ctx->b_calls.p, n_reads * 4, hipMemcpyDeviceToHost, s));
  if (taxa) HIP_TRY(hipMemcpyAsync(taxa, ctx->b_taxa.p, n_bytes * 4, hipMemcpyDeviceToHost, s));
  if (hits) HIP_TRY(hipMemcpyAsync(hits, ctx->b_hits.p, n_reads * 4, hipMemcpyDeviceToHost, s));
  HIP_TRY(hipStreamSynchronize(s));
  return KU_OK;
}

// run-length encode d_taxa into ctx->b_runs and bring calls / hits / (run_off, run_cnt) / the run total to the host
static int rle_and_fetch(ku_ctx *ctx, const uint32_t *d_taxa, const uint64_t *d_off, const uint32_t *d_len, uint64_t n_reads,
                         uint64_t runs_cap, bool quick, uint32_t *calls, uint32_t *hits, uint64_t *run_off,
                         uint32_t *run_cnt, uint64_t *n_runs) {
  hipStream_t s = ctx->stream;
  unsigned long long *d_counter = (unsigned long long *)(ctx->d_scalar + 2);
  if (quick) {  // quick mode stops at the first hits: no per-k-mer codes, no runs
    HIP_TRY(hipMemsetAsync(d_counter, 0, 8, s));
    HIP_TRY(hipMemsetAsync(ctx->b_roff.p, 0, n_reads * 8, s));
    HIP_TRY(hipMemsetAsync(ctx->b_rcnt.p, 0, n_reads * 4, s));
  } else {
    KU_TRY(ku_launch_rle(d_taxa, ctx->m.db.k, d_off, d_len, n_reads, runs_cap /* ~ bases of the batch */, ctx->b_runs.p, runs_cap, d_counter,
                         (uint64_t *)ctx->b_roff.p, (uint32_t *)ctx->b_rcnt.p, ctx->n_cu, s));
  }
  unsigned long long total = 0;
  HIP_TRY(hipMemcpyAsync(&total, d_counter, 8, hipMemcpyDeviceToHost, s));
  HIP_TRY(hipMemcpyAsync(calls, ctx->b_calls.p, n_reads * 4, hipMemcpyDeviceToHost, s));
  if (hits) HIP_TRY(hipMemcpyAsync(hits, ctx->b_hits.p, n_reads * 4, hipMemcpyDeviceToHost, s));
  HIP_TRY(hipMemcpyAsync(run_off, ctx->b_roff.p, n_reads * 8, hipMemcpyDeviceToHost, s));
  HIP_TRY(hipMemcpyAsync(run_cnt, ctx->b_rcnt.p, n_reads * 4, hipMemcpyDeviceToHost, s));
  HIP_TRY(hipStreamSynchronize(s));
  if (total > runs_cap) return fail(KU_EHIP, "run-length encoder overflowed its bound");
  *n_runs = ctx->n_runs = total;
  ctx->fetch_runs_src = ctx->b_runs.p;
  ctx->last_calls_dev = ctx->b_calls.p;
  return KU_OK;
}

// ---- ku_classify_batch_rle through the fused kernel with run-length encoded output (ku_short.hip, OUT >= 1): no
// per-k-mer array, no second kernel; with the sparse-mode emulation on, its fast path.

// The open unit's reads in tail form (bases of each read followed by '\n') go up to the device and through the exact
// evaluation as local unit `unit` of the pass `d`: a count-less lookup gives their slots (the flat kernel; these reads were
// classified, booked and marked when their batch went through the fused kernel), ku_sparse_insert_kernel feeds L / U as
// for any staged batch.  Positions start at 2; *pos_end = the first position the reads behind the tail may use.
static int sparse_tail_insert(ku_ctx *ctx, const KuSparseDev &d, const std::vector<char> &text, const std::vector<uint32_t> &lens,
                              uint32_t unit, hipStream_t s, uint32_t *pos_end) {
  ku_ctx::Sparse &sp = ctx->sp;
  const uint64_t n_reads = lens.size(), n_bytes = text.size();
  if (pos_end) *pos_end = (uint32_t)n_bytes;
  if (n_reads == 0) return KU_OK;
  std::vector<uint64_t> off(n_reads);
  uint64_t at = 0;
  for (uint64_t r = 0; r < n_reads; ++r) { off[r] = at; at += (uint64_t)lens[r] + 1; }
  if (at != n_bytes) return fail(KU_ESTATE, "sparse-mode emulation: the open unit's reads are inconsistent");
  if (sp.t_seqs.reserve(n_bytes + 16) || sp.t_off.reserve(n_reads * 8) || sp.t_len.reserve(n_reads * 4) || sp.t_taxa.reserve((n_bytes + 16) * 4) ||
      sp.t_unit.reserve(n_reads * 4))
    return fail(KU_ENOMEM, "device memory for the open work unit's reads");
  HIP_TRY(hipMemcpyAsync(sp.t_seqs.p, text.data(), n_bytes, hipMemcpyHostToDevice, s));
  HIP_TRY(hipMemcpyAsync(sp.t_off.p, off.data(), n_reads * 8, hipMemcpyHostToDevice, s));
  HIP_TRY(hipMemcpyAsync(sp.t_len.p, lens.data(), n_reads * 4, hipMemcpyHostToDevice, s));
  std::vector<uint32_t> units(n_reads, unit);
  HIP_TRY(hipMemcpyAsync(sp.t_unit.p, units.data(), n_reads * 4, hipMemcpyHostToDevice, s));
  HIP_TRY(hipStreamSynchronize(s));  // (`off`, `units` go out of scope; this path is rare)
  int st = ku_launch_lookup(ctx->m.db, ctx->cnt, (const uint8_t *)sp.t_seqs.p, n_bytes, (uint32_t *)sp.t_taxa.p, /*do_counts=*/false, false, false,
                            ctx->n_cu, s);
  if (st != KU_OK) return fail(st, "lookup kernel launch failed");
  return ku_launch_sparse_insert(d, ctx->m.db.k, (const uint8_t *)sp.t_seqs.p, (const uint64_t *)sp.t_off.p, (const uint32_t *)sp.t_len.p,
                                 (const uint32_t *)sp.t_unit.p, n_reads, (const uint32_t *)sp.t_taxa.p, 0u, ctx->n_cu, s);
}

// a read of the caller's batch joins the open unit's tail
static void sparse_tail_append(ku_ctx::Sparse &sp, const char *seqs, const uint64_t *seq_off, const uint32_t *seq_len, uint64_t r0, uint64_t r1) {
  for (uint64_t r = r0; r < r1; ++r) {
    sp.tail_text.insert(sp.tail_text.end(), seqs + seq_off[r], seqs + seq_off[r] + seq_len[r]);
    sp.tail_text.push_back('\n');
    sp.tail_len.push_back(seq_len[r]);
  }
}

// The open unit changes from tail form into the staged form (its L / U entries in the carry buffers): what a staged batch and
// ku_ctx_sparse_move_open_unit expect.  Every slot that is not dense is tracked, as the staged passes do.
static int sparse_tail_to_carry(ku_ctx *ctx) {
  ku_ctx::Sparse &sp = ctx->sp;
  if (!sp.tail_open) return KU_OK;
  hipStream_t s = ctx->stream;
  KU_TRY(sparse_reserve_global(ctx, sp.tail_text.size(), s));
  KuSparseDev d;
  KU_TRY(sparse_pass_tables(ctx, sp.tail_text.size(), &d, s));
  KU_TRY(sparse_tail_insert(ctx, d, sp.tail_text, sp.tail_len, 0u, s, nullptr));
  KU_TRY(ku_launch_sparse_close(d, 0u, s));  // nothing closes: the largest first positions for the carry
  HIP_TRY(hipMemsetAsync(sp.d_counters + 1, 0, 16, s));
  KU_TRY(ku_launch_sparse_carry_out(d, 0u, (unsigned long long *)sp.carry_l.p, (uint32_t *)sp.carry_u.p, sp.d_counters + 1, sp.cap_carry_l,
                                    sp.cap_carry_u, s));
  unsigned long long c[3] = {0, 0, 0};
  uint32_t err = 0;
  HIP_TRY(hipMemcpyAsync(c, sp.d_counters, 24, hipMemcpyDeviceToHost, s));
  HIP_TRY(hipMemcpyAsync(&err, sp.dev.err, 4, hipMemcpyDeviceToHost, s));
  HIP_TRY(hipStreamSynchronize(s));
  if (err) return fail(KU_ENOMEM, "sparse-mode emulation: a device table is full");
  sp.g_count = c[0];
  sp.n_carry_l = std::min<uint64_t>(c[1], sp.cap_carry_l);
  sp.n_carry_u = std::min<uint64_t>(c[2], sp.cap_carry_u);
  sp.open = true;
  sp.tail_open = false;
  sp.tail_text.clear();
  sp.tail_len.clear();
  return KU_OK;
}

// The open unit in tail form ends here (end of an input file / of the run): it can only have turned a sketch dense if it gave
// it >= 1025 inserts -- then, and only then, the exact evaluation runs over its reads.
static int sparse_tail_close(ku_ctx *ctx) {
  ku_ctx::Sparse &sp = ctx->sp;
  if (!sp.tail_open) return KU_OK;
  hipStream_t s = ctx->stream;
  if (sp.u_flag.reserve(4)) return fail(KU_ENOMEM, "device memory for the work-unit counters");
  HIP_TRY(hipMemsetAsync(sp.u_flag.p, 0, 4, s));
  KU_TRY(ku_launch_sparse_flag_units((const uint32_t *)sp.tail_row.p, ctx->tax.n_slots, ctx->tax.n_slots, sp.dev.dense, (uint8_t *)sp.u_flag.p, s));
  uint32_t flag = 0;
  HIP_TRY(hipMemcpyAsync(&flag, sp.u_flag.p, 4, hipMemcpyDeviceToHost, s));
  HIP_TRY(hipStreamSynchronize(s));
  if (flag & 0xFFu) {
    KU_TRY(sparse_reserve_global(ctx, sp.tail_text.size(), s));
    KuSparseDev d;
    KU_TRY(sparse_pass_tables(ctx, sp.tail_text.size(), &d, s));
    KU_TRY(sparse_tail_insert(ctx, d, sp.tail_text, sp.tail_len, 0u, s, nullptr));
    KU_TRY(ku_launch_sparse_close(d, 1u, s, /*skip_hits=*/true));
    unsigned long long c = 0;
    uint32_t err = 0;
    HIP_TRY(hipMemcpyAsync(&c, sp.d_counters, 8, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipMemcpyAsync(&err, sp.dev.err, 4, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    if (err) return fail(KU_ENOMEM, "sparse-mode emulation: a device table is full");
    sp.g_count = c;
  }
  sp.tail_open = false;
  sp.tail_text.clear();
  sp.tail_len.clear();
  return KU_OK;
}

// KU_RLE_TIMES=1: where the batch calls spend their time on the host, summed over the run, printed when the context goes
static double g_rle_t[10];  // checks, plan + enqueue, waiting for the device in _finish, behind the wait, calls; of the enqueue: [5] buffers + plan ([9]: sparse_reserve_global in it), [6] uploads, [7] launches, [8] copies back + events
static double g_rle_x[6];  // of 'behind the wait': [0] flagging again, [1] exact passes, [2] their number, [3] units they evaluated, [4] reads in them
static double g_rle_kernel_ms = 0;  // HIP events around every batch's kernels (fused kernel + the emulation's flag kernel): the time
                                    // covered by the batches' intervals -- they overlap since the batches' kernels run on two streams --
static double g_rle_kernel_sum_ms = 0, g_rle_cover_end = 0;  // ... their plain sum, and where the covered time ends (ms behind g_rle_ref)
static hipEvent_t g_rle_ref = nullptr;
static unsigned long long g_rle_reads = 0;
static const bool g_rle_times = getenv("KU_RLE_TIMES") != nullptr;
static double rle_now() { timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return (double)t.tv_sec + 1e-9 * (double)t.tv_nsec; }
static void rle_times_print() {
  if (g_rle_times && g_rle_t[4] > 0)
    fprintf(stderr, "ku_classify_batch_rle over %.0f batches: checks %.3f s, plan + enqueue %.3f s, waiting for the device %.3f s, behind the wait %.3f s; "
                    "kernels %.3f ms for %llu reads (HIP events on their streams: the time the batches' intervals cover; their sum is %.3f ms); of the enqueue: buffers + plan %.3f s, uploads %.3f s, launches %.3f s, "
                    "copies back + events %.3f s; of buffers + plan: room in the emulation's run-wide set %.3f s\n",
            g_rle_t[4], g_rle_t[0], g_rle_t[1], g_rle_t[2], g_rle_t[3], g_rle_kernel_ms, g_rle_reads, g_rle_kernel_sum_ms, g_rle_t[5], g_rle_t[6], g_rle_t[7], g_rle_t[8], g_rle_t[9]);
  if (g_rle_times && g_rle_x[2] > 0)
    fprintf(stderr, "ku_classify_batch_rle, behind the wait: flagging again %.3f s, %.0f exact passes over %.0f work units (%.0f reads) %.3f s, of it %.3f s waiting for their kernels\n", g_rle_x[0], g_rle_x[2],
            g_rle_x[3], g_rle_x[4], g_rle_x[1], g_rle_x[5]);
}

// The exact per-unit evaluation of the emulation for the units the fused kernel could not settle by counting: `flagged`
// (ascending unit numbers of the job's batch; flag_all: every slot of the unit is tracked -- a unit in the staged form).
// last_is_open: the last flagged unit stays open behind the batch (staged form: its entries go into the carry buffers).
static int sparse_fast_exact(ku_ctx *ctx, RleJob &j, const std::vector<uint32_t> &flagged, const std::vector<uint8_t> &flag_all,
                             bool last_is_open, hipStream_t s) {
  ku_ctx::Sparse &sp = ctx->sp;
  const uint32_t *h_len = j.h_len;
  const std::vector<uint64_t> &unit_first_read = j.unit_first_read;
  const uint32_t *d_u_cnt = (const uint32_t *)j.u_cnt.p;
  // the batch's runs: its own run array -- or, when that overflowed, the context's buffers, where rle_job_finish's redo (per-k-mer
  // codes + ku_rle_kernel, whose bound cannot overflow) left them; counts, SEEN marks and u_cnt are the fused kernel's either way
  const void *d_runs = j.runs_in_ctx ? ctx->b_runs.p : j.runs.p;
  const uint64_t *d_roff = (const uint64_t *)(j.runs_in_ctx ? ctx->b_roff.p : j.roff.p);
  const uint32_t *d_rcnt = (const uint32_t *)(j.runs_in_ctx ? ctx->b_rcnt.p : j.rcnt.p);
  // unit 0 continues a unit in tail form: its earlier reads are evaluated with it, and come first in the position space
  const bool with_tail = j.cont_tail && !flagged.empty() && flagged[0] == 0 && !j.tail_len.empty();
  const uint64_t tail_bytes = with_tail ? j.tail_text.size() : 0;
  size_t at = 0;
  bool first_pass = true;
  while (at < flagged.size()) {
    // units of this pass: at most 2^25 bases and KU_SPARSE_MAX_UNITS units (the tables of ku_sparse.hip)
    size_t end = at;
    uint64_t bases = first_pass ? tail_bytes : 0, n_list = 0;
    while (end < flagged.size() && end - at < KU_SPARSE_MAX_UNITS) {
      const uint32_t u = flagged[end];
      uint64_t ub = 0;
      for (uint64_t r = unit_first_read[u]; r < unit_first_read[u + 1]; ++r) ub += h_len[r];
      if (end > at && bases + ub > (1ull << 25)) break;
      bases += ub;
      n_list += unit_first_read[u + 1] - unit_first_read[u];
      ++end;
    }
    const bool has_open = last_is_open && end == flagged.size();
    std::vector<uint32_t> list(3 * n_list);
    uint64_t li = 0;
    for (size_t f = at; f < end; ++f) {
      const uint32_t u = flagged[f];
      for (uint64_t r = unit_first_read[u]; r < unit_first_read[u + 1]; ++r, ++li) {
        list[li] = (uint32_t)r;
        list[n_list + li] = (uint32_t)(f - at) | (flag_all[f] ? 0x80000000u : 0u);
        list[2 * n_list + li] = u;
      }
    }
    if (sp.list.reserve(std::max<uint64_t>(n_list, 1) * 12) != KU_OK) return fail(KU_ENOMEM, "device memory for the flagged work units' reads");
    KU_TRY(sparse_reserve_global(ctx, bases + sp.n_carry_l, s));
    KuSparseDev d;  // this pass's view: the run-wide set as it is now, L / U sized for the pass
    KU_TRY(sparse_pass_tables(ctx, bases + sp.n_carry_l + sp.n_carry_u, &d, s));
    if (first_pass && j.cont_carry)  // the unit carried over from the batch before is local unit 0 of the first pass
      KU_TRY(ku_launch_sparse_carry_in(d, (const unsigned long long *)sp.carry_l.p, sp.n_carry_l, (const uint32_t *)sp.carry_u.p, sp.n_carry_u, s));
    uint32_t pos_base = 0;
    if (first_pass && with_tail) KU_TRY(sparse_tail_insert(ctx, d, j.tail_text, j.tail_len, 0u, s, &pos_base));
    if (n_list) HIP_TRY(hipMemcpyAsync(sp.list.p, list.data(), n_list * 12, hipMemcpyHostToDevice, s));
    const uint32_t *dl = (const uint32_t *)sp.list.p;
    KU_TRY(ku_launch_sparse_insert_runs(d, ctx->m.db.k, (const uint8_t *)j.seqs.p, (const uint64_t *)j.off.p, (const uint32_t *)j.len.p,
                                        dl, dl + n_list, dl + 2 * n_list, n_list, d_runs, d_roff, d_rcnt, ctx->d_slot_taxid, ctx->tax.n_slots,
                                        d_u_cnt, ctx->n_cu, s, pos_base));
    const uint32_t n_local = (uint32_t)(end - at);
    // (a unit in the staged form may hold k-mers of a staged batch, which marks nothing in the probe table: its entries all
    // go into the set; the fast path's own units only contribute their misses)
    KU_TRY(ku_launch_sparse_close(d, has_open ? n_local - 1 : n_local, s, /*skip_hits=*/!j.cont_carry));
    if (first_pass && j.cont_carry) sp.n_carry_l = sp.n_carry_u = 0;
    unsigned long long c[3] = {0, 0, 0};
    if (has_open) {
      HIP_TRY(hipMemsetAsync(sp.d_counters + 1, 0, 16, s));
      KU_TRY(ku_launch_sparse_carry_out(d, n_local - 1, (unsigned long long *)sp.carry_l.p, (uint32_t *)sp.carry_u.p, sp.d_counters + 1,
                                        sp.cap_carry_l, sp.cap_carry_u, s));
    }
    HIP_TRY(hipMemcpyAsync(c, sp.d_counters, 24, hipMemcpyDeviceToHost, s));
    const double t_sy0 = g_rle_times ? rle_now() : 0.0;
    HIP_TRY(hipStreamSynchronize(s));  // `list` goes out of scope
    if (g_rle_times) g_rle_x[5] += rle_now() - t_sy0;
    sp.g_count = std::max<uint64_t>(sp.g_count, c[0]);
    if (has_open) {
      sp.n_carry_l = std::min<uint64_t>(c[1], sp.cap_carry_l);
      sp.n_carry_u = std::min<uint64_t>(c[2], sp.cap_carry_u);
    }
    first_pass = false;
    at = end;
  }
  return KU_OK;
}

// may the batch take the fused kernel with run-length encoded output?  (the same conditions as the fused path of
// classify_device_impl, plus what the emulation's fast path needs)
static bool rle_fused_eligible(ku_ctx *ctx, uint32_t flags, uint32_t max_n, uint64_t n_bytes, uint64_t n_reads, bool monotonic) {
  if (getenv("KU_NO_FUSED") || getenv("KU_NO_FUSED_RLE") || !ctx->extra.empty() || ctx->d_exact_set) return false;
  if (flags & (KU_F_QUICK | KU_F_KEEP_SLOTS)) return false;
  const uint32_t short_max = ku_short_max_kmers(ctx->m.db);
  if (!short_max) return false;
  if (max_n > short_max && (max_n > ku_short_max_kmers_windowed(ctx->m.db) || getenv("KU_NO_WINDOWED"))) return false;
  if (n_reads >= (1ull << 32)) return false;
  const bool sparse = ctx->sp.on && !(flags & KU_F_NO_COUNTS);
  if (sparse) {
    const ku_ctx::Sparse &sp = ctx->sp;
    // (positions of the exact evaluation are 32-bit: the batch, behind the reads of an open unit of at most 2^24 nt + one read)
    if (getenv("KU_NO_SPARSE_FAST") || !monotonic || sp.unit_nt == 0 || sp.unit_nt > (1ull << 24) || n_bytes + (1ull << 26) >= (1ull << 32)) return false;
    const uint64_t max_units = n_bytes / sp.unit_nt + 2;
    if (max_units * ctx->tax.n_slots > (1ull << 29)) return false;  // the (unit, slot) counters: at most 2 GiB
  }
  return true;
}

// chunk of the run array a wave claims at a time: large enough for few claims, small enough that the unused tails of the
// last chunks do not dominate a small batch
static uint32_t rle_chunk(uint64_t n_reads, uint64_t total_waves) {
  const uint64_t reads_per_wave = n_reads / std::max<uint64_t>(total_waves, 1);
  return reads_per_wave >= 64 ? 256u : (reads_per_wave >= 16 ? 64u : 16u);
}

extern "C" uint64_t ku_device_rle_runs_cap(const ku_ctx *ctx, uint64_t n_bytes, uint64_t n_reads, uint32_t max_read_len) {
  if (!ctx || !ctx->tax_set) return 0;
  const uint32_t max_n = max_read_len >= ctx->m.db.k ? max_read_len - ctx->m.db.k + 1 : 0;
  const uint64_t waves = ku_short_grid_waves(n_reads, max_n, ctx->n_cu);
  return n_bytes / 6 + 4 * n_reads + waves * rle_chunk(n_reads, waves) + 4096;
}

extern "C" int ku_classify_batch_device_rle(ku_ctx *ctx, const void *d_seqs, uint64_t n_bytes, const uint64_t *d_seq_off,
                                            const uint32_t *d_seq_len, uint64_t n_reads, const ku_opts *opts, uint32_t *d_calls,
                                            ku_run *d_runs, uint64_t runs_cap, uint64_t *d_run_off, uint32_t *d_run_cnt,
                                            uint64_t *d_n_runs, void *stream) {
  KU_TRY(check_ready(ctx));
  if (!d_n_runs || (n_reads && (!d_seqs || !d_seq_off || !d_seq_len || !d_calls || !d_runs || !d_run_off || !d_run_cnt)))
    return fail(KU_EINVAL, "ku_classify_batch_device_rle: null buffer");
  const ku_opts o = opts ? *opts : ku_opts{0, 1, 0, 0};
  hipStream_t s = stream ? (hipStream_t)stream : ctx->stream;
  HIP_TRY(hipMemsetAsync(d_n_runs, 0, 8, s));
  if (n_reads == 0) return KU_OK;
  if (o.max_read_len == 0) return fail(KU_EINVAL, "ku_classify_batch_device_rle: opts->max_read_len (the longest read of the batch) is required");
  const uint32_t max_n = o.max_read_len >= ctx->m.db.k ? o.max_read_len - ctx->m.db.k + 1 : 0;
  if (ctx->sp.on && !(o.flags & KU_F_NO_COUNTS))
    return fail(KU_EUNSUP, "the sparse-mode emulation runs through the host-buffer entry points (it needs the read lengths on the host)");
  if (!store_whole(ctx->m) || !rle_fused_eligible(ctx, o.flags, max_n, n_bytes, n_reads, true))
    return fail(KU_EUNSUP, "ku_classify_batch_device_rle: the fused kernel does not apply to this context / these options (ku_classify_batch_device does)");
  uint64_t ws = 0;
  if (max_n > ku_short_max_kmers(ctx->m.db)) {  // windowed instance: its spill workspace
    ws = ku_short_workspace_bytes(max_n, ctx->tax.n_slots, n_reads, ctx->n_cu);
    if (ws > ctx->b_ws.cap) HIP_TRY(hipStreamSynchronize(s));
    if (ctx->b_ws.reserve(ws)) return fail(KU_ENOMEM, "device memory for the windowed kernel's workspace");
  }
  const uint64_t waves = ku_short_grid_waves(n_reads, max_n, ctx->n_cu);
  KuRunsOut ro{};
  ro.runs = (uint2 *)d_runs;
  ro.counter = (unsigned long long *)d_n_runs;
  ro.cap = runs_cap;
  ro.chunk = rle_chunk(n_reads, waves);
  ro.run_off = d_run_off;
  ro.run_cnt = d_run_cnt;
  // every wave owns its first chunk, the counter starts behind those (as in rle_job_enqueue: no claim storm at the launch's start)
  if (waves * ro.chunk <= runs_cap && waves * ro.chunk < (1ull << 31)) {
    HIP_TRY(hipMemsetD32Async((hipDeviceptr_t)d_n_runs, (int)(waves * ro.chunk), 1, s));
    ro.pre_base1 = 1;
  }
  int st = ku_launch_classify_short(ctx->m.db, ctx->tax, ctx->cnt, (const uint8_t *)d_seqs, n_bytes, d_seq_off, d_seq_len, n_reads, max_n, o.flags,
                                    d_calls, nullptr, nullptr, ctx->b_ws.p, ctx->b_ws.cap, ctx->n_cu, s, &ro, nullptr);
  return st == KU_OK ? KU_OK : fail(st, "fused kernel launch failed");
}

// no batch may be in flight (entry points that read or change what the batches in flight work on)
static int rle_idle(const ku_ctx *ctx, const char *who) {
  if (ctx->rle_in_flight) return fail(KU_ESTATE, std::string(who) + ": batches are in flight (ku_classify_batch_rle_finish first)");
  return KU_OK;
}

// ---- step one: plan the batch, start its upload (in segments, on the copy stream), its kernels and the copies back.
// Nothing here waits for the device.
static int rle_job_enqueue(ku_ctx *ctx, RleJob &j, const char *seqs, uint64_t n_bytes, const uint64_t *seq_off, const uint32_t *seq_len,
                           uint64_t n_reads, const ku_opts &o, uint32_t max_n, bool monotonic, uint32_t *calls, uint32_t *hits,
                           uint64_t *run_off, uint32_t *run_cnt, ku_run *h_runs, uint64_t h_runs_cap) {
  // the batch's kernels: consecutive batches take the two kernel streams in turn (KU_RLE_KERNEL_STREAMS=1: the context's one stream)
  static const bool one_kernel_stream = getenv("KU_RLE_KERNEL_STREAMS") && atoi(getenv("KU_RLE_KERNEL_STREAMS")) == 1;
  hipStream_t s = ctx->stream;
  if (!one_kernel_stream) {
    hipStream_t &ks = ctx->k_streams[(&j - &ctx->rle[0]) & 1];
    if (!ks) HIP_TRY(hipStreamCreateWithFlags(&ks, hipStreamNonBlocking));
    if (!ctx->main_ev) HIP_TRY(hipEventCreateWithFlags(&ctx->main_ev, hipEventDisableTiming));
    s = ks;
  }
  const double t_in = g_rle_times ? rle_now() : 0.0;
  const bool counts = !(o.flags & KU_F_NO_COUNTS);
  const bool sparse = ctx->sp.on && counts;
  ku_ctx::Sparse &sp = ctx->sp;
  // ---- segments of the batch: cut at read boundaries, uploaded one after the other on the copy stream while the
  // compute stream classifies the ones before
  uint64_t n_seg = 1;
  // (16 MiB per segment since round 5 -- was 8: with several batches in flight the overlap of upload and kernels comes from the
  // OTHER batches, and a launch of 120 k reads costs 3.0 us per thousand reads where two of 60 k cost 3.8, ku_short.hip)
  static const uint64_t seg_bytes = (uint64_t)std::max(1, getenv("KU_RLE_SEG_MB") ? atoi(getenv("KU_RLE_SEG_MB")) : 16) << 20;
  if (monotonic && !getenv("KU_NO_H2D_OVERLAP")) n_seg = std::min<uint64_t>(8, std::max<uint64_t>(1, n_bytes / seg_bytes));
  std::vector<uint64_t> seg(n_seg + 1, 0);
  for (uint64_t g = 1; g < n_seg; ++g) {
    const uint64_t target = n_bytes / n_seg * g;
    seg[g] = std::max<uint64_t>(seg[g - 1], (uint64_t)(std::lower_bound(seq_off, seq_off + n_reads, target) - seq_off));
  }
  seg[n_seg] = n_reads;
  uint64_t total_waves = 0, max_seg_reads = 0;
  for (uint64_t g = 0; g < n_seg; ++g) {
    total_waves += ku_short_grid_waves(seg[g + 1] - seg[g], max_n, ctx->n_cu);
    max_seg_reads = std::max(max_seg_reads, seg[g + 1] - seg[g]);
  }
  // a wave claims `chunk` run entries at a time: large enough for few claims, small enough that the unused tails of the
  // last chunks do not dominate a small batch
  const uint32_t chunk = rle_chunk(n_reads, total_waves);
  // room for ~ one run per 6 bases + the chunk tails; a batch that needs more (many taxa per read) is redone through
  // the per-k-mer array (in _finish), whose run-length encoder cannot overflow
  uint64_t runs_cap = n_bytes / 6 + 4 * n_reads + total_waves * chunk + 4096;
  if (const char *e = getenv("KU_RUNS_CAP")) runs_cap = std::max<uint64_t>(1, (uint64_t)atoll(e));  // test hook
  uint64_t ws = 0;
  if (max_n > ku_short_max_kmers(ctx->m.db)) ws = ku_short_workspace_bytes(max_n, ctx->tax.n_slots, max_seg_reads, ctx->n_cu);
  // (the job's buffers are its own and its previous batch is through: growing them needs no synchronisation of ours)
  if (j.seqs.reserve(n_bytes + 16) || j.off.reserve(n_reads * 8) || j.len.reserve(n_reads * 4) || j.calls.reserve(n_reads * 4) ||
      j.runs.reserve(runs_cap * 8) || j.roff.reserve(n_reads * 8) || j.rcnt.reserve(n_reads * 4) || j.ws.reserve(ws))
    return fail(KU_ENOMEM, "device batch buffers");
  if (!ctx->h2d_stream) HIP_TRY(hipStreamCreateWithFlags(&ctx->h2d_stream, hipStreamNonBlocking));
  if (!ctx->d2h_stream) HIP_TRY(hipStreamCreateWithFlags(&ctx->d2h_stream, hipStreamNonBlocking));
  if (!j.done) HIP_TRY(hipEventCreateWithFlags(&j.done, hipEventDisableTiming));
  if (!j.kernels_done) HIP_TRY(hipEventCreateWithFlags(&j.kernels_done, hipEventDisableTiming));
  while (j.seg_events.size() < n_seg) {
    hipEvent_t e;
    HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    j.seg_events.push_back(e);
  }
  // ---- sparse fast path: work-unit plan (the unit closes behind the read that fills it, classify.cpp:510-521)
  j.sparse = sparse;
  j.cont_carry = j.cont_tail = j.open_after = false;
  j.n_units = 0;
  j.kmers = 0;
  j.unit_first_read.clear();
  j.tail_text.clear();
  j.tail_len.clear();
  KuSparseFast sf{};
  uint32_t *h_unit = nullptr;
  uint8_t *h_flag = nullptr;
  if (sparse) {
    if (j.pin_unit.reserve(n_reads * 4)) return fail(KU_ENOMEM, "page-locked memory for the work-unit plan");
    h_unit = (uint32_t *)j.pin_unit.p;
    uint64_t acc = sp.acc_nt;
    uint32_t cur = 0;
    j.cont_carry = sp.open;
    j.cont_tail = sp.tail_open && !sp.open;
    j.unit_first_read.push_back(0);
    for (uint64_t r = 0; r < n_reads; ++r) {
      h_unit[r] = cur;
      acc += seq_len[r];
      if (acc >= sp.unit_nt) { ++cur; acc = 0; j.unit_first_read.push_back(r + 1); }
    }
    j.open_after = acc > 0;
    j.acc_after = acc;
    j.n_units = cur + (j.unit_first_read.back() < n_reads ? 1u : 0u);
    if (j.unit_first_read.back() < n_reads) j.unit_first_read.push_back(n_reads);
    const uint64_t cells = (uint64_t)j.n_units * ctx->tax.n_slots;
    if (j.unit.reserve(n_reads * 4) || j.u_cnt.reserve(std::max<uint64_t>(cells, 1) * 4) || j.u_flag.reserve(((uint64_t)std::max<uint32_t>(j.n_units, 1) + 3) & ~3ull) ||
        sp.tail_row.reserve((size_t)ctx->tax.n_slots * 4))
      return fail(KU_ENOMEM, "device memory for the work-unit counters");
    for (uint64_t r = 0; r < n_reads; ++r) j.kmers += seq_len[r] >= ctx->m.db.k ? seq_len[r] - ctx->m.db.k + 1 : 0;
    uint64_t in_flight_kmers = 0;  // what the batches in flight may still add: the host's count of the set lags behind them
    for (const RleJob &q : ctx->rle) if (q.busy && &q != &j && q.sparse) in_flight_kmers += q.kmers;
    const double t_g0 = g_rle_times ? rle_now() : 0.0;
    KU_TRY(sparse_reserve_global(ctx, j.kmers + in_flight_kmers + sp.n_carry_l, ctx->stream));  // (drains the batches in flight when it grows the set)
    if (g_rle_times) g_rle_t[9] += rle_now() - t_g0;
    sf.g_key = sp.dev.g_key;
    sf.g_mask = sp.dev.g_mask;
    sf.g_count = sp.dev.g_count;
    sf.dense = sp.dev.dense;
    sf.u_cnt = (uint32_t *)j.u_cnt.p;
    sf.err = sp.dev.err;
    sf.n_slots = ctx->tax.n_slots;
    sf.unit_base = 0;
    ctx->m.seen_dirty = true;  // the kernel books the k-mers the database holds by marking their table entries
  }
  if (j.pin.reserve(64 + (size_t)std::max<uint32_t>(j.n_units, 1) + 8)) return fail(KU_ENOMEM, "page-locked memory for the batch totals");
  unsigned long long *h_tot = (unsigned long long *)j.pin.p;
  h_flag = (uint8_t *)j.pin.p + 64;
  if (g_rle_times) g_rle_t[5] += rle_now() - t_in;
  if (s != ctx->stream) {  // whatever was queued on the context's own stream before this batch comes first
    HIP_TRY(hipEventRecord(ctx->main_ev, ctx->stream));
    HIP_TRY(hipStreamWaitEvent(s, ctx->main_ev, 0));
  }
  unsigned long long *d_counter = j.d_counter;
  // (emulation) the per-(unit, slot) insert counts and the unit flags start at zero: one launch
  if (sparse) {
    if (ku_launch_zero3(d_counter, 2, j.u_cnt.p, std::max<uint64_t>((uint64_t)j.n_units * ctx->tax.n_slots, 1), j.u_flag.p,
                        ((uint64_t)std::max<uint32_t>(j.n_units, 1) + 3) / 4, s) != KU_OK)
      return fail(KU_EHIP, "clearing the batch counters failed");
    // (unit 0 continues the open unit: the inserts that unit has had so far join its row BEHIND the kernels, below)
  }
  // the run counter starts behind the chunks the waves own from the start (one per wave of every segment's launch)
  h_tot[3] = total_waves * (unsigned long long)chunk;
  HIP_TRY(hipMemcpyAsync(d_counter, &h_tot[3], 8, hipMemcpyHostToDevice, s));
  KuRunsOut ro{};
  ro.runs = (uint2 *)j.runs.p;
  ro.counter = d_counter;
  ro.cap = runs_cap;
  ro.chunk = chunk;
  if (g_rle_times && !j.t_k0) { HIP_TRY(hipEventCreate(&j.t_k0)); HIP_TRY(hipEventCreate(&j.t_k1)); }
  bool clock_started = false;
  uint64_t waves_before = 0;
  for (uint64_t g = 0; g < n_seg; ++g) {
    const uint64_t a = seg[g], b = seg[g + 1];
    const uint64_t lo = g == 0 ? 0 : seq_off[a], hi = g + 1 == n_seg ? n_bytes : seq_off[b];
    // (always the copy stream: with a batch in flight, this one's upload runs under that one's kernels)
    hipStream_t cs = ctx->h2d_stream;
    const double t_u0 = g_rle_times ? rle_now() : 0.0;
    if (hi > lo) HIP_TRY(hipMemcpyAsync((char *)j.seqs.p + lo, seqs + lo, hi - lo, hipMemcpyHostToDevice, cs));
    if (b > a) {
      HIP_TRY(hipMemcpyAsync((uint64_t *)j.off.p + a, seq_off + a, (b - a) * 8, hipMemcpyHostToDevice, cs));
      HIP_TRY(hipMemcpyAsync((uint32_t *)j.len.p + a, seq_len + a, (b - a) * 4, hipMemcpyHostToDevice, cs));
      if (sparse) HIP_TRY(hipMemcpyAsync((uint32_t *)j.unit.p + a, h_unit + a, (b - a) * 4, hipMemcpyHostToDevice, cs));
    }
    HIP_TRY(hipEventRecord(j.seg_events[g], cs));
    HIP_TRY(hipStreamWaitEvent(s, j.seg_events[g], 0));
    const double t_u1 = g_rle_times ? rle_now() : 0.0;
    if (g_rle_times) g_rle_t[6] += t_u1 - t_u0;
    if (b == a) continue;
    ro.run_off = (uint64_t *)j.roff.p + a;
    ro.run_cnt = (uint32_t *)j.rcnt.p + a;
    ro.pre_base1 = (uint32_t)(1 + waves_before);
    waves_before += ku_short_grid_waves(b - a, max_n, ctx->n_cu);
    sf.unit_of = sparse ? (const uint32_t *)j.unit.p + a : nullptr;
    if (g_rle_times && !clock_started) {  // (behind the first segment's upload)
      if (!g_rle_ref) { HIP_TRY(hipEventCreate(&g_rle_ref)); HIP_TRY(hipEventRecord(g_rle_ref, s)); }
      HIP_TRY(hipEventRecord(j.t_k0, s));
      clock_started = true;
    }
    int st = ku_launch_classify_short(ctx->m.db, ctx->tax, ctx->cnt, (const uint8_t *)j.seqs.p, n_bytes, (const uint64_t *)j.off.p + a,
                                      (const uint32_t *)j.len.p + a, b - a, max_n, o.flags, (uint32_t *)j.calls.p + a, nullptr, nullptr,
                                      j.ws.p, j.ws.cap, ctx->n_cu, s, &ro, sparse ? &sf : nullptr);
    if (st != KU_OK) { (void)hipStreamSynchronize(ctx->h2d_stream); (void)hipStreamSynchronize(s); return fail(st, "fused kernel launch failed"); }
    if (g_rle_times) g_rle_t[7] += rle_now() - t_u1;
  }
  const double t_c0 = g_rle_times ? rle_now() : 0.0;
  if (sparse) {
    // unit 0 continues the open unit: the inserts that unit had before this batch join its row here, behind the kernels -- the
    // batch before this one writes them behind ITS kernels, on the other stream (tail_ready), and only this small step waits
    if (j.cont_tail && j.n_units) {
      if (ctx->tail_ready_set && s != ctx->stream) HIP_TRY(hipStreamWaitEvent(s, ctx->tail_ready, 0));
      KU_TRY(ku_launch_add_u32((uint32_t *)j.u_cnt.p, (const uint32_t *)sp.tail_row.p, ctx->tax.n_slots, s));
    }
    KU_TRY(ku_launch_sparse_flag_units((const uint32_t *)j.u_cnt.p, (uint64_t)j.n_units * ctx->tax.n_slots, ctx->tax.n_slots, sp.dev.dense,
                                       (uint8_t *)j.u_flag.p, s));
    // the unit that stays open (tail form): its insert counts so far
    if (j.open_after && !(j.cont_carry && j.n_units == 1)) {
      HIP_TRY(hipMemcpyAsync(sp.tail_row.p, (const uint32_t *)j.u_cnt.p + (size_t)(j.n_units - 1) * ctx->tax.n_slots, (size_t)ctx->tax.n_slots * 4,
                             hipMemcpyDeviceToDevice, s));
      if (s != ctx->stream) {
        if (!ctx->tail_ready) HIP_TRY(hipEventCreateWithFlags(&ctx->tail_ready, hipEventDisableTiming));
        HIP_TRY(hipEventRecord(ctx->tail_ready, s));
        ctx->tail_ready_set = true;
      }
    }
  }
  if (g_rle_times && clock_started) HIP_TRY(hipEventRecord(j.t_k1, s));
  // ---- the copies back run on a stream of their own, behind this batch's kernels -- not in front of the next batch's
  static const bool own_d2h_stream = !(getenv("KU_RLE_D2H_STREAM") && atoi(getenv("KU_RLE_D2H_STREAM")) == 0);
  hipStream_t ds = own_d2h_stream ? ctx->d2h_stream : s;
  HIP_TRY(hipEventRecord(j.kernels_done, s));
  if (own_d2h_stream) {
    HIP_TRY(hipStreamWaitEvent(ds, j.kernels_done, 0));
  }
  HIP_TRY(hipMemcpyAsync(calls, j.calls.p, n_reads * 4, hipMemcpyDeviceToHost, ds));
  HIP_TRY(hipMemcpyAsync(run_off, j.roff.p, n_reads * 8, hipMemcpyDeviceToHost, ds));
  HIP_TRY(hipMemcpyAsync(run_cnt, j.rcnt.p, n_reads * 4, hipMemcpyDeviceToHost, ds));
  HIP_TRY(hipMemcpyAsync(&h_tot[0], d_counter, 8, hipMemcpyDeviceToHost, ds));
  // the runs themselves, when the caller said where they go: as much of the run array as its buffer holds (the extent in use
  // is only known behind the kernel; _finish tells whether it fitted -- else ku_fetch_runs, into a larger buffer)
  j.runs_copied = h_runs ? std::min<uint64_t>(h_runs_cap, runs_cap) : 0;
  if (j.runs_copied) HIP_TRY(hipMemcpyAsync(h_runs, j.runs.p, j.runs_copied * 8, hipMemcpyDeviceToHost, ds));
  if (sparse) {
    HIP_TRY(hipMemcpyAsync(h_flag, j.u_flag.p, std::max<uint32_t>(j.n_units, 1), hipMemcpyDeviceToHost, ds));
    HIP_TRY(hipMemcpyAsync(&h_tot[1], sp.dev.g_count, 8, hipMemcpyDeviceToHost, ds));
    HIP_TRY(hipMemcpyAsync(&h_tot[2], sp.dev.err, 4, hipMemcpyDeviceToHost, ds));
  }
  HIP_TRY(hipEventRecord(j.done, ds));
  if (g_rle_times) g_rle_t[8] += rle_now() - t_c0;
  // ---- the emulation's state behind this batch (what the next batch's plan starts from)
  if (sparse && j.n_units) {
    const bool whole_batch_one_open_unit = j.n_units == 1 && j.open_after;
    if (j.cont_tail && !whole_batch_one_open_unit) {  // unit 0 closes in this batch: its earlier reads go with the job
      j.tail_text.swap(sp.tail_text);
      j.tail_len.swap(sp.tail_len);
      sp.tail_text.clear();
      sp.tail_len.clear();
    }
    if (j.cont_carry && whole_batch_one_open_unit) {
      // (staged form, still open: stays in the carry buffers -- _finish writes them)
    } else if (j.open_after) {
      if (!(j.cont_tail && whole_batch_one_open_unit)) { sp.tail_text.clear(); sp.tail_len.clear(); }
      sparse_tail_append(sp, seqs, seq_off, seq_len, j.unit_first_read[j.n_units - 1], n_reads);
      sp.tail_open = true;
      sp.open = false;
    } else {
      sp.tail_open = false;
      sp.open = false;
      sp.tail_text.clear();
      sp.tail_len.clear();
    }
    sp.acc_nt = j.acc_after;
  }
  j.n_bytes = n_bytes;
  j.n_reads = n_reads;
  j.runs_cap = runs_cap;
  j.max_n = max_n;
  j.o = o;
  j.h_len = seq_len;
  j.h_calls = calls;
  j.h_hits = hits;
  j.h_roff = run_off;
  j.h_rcnt = run_cnt;
  j.settled = false;
  j.runs_in_ctx = false;
  j.busy = true;
  if (g_rle_times) g_rle_t[1] += rle_now() - t_in;
  return KU_OK;
}

// ---- step two: wait for the batch (one event), settle what the emulation has to settle for it
static int rle_job_finish(ku_ctx *ctx, RleJob &j, uint64_t *n_runs, bool *classified) {
  hipStream_t s = ctx->stream;
  *classified = false;
  ctx->last_runs_copied = 0;
  if (j.settled) {
    j.busy = false;
    *n_runs = ctx->n_runs = j.n_runs;
    *classified = true;
    return KU_OK;
  }
  const double t_w0 = g_rle_times ? rle_now() : 0.0;
  HIP_TRY(hipEventSynchronize(j.done));
  const double t_w1 = g_rle_times ? rle_now() : 0.0;
  if (g_rle_times && j.t_k0 && j.n_reads) {
    float ms = 0;
    float a = 0, b = 0;
    if (g_rle_ref && hipEventElapsedTime(&ms, j.t_k0, j.t_k1) == hipSuccess && hipEventElapsedTime(&a, g_rle_ref, j.t_k0) == hipSuccess &&
        hipEventElapsedTime(&b, g_rle_ref, j.t_k1) == hipSuccess) {
      g_rle_kernel_sum_ms += ms;
      g_rle_kernel_ms += std::max(0.0, (double)b - std::max((double)a, g_rle_cover_end));  // (the batches come in the order of their starts)
      g_rle_cover_end = std::max(g_rle_cover_end, (double)b);
      g_rle_reads += j.n_reads;
    } else (void)hipGetLastError();
  }
  struct Lap { double a, b; ~Lap() { if (g_rle_times) { g_rle_t[2] += b - a; g_rle_t[3] += rle_now() - b; g_rle_t[4] += 1; } } } lap_{t_w0, t_w1};
  j.busy = false;
  ku_ctx::Sparse &sp = ctx->sp;
  const unsigned long long *h_tot = (const unsigned long long *)j.pin.p;
  const uint8_t *h_flag = (const uint8_t *)j.pin.p + 64;
  const unsigned long long total = h_tot[0];
  if (j.h_hits) memset(j.h_hits, 0, j.n_reads * 4);  // "Q:n" is quick mode only
  ctx->last_calls_dev = j.calls.p;
  if (total > j.runs_cap) {
    // the run array was too small for this batch (reads that change taxon every few k-mers): the per-k-mer codes once more
    // without any accounting, through the array parallel to the reads and its own run-length encoder (the context's buffers)
    if (ctx->b_taxa.reserve((j.n_bytes + 16) * 4) || ctx->b_runs.reserve((j.n_bytes + 1) * 8) || ctx->b_roff.reserve(j.n_reads * 8) ||
        ctx->b_rcnt.reserve(j.n_reads * 4) || ctx->b_calls.reserve(j.n_reads * 4))
      return fail(KU_ENOMEM, "device batch buffers");
    uint64_t ws2 = 0;
    if (j.max_n > ku_short_max_kmers(ctx->m.db)) ws2 = ku_short_workspace_bytes(j.max_n, ctx->tax.n_slots, j.n_reads, ctx->n_cu);
    if (ctx->b_ws.reserve(ws2)) return fail(KU_ENOMEM, "device batch buffers");
    int st = ku_launch_classify_short(ctx->m.db, ctx->tax, ctx->cnt, (const uint8_t *)j.seqs.p, j.n_bytes, (const uint64_t *)j.off.p,
                                      (const uint32_t *)j.len.p, j.n_reads, j.max_n, j.o.flags | KU_F_NO_COUNTS, (uint32_t *)ctx->b_calls.p,
                                      (uint32_t *)ctx->b_taxa.p, nullptr, ctx->b_ws.p, ctx->b_ws.cap, ctx->n_cu, s);
    if (st != KU_OK) return fail(st, "fused kernel launch failed");
    KU_TRY(rle_and_fetch(ctx, (const uint32_t *)ctx->b_taxa.p, (const uint64_t *)j.off.p, (const uint32_t *)j.len.p, j.n_reads, j.n_bytes + 1,
                         false, j.h_calls, nullptr, j.h_roff, j.h_rcnt, n_runs));
    j.runs_in_ctx = true;
  } else {
    *n_runs = ctx->n_runs = total;
    ctx->fetch_runs_src = j.runs.p;
    ctx->last_runs_copied = j.runs_copied;
  }
  *classified = true;  // what follows only concerns the emulation's state
  if (j.sparse && sp.on) {
    if ((uint32_t)h_tot[2]) return fail(KU_ENOMEM, "sparse-mode emulation: the run-wide set is full");
    sp.g_count = std::max<uint64_t>(sp.g_count, h_tot[1]);
    // Units the counting could not settle.  A unit that is still open behind the batch waits (tail form: it is looked at
    // when it closes, with everything it got); a unit in the staged form (it came from a staged batch) is always tracked.
    std::vector<uint32_t> flagged;
    std::vector<uint8_t> flag_all;
    const bool carry_stays_open = j.cont_carry && j.n_units == 1 && j.open_after;
    // The flags are from when the batch's kernels ran -- with several batches in flight, before the exact pass of a batch AHEAD
    // of this one turned dense the very sketch that flags these units (the first units of a run: taxon 0's; without this, every
    // unit of the two batches behind went through the exact evaluation for nothing, 25-75 ms per 10 M reads).  Sketches only ever
    // turn dense, so flagging once more with the state as it is now can only take flags away; the counts are complete (the
    // batch's event), the state is at rest (exact passes end synchronised), and a stream of its own does not queue behind the
    // kernels of the batches in flight.
    bool any_flag = false;
    for (uint32_t u = 0; u < j.n_units; ++u) any_flag |= !(u + 1 == j.n_units && j.open_after) && h_flag[u];
    const double t_rf0 = g_rle_times ? rle_now() : 0.0;
    if (any_flag && !getenv("KU_NO_REFLAG")) {  // (test hook: the flags as the kernels left them)
      // (its own stream: d2h_stream holds the waits for the kernels and the copies back of the batches in flight)
      if (!ctx->fetch_stream) HIP_TRY(hipStreamCreateWithFlags(&ctx->fetch_stream, hipStreamNonBlocking));
      hipStream_t fs = ctx->fetch_stream;
      const size_t fb = ((size_t)j.n_units + 3) & ~(size_t)3;
      HIP_TRY(hipMemsetAsync(j.u_flag.p, 0, fb, fs));
      KU_TRY(ku_launch_sparse_flag_units((const uint32_t *)j.u_cnt.p, (uint64_t)j.n_units * ctx->tax.n_slots, ctx->tax.n_slots, sp.dev.dense,
                                         (uint8_t *)j.u_flag.p, fs));
      HIP_TRY(hipMemcpyAsync((uint8_t *)j.pin.p + 64, j.u_flag.p, j.n_units, hipMemcpyDeviceToHost, fs));
      HIP_TRY(hipStreamSynchronize(fs));
    }
    for (uint32_t u = 0; u < j.n_units; ++u) {
      const bool open = u + 1 == j.n_units && j.open_after;
      if (u == 0 && j.cont_carry) { flagged.push_back(u); flag_all.push_back(1); }
      else if (!open && h_flag[u]) { flagged.push_back(u); flag_all.push_back(0); }
    }
    const double t_ex0 = g_rle_times ? rle_now() : 0.0;
    if (g_rle_times) g_rle_x[0] += t_ex0 - t_rf0;
    if (!flagged.empty()) KU_TRY(sparse_fast_exact(ctx, j, flagged, flag_all, carry_stays_open, s));
    if (g_rle_times && !flagged.empty()) {
      g_rle_x[1] += rle_now() - t_ex0;
      g_rle_x[2] += 1;
      g_rle_x[3] += (double)flagged.size();
      for (uint32_t u : flagged) g_rle_x[4] += (double)(j.unit_first_read[u + 1] - j.unit_first_read[u]);
    }
  }
  return KU_OK;
}

static int rle_drain_kernels(ku_ctx *ctx) {
  for (RleJob &q : ctx->rle)
    if (q.busy && q.kernels_done) HIP_TRY(hipEventSynchronize(q.kernels_done));
  return KU_OK;
}

// which of the two jobs takes the next batch / is the oldest in flight
static RleJob &rle_next_job(ku_ctx *ctx) {
  RleJob &j = ctx->rle[(ctx->rle_head + ctx->rle_in_flight) % KU_RLE_MAX_IN_FLIGHT];
  static const int counter_at[KU_RLE_MAX_IN_FLIGHT] = {26, 20, 22, 24};  // (dwords of the context's 32 scalars nobody else uses: dword 2 is
                                                                          // rle_and_fetch's counter, which an overflow redo launches on while batches are in flight)
  if (!j.d_counter) j.d_counter = (unsigned long long *)(ctx->d_scalar + counter_at[&j - &ctx->rle[0]]);
  return j;
}

static int rle_check_batch(const char *seqs, uint64_t n_bytes, const uint64_t *seq_off, const uint32_t *seq_len, uint64_t n_reads,
                           uint32_t *calls, uint64_t *run_off, uint32_t *run_cnt, ku_opts &o, bool &monotonic) {
  if ((n_bytes && !seqs) || (n_reads && (!seq_off || !seq_len || !calls || !run_off || !run_cnt)))
    return fail(KU_EINVAL, "ku_classify_batch_rle: null buffer");
  o.flags &= ~KU_F_KEEP_SLOTS;
  const double t_chk = g_rle_times ? rle_now() : 0.0;
  if (o.max_read_len == 0) for (uint64_t i = 0; i < n_reads; ++i) o.max_read_len = std::max(o.max_read_len, seq_len[i]);
  monotonic = true;
  for (uint64_t i = 0; i < n_reads; ++i) {
    if (seq_off[i] + seq_len[i] > n_bytes) return fail(KU_EINVAL, "read " + std::to_string(i) + " exceeds the sequence buffer");
    if (i && seq_off[i] < seq_off[i - 1] + seq_len[i - 1]) monotonic = false;
  }
  if (g_rle_times) g_rle_t[0] += rle_now() - t_chk;
  return KU_OK;
}

// the one-step paths (quick mode, several databases, sorted layout, shards, reads beyond 65535 k-mers, ...): through the
// context's own buffers, synchronously
static int rle_staged_batch(ku_ctx *ctx, const char *seqs, uint64_t n_bytes, const uint64_t *seq_off, const uint32_t *seq_len, uint64_t n_reads,
                            const ku_opts &o, uint32_t *calls, uint32_t *hits, uint64_t *run_off, uint32_t *run_cnt, uint64_t *n_runs) {
  // a run needs at least one k-mer, so n_bytes bounds the number of runs: the device side cannot overflow
  const uint64_t runs_cap = n_bytes + 1;
  if (ctx->b_seqs.reserve(n_bytes + 16) || ctx->b_off.reserve(n_reads * 8) || ctx->b_len.reserve(n_reads * 4) ||
      ctx->b_calls.reserve(n_reads * 4) || ctx->b_taxa.reserve((n_bytes + 16) * 4) || ctx->b_hits.reserve(n_reads * 4) ||
      ctx->b_runs.reserve(runs_cap * 8) || ctx->b_roff.reserve(n_reads * 8) || ctx->b_rcnt.reserve(n_reads * 4))
    return fail(KU_ENOMEM, "device batch buffers");
  hipStream_t s = ctx->stream;
  HIP_TRY(hipMemcpyAsync(ctx->b_seqs.p, seqs, n_bytes, hipMemcpyHostToDevice, s));
  HIP_TRY(hipMemcpyAsync(ctx->b_off.p, seq_off, n_reads * 8, hipMemcpyHostToDevice, s));
  HIP_TRY(hipMemcpyAsync(ctx->b_len.p, seq_len, n_reads * 4, hipMemcpyHostToDevice, s));
  KU_TRY(classify_device_impl(ctx, ctx->b_seqs.p, n_bytes, (const uint64_t *)ctx->b_off.p, (const uint32_t *)ctx->b_len.p,
                              n_reads, &o, (uint32_t *)ctx->b_calls.p, (uint32_t *)ctx->b_taxa.p,
                              (uint32_t *)ctx->b_hits.p, s, seq_off, seq_len));
  return rle_and_fetch(ctx, (const uint32_t *)ctx->b_taxa.p, (const uint64_t *)ctx->b_off.p, (const uint32_t *)ctx->b_len.p, n_reads,
                       runs_cap, (o.flags & KU_F_QUICK) != 0, calls, hits, run_off, run_cnt, n_runs);
}

// The buffers of `n_jobs` batches of up to n_bytes / n_reads ahead of the first batch (device memory, page-locked scratch,
// streams, events): what _enqueue would otherwise set up inside the caller's timing window, a few milliseconds per job.
extern "C" int ku_classify_batch_rle_reserve(ku_ctx *ctx, uint64_t n_bytes, uint64_t n_reads, uint32_t max_read_len, uint32_t n_jobs) {
  KU_TRY(check_ready(ctx));
  KU_TRY(rle_idle(ctx, "ku_classify_batch_rle_reserve"));
  const uint32_t k = ctx->m.db.k;
  const uint32_t max_n = max_read_len >= k ? max_read_len - k + 1 : 0;
  if (!ku_short_max_kmers(ctx->m.db)) return KU_OK;  // (the fused kernel does not apply: the one-step paths use the context's own buffers)
  if (!ctx->h2d_stream) HIP_TRY(hipStreamCreateWithFlags(&ctx->h2d_stream, hipStreamNonBlocking));
  if (!ctx->d2h_stream) HIP_TRY(hipStreamCreateWithFlags(&ctx->d2h_stream, hipStreamNonBlocking));
  if (!ctx->fetch_stream) HIP_TRY(hipStreamCreateWithFlags(&ctx->fetch_stream, hipStreamNonBlocking));
  const uint64_t waves = ku_short_grid_waves(n_reads, max_n, ctx->n_cu);
  const uint64_t runs_cap = n_bytes / 6 + 4 * n_reads + waves * rle_chunk(n_reads, waves) + 4096;
  uint64_t ws = 0;
  if (max_n > ku_short_max_kmers(ctx->m.db) && max_n <= ku_short_max_kmers_windowed(ctx->m.db)) ws = ku_short_workspace_bytes(max_n, ctx->tax.n_slots, n_reads, ctx->n_cu);
  const bool sparse = ctx->sp.on && ctx->sp.unit_nt;
  const uint64_t n_units = sparse ? n_bytes / ctx->sp.unit_nt + 2 : 0;
  for (uint32_t q = 0; q < std::min<uint32_t>(n_jobs, KU_RLE_MAX_IN_FLIGHT); ++q) {
    RleJob &j = ctx->rle[q];
    if (j.seqs.reserve(n_bytes + 16) || j.off.reserve(n_reads * 8) || j.len.reserve(n_reads * 4) || j.calls.reserve(n_reads * 4) ||
        j.runs.reserve(runs_cap * 8) || j.roff.reserve(n_reads * 8) || j.rcnt.reserve(n_reads * 4) || j.ws.reserve(ws) ||
        j.pin.reserve(64 + (size_t)n_units + 64))
      return fail(KU_ENOMEM, "device batch buffers");
    if (sparse && (j.pin_unit.reserve(n_reads * 4) || j.unit.reserve(n_reads * 4) || j.u_cnt.reserve(std::max<uint64_t>(n_units * ctx->tax.n_slots, 1) * 4) ||
                   j.u_flag.reserve((n_units + 3) & ~3ull)))
      return fail(KU_ENOMEM, "device memory for the work-unit counters");
    if (!j.done) HIP_TRY(hipEventCreateWithFlags(&j.done, hipEventDisableTiming));
    if (!j.kernels_done) HIP_TRY(hipEventCreateWithFlags(&j.kernels_done, hipEventDisableTiming));
    while (j.seg_events.size() < 2) {
      hipEvent_t e;
      HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
      j.seg_events.push_back(e);
    }
  }
  if (sparse && ctx->sp.tail_row.reserve((size_t)ctx->tax.n_slots * 4)) return fail(KU_ENOMEM, "device memory for the work-unit counters");
  // One read through the whole path, count-less (no state changes): the fused kernel's code object is loaded, the copy streams get
  // their queues, the events exist -- here rather than under the caller's first batch, whose enqueue took 22 ms of a 10 M-read
  // `classify` run's 0.19 s window (profiles/r05_e2e_sweep.log, the pipeline trace).  Failure is harmless.
  if (!getenv("KU_NO_WARMUP")) {
    // A dozen batches of the caller's size class through the whole path, count-less (no state changes), three in flight, from and
    // into page-locked memory like the executable's.  What a run's first batches otherwise pay for, as the pipeline trace of
    // `classify` showed it one by one (profiles/r05_e2e_sweep.log): the fused kernel's code object; the copy streams' queues (the
    // first copy of some hundred KB from the device into page-locked memory on a stream: 23.5 ms -- 4-byte copies go another
    // way); each job's events and scratch; scratch memory for the counting instances (they spill a few bytes per lane); and a
    // 13 ms ioctl of the runtime's under the first enqueue that follows a finished batch.  Failure is harmless.
    (void)ku_launch_warm_scratch(ctx->n_cu, ctx->stream);
    double keep_t[10], keep_x[6];  // (KU_RLE_TIMES: the warm-up's batches are none of the caller's)
    memcpy(keep_t, g_rle_t, sizeof keep_t);
    memcpy(keep_x, g_rle_x, sizeof keep_x);
    const double keep_ms = g_rle_kernel_ms, keep_sum = g_rle_kernel_sum_ms;
    const unsigned long long keep_reads = g_rle_reads;
    const uint64_t wn = (std::min<uint64_t>(std::max<uint64_t>(n_reads, 1), 65536) + 1) & ~1ull, stride = 101;  // (even: the arrays behind stay 8-byte aligned)
    const size_t per_slot = (size_t)wn * (4 + 4 + 4 + 8) + (size_t)wn * 8 * 8;
    PinBuf w;
    if (w.reserve((size_t)wn * (stride + 12) + KU_RLE_MAX_IN_FLIGHT * per_slot + 4096) == 0) {
      memset(w.p, 0, w.cap);
      char *text = (char *)w.p;
      uint64_t *w_off = (uint64_t *)(text + ((wn * stride + 63) & ~63ull));
      uint32_t *w_len = (uint32_t *)(w_off + wn);
      char *slots = (char *)(w_len + wn);
      for (uint64_t r = 0; r < wn; ++r) {
        char *t = text + r * stride;
        for (int i = 0; i < 100; ++i) t[i] = "ACGTTGCAAGCTTCGA"[(i * 7 + i / 16 + r) & 15];
        t[100] = '\n';
        w_off[r] = r * stride;
        w_len[r] = 100;
      }
      const ku_opts wo = {KU_F_NO_COUNTS, 1, 100, 0};
      uint64_t w_runs = 0;
      int flying = 0;
      for (int rep = 0; rep < 12; ++rep) {
        if (flying == 3) { (void)ku_classify_batch_rle_finish(ctx, &w_runs); --flying; }
        char *sl = slots + (size_t)(rep % KU_RLE_MAX_IN_FLIGHT) * per_slot;
        uint64_t *roff = (uint64_t *)sl;
        uint32_t *calls = (uint32_t *)(roff + wn), *hits = calls + wn, *rcnt = hits + wn;
        ku_run *runs = (ku_run *)(rcnt + wn);
        if (ku_classify_batch_rle_enqueue(ctx, text, wn * stride, w_off, w_len, wn, &wo, calls, hits, roff, rcnt, runs, wn * 8) == KU_OK) ++flying;
      }
      while (flying-- > 0) (void)ku_classify_batch_rle_finish(ctx, &w_runs);
      if (ctx->fetch_stream) {  // (ku_fetch_runs' stream)
        (void)hipMemcpyAsync(w.p, ctx->rle[0].seqs.p, std::min<size_t>(ctx->rle[0].seqs.cap, 1u << 20), hipMemcpyDeviceToHost, ctx->fetch_stream);
        (void)hipStreamSynchronize(ctx->fetch_stream);
      }
      (void)hipDeviceSynchronize();
      w.release();
    }
    (void)hipGetLastError();
    memcpy(g_rle_t, keep_t, sizeof keep_t);
    memcpy(g_rle_x, keep_x, sizeof keep_x);
    g_rle_kernel_ms = keep_ms;
    g_rle_kernel_sum_ms = keep_sum;
    g_rle_reads = keep_reads;
  }
  return KU_OK;
}

extern "C" int ku_classify_batch_rle_in_flight(const ku_ctx *ctx) { return ctx ? ctx->rle_in_flight : 0; }
extern "C" uint64_t ku_classify_batch_rle_copied(const ku_ctx *ctx) { return ctx ? ctx->last_runs_copied : 0; }

extern "C" int ku_classify_batch_rle_enqueue(ku_ctx *ctx, const char *seqs, uint64_t n_bytes, const uint64_t *seq_off,
                                             const uint32_t *seq_len, uint64_t n_reads, const ku_opts *opts, uint32_t *calls,
                                             uint32_t *hits, uint64_t *run_off, uint32_t *run_cnt, ku_run *runs, uint64_t runs_cap) {
  KU_TRY(check_ready(ctx));
  if (ctx->rle_in_flight >= KU_RLE_MAX_IN_FLIGHT)
    return fail(KU_ESTATE, "ku_classify_batch_rle_enqueue: " + std::to_string(KU_RLE_MAX_IN_FLIGHT) + " batches are in flight (ku_classify_batch_rle_finish first)");
  ku_opts o = opts ? *opts : ku_opts{0, 1, 0, 0};
  bool monotonic = true;
  KU_TRY(rle_check_batch(seqs, n_bytes, seq_off, seq_len, n_reads, calls, run_off, run_cnt, o, monotonic));
  RleJob &j = rle_next_job(ctx);
  const uint32_t max_n = o.max_read_len >= ctx->m.db.k ? o.max_read_len - ctx->m.db.k + 1 : 0;
  // a unit in the staged form (carry buffers) is settled batch by batch, synchronously: such a batch goes in one step, too
  const bool in_steps = n_reads && rle_fused_eligible(ctx, o.flags, max_n, n_bytes, n_reads, monotonic) &&
                        !(ctx->sp.on && !(o.flags & KU_F_NO_COUNTS) && ctx->sp.open);
  if (in_steps) {
    int st = rle_job_enqueue(ctx, j, seqs, n_bytes, seq_off, seq_len, n_reads, o, max_n, monotonic, calls, hits, run_off, run_cnt, runs, runs_cap);
    if (st == KU_ENOMEM && ctx->sp.on && !(o.flags & KU_F_NO_COUNTS)) {
      // no room for the emulation's tables: the classification itself does not depend on them (see classify_device_impl):
      // the run goes on with the dense registers alone; nothing of this batch had been started
      (void)hipStreamSynchronize(ctx->stream);
      for (hipStream_t ks : ctx->k_streams) if (ks) (void)hipStreamSynchronize(ks);
      if (ctx->d2h_stream) (void)hipStreamSynchronize(ctx->d2h_stream);
      (void)hipGetLastError();
      ctx_free_sparse(ctx);
      ctx->sp.gave_up = true;
      st = rle_job_enqueue(ctx, j, seqs, n_bytes, seq_off, seq_len, n_reads, o, max_n, monotonic, calls, hits, run_off, run_cnt, runs, runs_cap);
    }
    KU_TRY(st);
    ++ctx->rle_in_flight;
    return KU_OK;
  }
  if (ctx->rle_in_flight) return fail(KU_ESTATE, "ku_classify_batch_rle_enqueue: this batch takes a path that cannot overlap with the batch in flight "
                                                 "(ku_classify_batch_rle_finish first, then enqueue it again)");
  // classified here and now; _finish hands the totals over
  uint64_t nr = 0;
  ctx->n_runs = 0;
  if (n_reads) {
    if (rle_fused_eligible(ctx, o.flags, max_n, n_bytes, n_reads, monotonic)) {  // (fused, but a unit in the staged form is open)
      int st = rle_job_enqueue(ctx, j, seqs, n_bytes, seq_off, seq_len, n_reads, o, max_n, monotonic, calls, hits, run_off, run_cnt, runs, runs_cap);
      bool classified = false;
      if (st == KU_OK) st = rle_job_finish(ctx, j, &nr, &classified);
      j.busy = false;
      if (st == KU_ENOMEM && ctx->sp.on && !(o.flags & KU_F_NO_COUNTS)) {
        (void)hipStreamSynchronize(ctx->stream);
        (void)hipGetLastError();
        ctx_free_sparse(ctx);
        ctx->sp.gave_up = true;
        if (!classified) {
          st = rle_job_enqueue(ctx, j, seqs, n_bytes, seq_off, seq_len, n_reads, o, max_n, monotonic, calls, hits, run_off, run_cnt, runs, runs_cap);
          if (st == KU_OK) st = rle_job_finish(ctx, j, &nr, &classified);
          j.busy = false;
        } else st = KU_OK;
      }
      KU_TRY(st);
    } else {
      KU_TRY(rle_staged_batch(ctx, seqs, n_bytes, seq_off, seq_len, n_reads, o, calls, hits, run_off, run_cnt, &nr));
      j.runs_in_ctx = true;
    }
  }
  j.settled = true;
  j.busy = true;
  j.n_runs = nr;
  j.runs_copied = 0;
  ++ctx->rle_in_flight;
  return KU_OK;
}

extern "C" int ku_classify_batch_rle_finish(ku_ctx *ctx, uint64_t *n_runs) {
  if (!ctx || !n_runs) return fail(KU_EINVAL, "ku_classify_batch_rle_finish: null argument");
  *n_runs = 0;
  if (!ctx->rle_in_flight) return fail(KU_ESTATE, "ku_classify_batch_rle_finish: no batch is in flight");
  KU_TRY(ctx_activate(ctx));
  RleJob &j = ctx->rle[ctx->rle_head];
  ctx->rle_head = (ctx->rle_head + 1) % KU_RLE_MAX_IN_FLIGHT;
  --ctx->rle_in_flight;
  bool classified = false;
  int st = rle_job_finish(ctx, j, n_runs, &classified);
  j.busy = false;
  if (st == KU_ENOMEM && classified && ctx->sp.on) {
    // the emulation ran out of room behind the classification: it is given up, the run goes on (ku_ctx_sparse_state says 2)
    (void)hipStreamSynchronize(ctx->stream);
    for (hipStream_t ks : ctx->k_streams) if (ks) (void)hipStreamSynchronize(ks);
    if (ctx->d2h_stream) (void)hipStreamSynchronize(ctx->d2h_stream);
    (void)hipGetLastError();
    ctx_free_sparse(ctx);
    ctx->sp.gave_up = true;
    st = KU_OK;
  }
  return st;
}

extern "C" int ku_classify_batch_rle(ku_ctx *ctx, const char *seqs, uint64_t n_bytes, const uint64_t *seq_off,
                                     const uint32_t *seq_len, uint64_t n_reads, const ku_opts *opts, uint32_t *calls,
                                     uint32_t *hits, uint64_t *run_off, uint32_t *run_cnt, uint64_t *n_runs) {
  if (!n_runs) return fail(KU_EINVAL, "ku_classify_batch_rle: null buffer");
  *n_runs = 0;
  KU_TRY(check_ready(ctx));
  KU_TRY(rle_idle(ctx, "ku_classify_batch_rle"));
  KU_TRY(ku_classify_batch_rle_enqueue(ctx, seqs, n_bytes, seq_off, seq_len, n_reads, opts, calls, hits, run_off, run_cnt, nullptr, 0));
  return ku_classify_batch_rle_finish(ctx, n_runs);
}

extern "C" int ku_ctx_replace_calls(ku_ctx *ctx, const uint32_t *new_calls, uint64_t n_reads, uint64_t *n_dropped) {
  KU_TRY(check_ready(ctx));
  if (n_dropped) *n_dropped = 0;
  if (n_reads == 0) return KU_OK;
  if (!new_calls) return fail(KU_EINVAL, "ku_ctx_replace_calls: null argument");
  KU_TRY(rle_idle(ctx, "ku_ctx_replace_calls"));
  if (!ctx->last_calls_dev) return fail(KU_ESTATE, "ku_ctx_replace_calls: the context holds no batch");
  if (ctx->b_hits.reserve(n_reads * 4) != KU_OK) return fail(KU_ENOMEM, "device batch buffers");
  hipStream_t s = ctx->stream;
  unsigned long long *d_dropped = (unsigned long long *)(ctx->d_scalar + 16);
  HIP_TRY(hipMemsetAsync(d_dropped, 0, 8, s));
  HIP_TRY(hipMemcpyAsync(ctx->b_hits.p, new_calls, n_reads * 4, hipMemcpyHostToDevice, s));
  KU_TRY(ku_launch_replace_calls((const uint32_t *)ctx->last_calls_dev, (const uint32_t *)ctx->b_hits.p, n_reads, ctx->d_node_taxid, ctx->tax.n_nodes,
                                 ctx->cnt.n_reads, d_dropped, s));
  HIP_TRY(hipMemcpyAsync((void *)ctx->last_calls_dev, ctx->b_hits.p, n_reads * 4, hipMemcpyDeviceToDevice, s));  // a second replacement starts from these
  unsigned long long dropped = 0;
  HIP_TRY(hipMemcpyAsync(&dropped, d_dropped, 8, hipMemcpyDeviceToHost, s));
  HIP_TRY(hipStreamSynchronize(s));
  if (n_dropped) *n_dropped = dropped;
  return KU_OK;
}

extern "C" int ku_fetch_runs(ku_ctx *ctx, ku_run *runs, uint64_t n_runs) {
  if (!ctx) return fail(KU_EINVAL, "ku_fetch_runs: null context");
  if (n_runs > ctx->n_runs) return fail(KU_EINVAL, "ku_fetch_runs: the last batch holds " + std::to_string(ctx->n_runs) + " runs");
  if (n_runs == 0) return KU_OK;
  if (!runs) return fail(KU_EINVAL, "ku_fetch_runs: null buffer");
  if (!ctx->fetch_runs_src) return fail(KU_ESTATE, "ku_fetch_runs: no batch was classified");
  // (a stream of its own: the copy queues neither behind the kernels of the batch in flight nor behind its copies back,
  // which wait for those kernels)
  KU_TRY(ctx_activate(ctx));
  if (!ctx->fetch_stream) HIP_TRY(hipStreamCreateWithFlags(&ctx->fetch_stream, hipStreamNonBlocking));
  HIP_TRY(hipMemcpyAsync(runs, ctx->fetch_runs_src, n_runs * 8, hipMemcpyDeviceToHost, ctx->fetch_stream));
  HIP_TRY(hipStreamSynchronize(ctx->fetch_stream));
  return KU_OK;
}

// ---------------------------------------------------------------------------- out-of-core run
extern "C" int ku_ctx_prefetch_shard(ku_ctx *ctx, const ku_db *db, uint64_t bin_lo, uint64_t bin_hi) {
  // Safe to call from a second host thread while the first one runs lookups on the resident shard: it works on its own
  // stream, its own scratch and its own store and only reads the (frozen) slot table of the context.
  if (!ctx || !db) return fail(KU_EINVAL, "ku_ctx_prefetch_shard: null argument");
  if (!ctx->db_loaded || !ctx->tax_set) return fail(KU_ESTATE, "ku_ctx_prefetch_shard: load a shard and the taxonomy first");
  if (bin_lo > bin_hi || bin_hi > db->info.n_bins) return fail(KU_EINVAL, "bin range out of bounds");
  if (!ctx->extra.empty()) return fail(KU_EUNSUP, "chunked runs use one database (as the reference's: classify.cpp:639)");
  if (db->info.k != ctx->m.db.k) return fail(KU_EINVAL, "ku_ctx_prefetch_shard: k differs from the resident shard's");
  KU_TRY(ctx_activate(ctx));
  ku_ctx::Prefetch &pf = ctx->pf;
  if (pf.valid) { store_free(pf.store); pf.valid = false; }
  if (!pf.stream) HIP_TRY(hipStreamCreateWithFlags(&pf.stream, hipStreamNonBlocking));
  if (!pf.d_scalar) HIP_TRY(hipMalloc((void **)&pf.d_scalar, 64));
  int st = store_upload(ctx, pf.store, db, bin_lo, bin_hi, /*scan_values=*/false, pf.stream);
  if (st == KU_OK) {
    pf.store.hash_layout = ctx->hash_layout;
    st = store_finalize(ctx, pf.store, pf.stream, pf.d_scalar);
    if (st == KU_EDATA) st = fail(KU_EINVAL, "ku_ctx_prefetch_shard: the slot table does not cover this shard's values "
                                             "(pass ku_db_values() of the whole database to ku_ctx_set_taxonomy)");
  }
  if (st != KU_OK) { store_free(pf.store); return st; }
  pf.db = db;
  pf.bin_lo = bin_lo;
  pf.bin_hi = bin_hi;
  pf.valid = true;
  return KU_OK;
}

extern "C" int ku_ctx_swap_shard(ku_ctx *ctx, const ku_db *db, uint64_t bin_lo, uint64_t bin_hi) {
  KU_TRY(check_ready(ctx));
  if (!db) return fail(KU_EINVAL, "ku_ctx_swap_shard: null argument");
  if (bin_lo > bin_hi || bin_hi > db->info.n_bins) return fail(KU_EINVAL, "bin range out of bounds");
  if (!ctx->extra.empty()) return fail(KU_EUNSUP, "chunked runs use one database (as the reference's: classify.cpp:639)");
  if (db->info.k != ctx->m.db.k) return fail(KU_EINVAL, "ku_ctx_swap_shard: k differs from the resident shard's");
  HIP_TRY(hipStreamSynchronize(ctx->stream));
  KU_TRY(ctx_seen_harvest(ctx));  // (marks of a fast-path run on the table that is about to go)
  ctx_drop_count_cache(ctx);
  if (ctx->pf.valid && ctx->pf.db == db && ctx->pf.bin_lo == bin_lo && ctx->pf.bin_hi == bin_hi) {
    // the chunk was prefetched (ku_ctx_prefetch_shard): it only has to change places with the resident one
    store_free(ctx->m);
    ctx->m = ctx->pf.store;
    ctx->pf.store = DbStore{};
    ctx->pf.valid = false;
    return KU_OK;
  }
  store_free(ctx->m);
  ctx->db_loaded = false;
  KU_TRY(store_upload(ctx, ctx->m, db, bin_lo, bin_hi, /*scan_values=*/false));
  int st = store_finalize(ctx, ctx->m);
  if (st == KU_EDATA) return fail(KU_EINVAL, "ku_ctx_swap_shard: the slot table does not cover this shard's values "
                                             "(pass ku_db_values() of the whole database to ku_ctx_set_taxonomy)");
  KU_TRY(st);
  ctx->db_loaded = true;
  return KU_OK;
}

extern "C" int ku_ctx_mem_info(ku_ctx *ctx, uint64_t *free_bytes, uint64_t *total_bytes) {
  if (!ctx) return fail(KU_EINVAL, "null context");
  KU_TRY(ctx_activate(ctx));
  size_t f = 0, t = 0;
  HIP_TRY(hipMemGetInfo(&f, &t));
  if (free_bytes) *free_bytes = f;
  if (total_bytes) *total_bytes = t;
  return KU_OK;
}

struct ku_batch {
  ku_ctx *ctx = nullptr;
  uint64_t n_bytes = 0, n_reads = 0;
  uint32_t max_len = 0;
  bool finished = false;  // ku_batch_finish translated the slots to taxids in place: no further passes
  void *d_seqs = nullptr;
  uint64_t *d_off = nullptr;
  uint32_t *d_len = nullptr, *d_taxa = nullptr;
  std::vector<uint64_t> h_off;  // host copies for the sparse-mode emulation's work-unit plan
  std::vector<uint32_t> h_len;
};

extern "C" void ku_batch_destroy(ku_batch *b) {
  if (!b) return;
  if (b->ctx) (void)hipSetDevice(b->ctx->device);
  for (void *p : {b->d_seqs, (void *)b->d_off, (void *)b->d_len, (void *)b->d_taxa})
    if (p) (void)hipFree(p);
  delete b;
}

extern "C" int ku_batch_create(ku_ctx *ctx, const char *seqs, uint64_t n_bytes, const uint64_t *seq_off,
                               const uint32_t *seq_len, uint64_t n_reads, ku_batch **out) {
  if (!ctx || !out || (n_bytes && !seqs) || (n_reads && (!seq_off || !seq_len))) return fail(KU_EINVAL, "ku_batch_create: null argument");
  *out = nullptr;
  KU_TRY(ctx_activate(ctx));
  uint32_t max_len = 0;
  for (uint64_t i = 0; i < n_reads; ++i) {
    if (seq_off[i] + seq_len[i] > n_bytes) return fail(KU_EINVAL, "read " + std::to_string(i) + " exceeds the sequence buffer");
    max_len = std::max(max_len, seq_len[i]);
  }
  ku_batch *b = new ku_batch();
  b->ctx = ctx; b->n_bytes = n_bytes; b->n_reads = n_reads; b->max_len = max_len;
  if (ctx->sp.on) { b->h_off.assign(seq_off, seq_off + n_reads); b->h_len.assign(seq_len, seq_len + n_reads); }
  hipStream_t s = ctx->stream;
  bool ok = hipMalloc(&b->d_seqs, n_bytes + 16) == hipSuccess && hipMalloc((void **)&b->d_off, std::max<uint64_t>(n_reads, 1) * 8) == hipSuccess &&
            hipMalloc((void **)&b->d_len, std::max<uint64_t>(n_reads, 1) * 4) == hipSuccess &&
            hipMalloc((void **)&b->d_taxa, (n_bytes + 16) * 4) == hipSuccess;
  if (!ok) { ku_batch_destroy(b); return fail(KU_ENOMEM, "device memory for a resident read batch"); }
  ok = (!n_bytes || hipMemcpyAsync(b->d_seqs, seqs, n_bytes, hipMemcpyHostToDevice, s) == hipSuccess) &&
       (!n_reads || (hipMemcpyAsync(b->d_off, seq_off, n_reads * 8, hipMemcpyHostToDevice, s) == hipSuccess &&
                     hipMemcpyAsync(b->d_len, seq_len, n_reads * 4, hipMemcpyHostToDevice, s) == hipSuccess)) &&
       hipMemsetAsync(b->d_taxa, 0, (n_bytes + 16) * 4, s) == hipSuccess && hipStreamSynchronize(s) == hipSuccess;
  if (!ok) { ku_batch_destroy(b); return fail(KU_EHIP, "upload of a resident read batch failed"); }
  *out = b;
  return KU_OK;
}

extern "C" int ku_batch_absorb(ku_ctx *ctx, ku_batch *dst, const ku_batch *src) {
  KU_TRY(check_ready(ctx));
  if (!dst || !src || dst->ctx != ctx || !src->ctx) return fail(KU_EINVAL, "ku_batch_absorb: null argument / batch of another context");
  if (dst->n_bytes != src->n_bytes || dst->n_reads != src->n_reads) return fail(KU_EINVAL, "ku_batch_absorb: the batches hold different reads");
  if (dst->finished || src->finished) return fail(KU_ESTATE, "ku_batch_absorb: a batch was already finished");
  if (dst->n_bytes == 0 || dst == src) return KU_OK;
  // the other copy's passes must be complete; then its slots come over (staged on this device when it lives on another)
  if (hipSetDevice(src->ctx->device) != hipSuccess || hipStreamSynchronize(src->ctx->stream) != hipSuccess) return fail(KU_EHIP, "ku_batch_absorb: the source context's stream failed");
  KU_TRY(ctx_activate(ctx));
  hipStream_t s = ctx->stream;
  const uint32_t *from = src->d_taxa;
  if (src->ctx->device != ctx->device) {
    HIP_TRY(hipStreamSynchronize(s));
    if (ctx->b_taxa.reserve(dst->n_bytes * 4) != KU_OK) return fail(KU_ENOMEM, "device memory for the slots of another GPU's batch");
    HIP_TRY(hipMemcpyAsync(ctx->b_taxa.p, src->d_taxa, dst->n_bytes * 4, hipMemcpyDefault, s));
    from = (const uint32_t *)ctx->b_taxa.p;
  }
  if (ku_launch_merge_max_u32(dst->d_taxa, from, dst->n_bytes, s) != KU_OK) return fail(KU_EHIP, "slot merge kernel launch failed");
  HIP_TRY(hipStreamSynchronize(s));
  return KU_OK;
}

extern "C" int ku_ctx_merge_state(ku_ctx *dst, ku_ctx *src) {
  KU_TRY(check_ready(dst));
  KU_TRY(check_ready(src));
  if (dst == src) return KU_OK;
  if (dst->tax.n_slots != src->tax.n_slots || dst->tax.n_nodes != src->tax.n_nodes) return fail(KU_EINVAL, "ku_ctx_merge_state: the contexts number their taxa differently");
  if (hipSetDevice(src->device) != hipSuccess || hipStreamSynchronize(src->stream) != hipSuccess) return fail(KU_EHIP, "ku_ctx_merge_state: the source context's stream failed");
  KU_TRY(ctx_activate(dst));
  hipStream_t s = dst->stream;
  const uint64_t n_regs = (uint64_t)dst->tax.n_slots * KU_HLL_M, n_slots = dst->tax.n_slots, n_nodes = dst->tax.n_nodes;
  const uint8_t *regs = src->cnt.registers;
  const unsigned long long *nk = src->cnt.n_kmers, *nr = src->cnt.n_reads;
  DevBuf stage;
  if (src->device != dst->device) {
    if (stage.reserve(n_regs + (n_slots + n_nodes) * 8) != KU_OK) return fail(KU_ENOMEM, "device memory for another GPU's per-taxon state");
    uint8_t *sp = (uint8_t *)stage.p;
    HIP_TRY(hipMemcpyAsync(sp, regs, n_regs, hipMemcpyDefault, s));
    HIP_TRY(hipMemcpyAsync(sp + n_regs, nk, n_slots * 8, hipMemcpyDefault, s));
    HIP_TRY(hipMemcpyAsync(sp + n_regs + n_slots * 8, nr, n_nodes * 8, hipMemcpyDefault, s));
    regs = sp;
    nk = (const unsigned long long *)(sp + n_regs);
    nr = (const unsigned long long *)(sp + n_regs + n_slots * 8);
  }
  if (ku_launch_merge_max_u8(dst->cnt.registers, regs, n_regs, s) != KU_OK || ku_launch_merge_add_u64(dst->cnt.n_kmers, nk, n_slots, s) != KU_OK ||
      ku_launch_merge_add_u64(dst->cnt.n_reads, nr, n_nodes, s) != KU_OK)
    return fail(KU_EHIP, "state merge kernel launch failed");
  HIP_TRY(hipStreamSynchronize(s));
  stage.release();
  return KU_OK;
}

extern "C" int ku_batch_lookup(ku_ctx *ctx, ku_batch *b, const ku_opts *opts) {
  KU_TRY(check_ready(ctx));
  if (!b || b->ctx != ctx) return fail(KU_EINVAL, "ku_batch_lookup: batch of another context");
  if (b->finished) return fail(KU_ESTATE, "ku_batch_lookup: the batch was already finished");
  ku_opts o = opts ? *opts : ku_opts{0, 1, 0, 0};
  // quick mode does not shorten a chunk pass: the reference's chunked run books every k-mer of every read and only
  // derives the call differently at the end (classify.cpp:686-737)
  o.flags = (o.flags & ~KU_F_QUICK) | KU_F_MERGE_CHUNK | KU_F_KEEP_SLOTS;
  return ku_lookup_device(ctx, b->d_seqs, b->n_bytes, &o, b->d_taxa, nullptr);
}

extern "C" int ku_batch_finish(ku_ctx *ctx, ku_batch *b, const ku_opts *opts, uint32_t *calls, uint32_t *hits,
                               uint64_t *run_off, uint32_t *run_cnt, uint64_t *n_runs) {
  KU_TRY(check_ready(ctx));
  if (!b || b->ctx != ctx) return fail(KU_EINVAL, "ku_batch_finish: batch of another context");
  if (!n_runs || (b->n_reads && (!calls || !run_off || !run_cnt))) return fail(KU_EINVAL, "ku_batch_finish: null buffer");
  if (b->finished) return fail(KU_ESTATE, "ku_batch_finish: the batch was already finished");
  *n_runs = 0;
  ctx->n_runs = 0;
  const uint64_t n_reads = b->n_reads;
  if (n_reads == 0) return KU_OK;
  ku_opts o = opts ? *opts : ku_opts{0, 1, 0, 0};
  o.flags &= ~(KU_F_KEEP_SLOTS | KU_F_MERGE_CHUNK);
  o.max_read_len = b->max_len;
  const uint64_t runs_cap = b->n_bytes + 1;
  if (ctx->b_calls.reserve(n_reads * 4) || ctx->b_hits.reserve(n_reads * 4) || ctx->b_runs.reserve(runs_cap * 8) ||
      ctx->b_roff.reserve(n_reads * 8) || ctx->b_rcnt.reserve(n_reads * 4))
    return fail(KU_ENOMEM, "device batch buffers");
  hipStream_t s = ctx->stream;
  if (ctx->sp.on && !(o.flags & KU_F_NO_COUNTS)) {  // the merged slots of all chunks are in place: the emulation's pass
    if (b->h_len.size() != n_reads) return fail(KU_ESTATE, "ku_batch_finish: enable the sparse-mode emulation before the batches are created");
    int sst = sparse_pass(ctx, b->d_seqs, b->d_off, b->d_len, b->h_off.data(), b->h_len.data(), n_reads, b->n_bytes, b->d_taxa, 0u, s);
    if (sst == KU_ENOMEM) {  // as in classify_device_impl: the run goes on without the emulation
      (void)hipStreamSynchronize(s);
      (void)hipGetLastError();
      ctx_free_sparse(ctx);
      ctx->sp.gave_up = true;
    } else if (sst != KU_OK) return sst;
  }
  if (ctx->d_exact_set && !(o.flags & KU_F_NO_COUNTS)) {
    // exact counting of a chunked run: the merged slots of all chunks are in place, and a chunked run books every k-mer of
    // every read whatever the mode (classify.cpp:686-737)
    int st = ku_launch_exact(ctx->m.db.k, (const uint8_t *)b->d_seqs, b->d_off, b->d_len, n_reads, b->d_taxa, ctx->d_exact_set, ctx->exact_mask,
                             ctx->d_exact_unique, ctx->d_scalar + 6, ctx->n_cu, s);
    if (st != KU_OK) return fail(st, "exact counting kernel launch failed");
  }
  if (o.flags & KU_F_QUICK) {  // the chunked run's quick mode: hits up to min_hits, call = the last k-mer's taxon
    int st = ku_launch_quick_chunked(ctx->tax, ctx->cnt, ctx->m.db.k, b->d_off, b->d_len, n_reads, o.flags, o.min_hits,
                                     (uint32_t *)ctx->b_calls.p, b->d_taxa, (uint32_t *)ctx->b_hits.p, ctx->n_cu, s);
    if (st != KU_OK) return fail(st, "quick-mode kernel launch failed");
  } else {
    KU_TRY(ku_resolve_device(ctx, b->d_seqs, b->d_off, b->d_len, n_reads, &o, (uint32_t *)ctx->b_calls.p, b->d_taxa,
                             (uint32_t *)ctx->b_hits.p, s));
  }
  b->finished = true;
  return rle_and_fetch(ctx, b->d_taxa, b->d_off, b->d_len, n_reads, runs_cap, (o.flags & KU_F_QUICK) != 0, calls, hits,
                       run_off, run_cnt, n_runs);
}

extern "C" int ku_ctx_synchronize(ku_ctx *ctx) {
  if (!ctx) return fail(KU_EINVAL, "null context");
  KU_TRY(ctx_activate(ctx));
  HIP_TRY(hipStreamSynchronize(ctx->stream));
  return KU_OK;
}

// ---------------------------------------------------------------------------- counts
extern "C" int ku_counts_dims_get(ku_ctx *ctx, ku_counts_dims *out) {
  if (!ctx || !out) return fail(KU_EINVAL, "ku_counts_dims_get: null argument");
  if (!ctx->tax_set) return fail(KU_ESTATE, "taxonomy not set");
  out->n_slots = ctx->tax.n_slots;
  out->n_nodes = ctx->tax.n_nodes;
  return KU_OK;
}

extern "C" int ku_counts_export(ku_ctx *ctx, uint32_t *slot_taxid, uint64_t *n_kmers, uint8_t *registers,
                                uint32_t *node_taxid, uint64_t *n_reads) {
  if (!ctx) return fail(KU_EINVAL, "null context");
  if (!ctx->tax_set) return fail(KU_ESTATE, "taxonomy not set");
  KU_TRY(ctx_activate(ctx));
  HIP_TRY(hipStreamSynchronize(ctx->stream));
  const size_t ns = ctx->tax.n_slots, nn = ctx->tax.n_nodes;
  if (slot_taxid) memcpy(slot_taxid, ctx->h_slot_taxid.data(), ns * 4);
  if (node_taxid) memcpy(node_taxid, ctx->h_node_taxid.data(), nn * 4);
  if (n_kmers) HIP_TRY(hipMemcpy(n_kmers, ctx->cnt.n_kmers, ns * 8, hipMemcpyDeviceToHost));
  if (registers) HIP_TRY(hipMemcpy(registers, ctx->cnt.registers, ns * KU_HLL_M, hipMemcpyDeviceToHost));
  if (n_reads) HIP_TRY(hipMemcpy(n_reads, ctx->cnt.n_reads, nn * 8, hipMemcpyDeviceToHost));
  return KU_OK;
}

// ---------------------------------------------------------------------------- report from the resident state
namespace {
struct DevTmp {  // device scratch of one ku_ctx_report call
  std::vector<void *> ptrs;
  ~DevTmp() { for (void *p : ptrs) (void)hipFree(p); }
  template <typename T> int put(T **dst, const std::vector<T> &src) {
    if (hipMalloc((void **)dst, std::max<size_t>(src.size(), 1) * sizeof(T)) != hipSuccess) { *dst = nullptr; return KU_ENOMEM; }
    ptrs.push_back(*dst);
    if (!src.empty() && hipMemcpy(*dst, src.data(), src.size() * sizeof(T), hipMemcpyHostToDevice) != hipSuccess) return KU_EHIP;
    return KU_OK;
  }
  // zeroed ON THE STREAM the kernels run on: a plain hipMemset goes to the null stream, which a non-blocking stream does not
  // wait for -- a large table could still be being cleared when the first kernel had already put entries into it (the
  // union sets of the sparse roll-up lost a few entries that way and counted their duplicates again)
  hipStream_t stream = nullptr;
  template <typename T> int zeros(T **dst, size_t n) {
    if (hipMalloc((void **)dst, std::max<size_t>(n, 1) * sizeof(T)) != hipSuccess) { *dst = nullptr; return KU_ENOMEM; }
    ptrs.push_back(*dst);
    return hipMemsetAsync(*dst, 0, std::max<size_t>(n, 1) * sizeof(T), stream) == hipSuccess ? KU_OK : KU_EHIP;
  }
};
}  // namespace

extern "C" int ku_ctx_report(ku_ctx *ctx, const ku_tax *tax, const char *const *counts_paths, uint32_t n_paths, char **out,
                             size_t *out_len) {
  return ku_ctx_report_cols(ctx, tax, counts_paths, n_paths, 0u, out, out_len);
}

extern "C" int ku_ctx_report_cols(ku_ctx *ctx, const ku_tax *tax, const char *const *counts_paths, uint32_t n_paths, uint32_t flags,
                                  char **out, size_t *out_len) {
  if (!ctx || !tax || !out || !out_len) return fail(KU_EINVAL, "ku_ctx_report: null argument");
  if (!ctx->tax_set) return fail(KU_ESTATE, "taxonomy not set");
  KU_TRY(rle_idle(ctx, "ku_ctx_report"));
  KU_TRY(ctx_activate(ctx));
  const size_t ns = ctx->tax.n_slots, nn = ctx->tax.n_nodes, nt = tax->ids.size();
  const bool six = (flags & KU_R_NO_KMER_COLS) != 0;  // `classify -p 0`: no k-mer columns, so no sketch is looked at
  const bool exact = ctx->d_exact_unique != nullptr || six, sparse = ctx->sp.on && !exact;
  // KU_REPORT_TIMES=1: where the call spends its time, on stderr
  const bool times = getenv("KU_REPORT_TIMES") != nullptr;
  auto now = [] { timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return (double)t.tv_sec + 1e-9 * (double)t.tv_nsec; };
  double t_last = now();
  auto lap = [&](const char *what) {
    if (!times) return;
    (void)hipStreamSynchronize(ctx->stream);
    const double t = now();
    fprintf(stderr, "ku_ctx_report: %-28s %8.1f ms\n", what, (t - t_last) * 1e3);
    t_last = t;
  };
  // the run-wide (slot, encoding) set of the sparse sketches is read where it lies (no compacted copy): the end of the run
  // closes the last, partial work unit (classify.cpp:522-523)
  uint64_t n_pairs = 0;
  std::vector<uint8_t> slot_sparse(ns, 0);
  if (sparse) {
    KU_TRY(sparse_close_open_unit(ctx));
    std::vector<uint32_t> dense(ns);
    unsigned long long total = 0;
    uint32_t err = 0;
    HIP_TRY(hipMemcpyAsync(&total, ctx->sp.dev.g_count, 8, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipMemcpyAsync(&err, ctx->sp.dev.err, 4, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipMemcpyAsync(dense.data(), ctx->sp.dev.dense, ns * 4, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    if (err) return fail(KU_ENOMEM, "sparse-mode emulation: a device table overflowed");
    for (size_t i = 0; i < ns; ++i) slot_sparse[i] = dense[i] ? 0 : 1;
    n_pairs = total;  // entries of the set (an upper bound of the sparse slots' entries)
    lap("close the last work unit");
  }
  HIP_TRY(hipStreamSynchronize(ctx->stream));
  std::vector<uint64_t> nk(ns), nr(nn), uq;
  HIP_TRY(hipMemcpy(nk.data(), ctx->cnt.n_kmers, ns * 8, hipMemcpyDeviceToHost));
  HIP_TRY(hipMemcpy(nr.data(), ctx->cnt.n_reads, nn * 8, hipMemcpyDeviceToHost));
  if (exact) {
    uq.assign(ns, 0);
    if (!six) HIP_TRY(hipMemcpy(uq.data(), ctx->d_exact_unique, ns * 8, hipMemcpyDeviceToHost));
  }
  // counted taxa (taxon_counts entries, classify.cpp:939,968) -> every entry of their root paths is a counted clade
  // (taxdb.hpp:928-973); taxa without a taxDB entry are dropped ("No entry for X in database!")
  std::vector<uint8_t> present(nt, 0);
  std::vector<uint64_t> c_reads(nt, 0), t_reads(nt, 0), c_kmers(nt, 0), c_uniq(nt, 0);
  std::vector<int32_t> clade_of(nt, -1);
  std::vector<uint32_t> clade_row;
  std::vector<std::pair<uint32_t, uint32_t>> memb;  // (clade, slot) over the root paths of the slots with k-mers
  std::vector<uint8_t> clade_dense;
  auto clade_id = [&](size_t row) {
    if (clade_of[row] < 0) { clade_of[row] = (int32_t)clade_row.size(); clade_row.push_back((uint32_t)row); clade_dense.push_back(0); present[row] = 1; }
    return (uint32_t)clade_of[row];
  };
  auto walk = [&](uint32_t taxid, auto &&visit) {
    auto it = tax->row.find(taxid);
    if (it == tax->row.end()) return;
    int64_t q = it->second;
    for (uint32_t guard = 0; q >= 0 && guard < 4096; ++guard, q = tax->parent_row((size_t)q)) visit((size_t)q);
  };
  for (size_t s = 0; s < ns; ++s) {
    if (!nk[s]) continue;
    const bool dense = !exact && !(sparse && slot_sparse[s]);
    walk(ctx->h_slot_taxid[s], [&](size_t row) {
      const uint32_t c = clade_id(row);
      c_kmers[row] += nk[s];
      if (exact) c_uniq[row] += uq[s];
      else memb.emplace_back(c, (uint32_t)s);
      if (dense) clade_dense[c] = 1;
    });
  }
  for (size_t i = 0; i < nn; ++i) {
    if (!nr[i]) continue;
    bool first = true;
    walk(ctx->h_node_taxid[i], [&](size_t row) {
      clade_id(row);
      c_reads[row] += nr[i];
      if (first) { t_reads[row] = nr[i]; first = false; }
    });
  }
  const uint32_t n_clades = (uint32_t)clade_row.size();
  if (!exact && n_clades) {
    DevTmp tmp;
    tmp.stream = ctx->stream;
    // members per clade (CSR)
    std::sort(memb.begin(), memb.end());
    std::vector<uint32_t> m_off(n_clades + 1, 0), m_slot(memb.size());
    for (size_t j = 0; j < memb.size(); ++j) { ++m_off[memb[j].first + 1]; m_slot[j] = memb[j].second; }
    for (uint32_t c = 0; c < n_clades; ++c) m_off[c + 1] += m_off[c];
    uint32_t *d_moff = nullptr, *d_mslot = nullptr, *d_hist = nullptr;
    uint8_t *d_dense = nullptr;
    int st = tmp.put(&d_moff, m_off);
    if (st == KU_OK) st = tmp.put(&d_mslot, m_slot);
    if (st == KU_OK) st = tmp.put(&d_dense, clade_dense);
    if (st == KU_OK) st = tmp.zeros(&d_hist, (size_t)n_clades * KU_ROLLUP_BINS);
    if (st != KU_OK) return fail(st, "ku_ctx_report: device memory for the clade roll-up");
    lap("clade lists (host)");
    KU_TRY(ku_launch_rollup_dense(ctx->cnt.registers, d_moff, d_mslot, d_dense, n_clades, d_hist, ctx->stream));
    lap("dense roll-up");
    // (entries of sparse sketches lie in the run-wide set G -- n_pairs of them -- and, since round 5, as SEEN marks in the probe
    // table: what the fused kernel's fast path booked, ku_device.h)
    if (sparse && (n_pairs || ctx->m.seen_dirty)) {
      // all-sparse clades per slot (its root path up to the first clade with a dense member: density is inherited upwards)
      std::vector<uint32_t> s_off(ns + 1, 0), s_clade;
      for (size_t s = 0; s < ns; ++s) {
        s_off[s] = (uint32_t)s_clade.size();
        if (!nk[s] || !slot_sparse[s]) continue;
        walk(ctx->h_slot_taxid[s], [&](size_t row) { if (!clade_dense[clade_of[row]]) s_clade.push_back((uint32_t)clade_of[row]); });
      }
      s_off[ns] = (uint32_t)s_clade.size();
      uint32_t *d_soff = nullptr, *d_sclade = nullptr, *d_err = nullptr, *d_set = nullptr, *d_setcells = nullptr;
      unsigned long long *d_setoff = nullptr;
      const KuSparseDev &sd = ctx->sp.dev;
      // What a slot may offer its clades: at most one entry per k-mer booked under it.  (Rounds 2-4 counted the set's entries
      // per slot first -- a pass over all of G through LDS tables, 25 ms of the report's 83 per 10 M reads; the bound sizes the
      // union sets generously instead, and the big clades take bitmaps of a fixed size anyway.)
      std::vector<unsigned long long> per_slot(ns, 0);
      for (size_t s = 0; s < ns; ++s)
        if (slot_sparse[s]) per_slot[s] = nk[s];
      std::vector<uint64_t> clade_pairs(n_clades, 0);  // entries each clade's histogram may receive
      for (size_t s = 0; s < ns; ++s)
        for (uint32_t j = s_off[s]; j < s_off[s + 1]; ++j) clade_pairs[s_clade[j]] += per_slot[s];
      // union sets, one table of 4-byte cells per clade (also for a clade with one member: the two sources may hold an
      // encoding twice): room for what its members offer -- at most every encoding there is (2^25 indices; the 2^12 of
      // them whose low 13 bits are zero come with up to 40 ranks) -- at a load of 2/3
      const uint64_t enc_space = (1ull << 25) + (1ull << 12) * 40;
      const uint64_t flag_space = (1ull << 12) * 40;  // encodings that carry the rank flag
      // BIG clades keep a bitmap over the 2^25 indices instead (4 MiB each; ku_report.hip): every clade that may receive
      // at least KU_ROLLUP_BITMAP_MIN entries (default 2^17; a test hook), most entries first and, among equals, nearest
      // the root first -- a clade's parent is offered at least as much as the clade, so whatever prefix of that order fits
      // the memory budget (a quarter of the free device memory) is closed upwards: above a bitmap there are only bitmaps
      std::vector<uint32_t> depth(n_clades, 0);
      for (uint32_t c = 0; c < n_clades; ++c) {
        int64_t q = tax->parent_row(clade_row[c]);
        for (uint32_t guard = 0; q >= 0 && guard < 4096; ++guard, q = tax->parent_row((size_t)q)) ++depth[c];
      }
      uint64_t bm_min = 1ull << 17;
      if (const char *e = getenv("KU_ROLLUP_BITMAP_MIN")) bm_min = std::max<uint64_t>(1, strtoull(e, nullptr, 10));
      std::vector<uint32_t> cand;
      for (uint32_t c = 0; c < n_clades; ++c)
        if (clade_pairs[c] >= bm_min && !clade_dense[c]) cand.push_back(c);
      std::sort(cand.begin(), cand.end(), [&](uint32_t a, uint32_t b) {
        return clade_pairs[a] != clade_pairs[b] ? clade_pairs[a] > clade_pairs[b] : (depth[a] != depth[b] ? depth[a] < depth[b] : a < b);
      });
      size_t free_b = 0, total_b = 0;
      (void)hipMemGetInfo(&free_b, &total_b);
      const size_t bm_budget = free_b / 4 / ((size_t)KU_BM_WORDS * 4);
      if (cand.size() > bm_budget) cand.resize(bm_budget);
      std::vector<uint32_t> bm_of(n_clades, KU_BM_NONE), bm_clade(cand);
      for (uint32_t b = 0; b < cand.size(); ++b) bm_of[cand[b]] = b;
      const uint32_t n_bm = (uint32_t)cand.size();
      // parents among the bitmap clades (the next clade up a slot's chain), children lists, parents by level
      std::vector<uint32_t> bm_parent(n_bm, KU_BM_NONE);
      for (size_t s = 0; s < ns; ++s)
        for (uint32_t j = s_off[s]; j + 1 < s_off[s + 1]; ++j)
          if (bm_of[s_clade[j]] != KU_BM_NONE) bm_parent[bm_of[s_clade[j]]] = bm_of[s_clade[j + 1]];
      std::vector<uint32_t> ch_off(n_bm + 1, 0), ch;
      for (uint32_t b = 0; b < n_bm; ++b)
        if (bm_parent[b] != KU_BM_NONE) ++ch_off[bm_parent[b] + 1];
      for (uint32_t b = 0; b < n_bm; ++b) ch_off[b + 1] += ch_off[b];
      ch.resize(ch_off[n_bm]);
      {
        std::vector<uint32_t> at(ch_off.begin(), ch_off.end() - 1);
        for (uint32_t b = 0; b < n_bm; ++b)
          if (bm_parent[b] != KU_BM_NONE) ch[at[bm_parent[b]]++] = b;
      }
      std::vector<uint32_t> bm_parents_by_level;  // parents with children, deepest level first
      std::vector<std::pair<uint32_t, uint32_t>> level_ranges;
      {
        std::vector<uint32_t> ps;
        for (uint32_t b = 0; b < n_bm; ++b)
          if (ch_off[b + 1] > ch_off[b]) ps.push_back(b);
        std::sort(ps.begin(), ps.end(), [&](uint32_t a, uint32_t b) { return depth[bm_clade[a]] != depth[bm_clade[b]] ? depth[bm_clade[a]] > depth[bm_clade[b]] : a < b; });
        for (size_t i = 0; i < ps.size();) {
          size_t j = i;
          while (j < ps.size() && depth[bm_clade[ps[j]]] == depth[bm_clade[ps[i]]]) ++j;
          level_ranges.emplace_back((uint32_t)i, (uint32_t)j);
          i = j;
        }
        bm_parents_by_level = ps;
      }
      std::vector<unsigned long long> set_off(n_clades, 0);
      std::vector<uint32_t> set_cells(n_clades, 0);
      uint64_t cells = 0;
      for (uint32_t c = 0; c < n_clades; ++c) {
        if (!clade_pairs[c] || clade_dense[c]) continue;
        // a bitmap clade's table only takes the entries with the rank flag (1 in 8192 of what hashes offer)
        const uint64_t bound = bm_of[c] != KU_BM_NONE ? std::min(clade_pairs[c] / 512 + 4096, flag_space) : std::min(clade_pairs[c], enc_space);
        set_off[c] = cells;
        set_cells[c] = (uint32_t)(bound + bound / 2 + 16);
        cells += set_cells[c];
      }
      // the busiest clades (the ones near the root) count in LDS
      std::vector<uint32_t> hot_clades(n_clades);
      for (uint32_t c = 0; c < n_clades; ++c) hot_clades[c] = c;
      const uint32_t n_hot = std::min<uint32_t>(KU_ROLLUP_HOT, n_clades);
      std::partial_sort(hot_clades.begin(), hot_clades.begin() + n_hot, hot_clades.end(),
                        [&](uint32_t a, uint32_t b) { return clade_pairs[a] != clade_pairs[b] ? clade_pairs[a] > clade_pairs[b] : a < b; });
      hot_clades.resize(n_hot);
      std::vector<uint16_t> clade_hot(n_clades, 0xFFFFu);
      for (uint32_t h = 0; h < n_hot; ++h) clade_hot[hot_clades[h]] = (uint16_t)h;
      uint16_t *d_chot = nullptr;
      uint32_t *d_hotc = nullptr;
      st = tmp.put(&d_soff, s_off);
      if (st == KU_OK) st = tmp.put(&d_sclade, s_clade);
      if (st == KU_OK) st = tmp.zeros(&d_err, 1);
      if (st == KU_OK) st = tmp.put(&d_chot, clade_hot);
      if (st == KU_OK) st = tmp.put(&d_hotc, hot_clades);
      if (st == KU_OK) st = tmp.put(&d_setoff, set_off);
      if (st == KU_OK) st = tmp.put(&d_setcells, set_cells);
      uint32_t *d_bmof = nullptr, *d_bm = nullptr, *d_bmclade = nullptr, *d_choff = nullptr, *d_ch = nullptr, *d_bmpar = nullptr;
      if (st == KU_OK) st = tmp.put(&d_bmof, bm_of);
      if (st == KU_OK) st = tmp.put(&d_bmclade, bm_clade);
      if (st == KU_OK) st = tmp.put(&d_choff, ch_off);
      if (st == KU_OK) st = tmp.put(&d_ch, ch);
      if (st == KU_OK) st = tmp.put(&d_bmpar, bm_parents_by_level);
      lap("union plan (host)");
      if (st == KU_OK) st = tmp.zeros(&d_set, cells);
      if (st == KU_OK) st = tmp.zeros(&d_bm, (size_t)std::max<uint32_t>(n_bm, 1) * (n_bm ? KU_BM_WORDS : 1));
      if (st != KU_OK) return fail(st, "ku_ctx_report: device memory for the union of the sparse sketches");
      lap("union set allocated + cleared");
      KuRollupPlan plan{};
      plan.dense = sd.dense; plan.slot_off = d_soff; plan.slot_clade = d_sclade; plan.set_off = d_setoff; plan.set_cells = d_setcells;
      plan.clade_hot = d_chot; plan.hot_clades = d_hotc; plan.n_hot = n_hot; plan.set = d_set; plan.hist = d_hist; plan.err = d_err;
      plan.bm_of = d_bmof; plan.bm = d_bm;
      if (n_pairs) KU_TRY(ku_launch_rollup_sparse(sd.g_key, sd.g_mask + 1, plan, ctx->n_cu, ctx->stream));
      if (ctx->m.seen_dirty && ctx->m.d_table) KU_TRY(ku_launch_rollup_table(ctx->m.d_table, ctx->m.db.n_lines, plan, ctx->n_cu, ctx->stream));
      for (const auto &lv : level_ranges)  // children into parents, deepest parents first
        KU_TRY(ku_launch_bitmap_or_children(d_bm, d_bmpar + lv.first, lv.second - lv.first, d_choff, d_ch, ctx->stream));
      KU_TRY(ku_launch_bitmap_hist(d_bm, d_bmclade, n_bm, d_hist, ctx->stream));
      uint32_t err = 0;
      HIP_TRY(hipMemcpyAsync(&err, d_err, 4, hipMemcpyDeviceToHost, ctx->stream));
      HIP_TRY(hipStreamSynchronize(ctx->stream));
      if (err) return fail(KU_EHIP, "ku_ctx_report: the sparse-union set overflowed");
      lap("sparse roll-up");
    }
    std::vector<uint32_t> hist((size_t)n_clades * KU_ROLLUP_BINS);
    HIP_TRY(hipMemcpyAsync(hist.data(), d_hist, hist.size() * 4, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    for (uint32_t c = 0; c < n_clades; ++c) {
      const size_t row = clade_row[c];
      if (c_reads[row] == 0) continue;  // not printed
      const bool has_members = m_off[c + 1] > m_off[c];
      // a clade counted through reads only has an empty sketch
      c_uniq[row] = has_members ? ku_hll_estimate_hist(hist.data() + (size_t)c * KU_ROLLUP_BINS, sparse && !clade_dense[c], c_kmers[row]) : 0;
    }
  }
  lap("estimates (host)");
  const int rst = ku_report_rows_cols(tax, counts_paths, n_paths, present.data(), c_reads.data(), t_reads.data(), c_kmers.data(), c_uniq.data(), nt,
                                      flags, out, out_len);
  lap("report text");
  return rst;
}

extern "C" int ku_counts_device_ptrs(ku_ctx *ctx, uint8_t **d_registers, uint64_t *n_register_bytes,
                                     uint64_t **d_n_kmers, uint64_t *n_slots, uint64_t **d_n_reads, uint64_t *n_nodes) {
  if (!ctx) return fail(KU_EINVAL, "null context");
  if (!ctx->tax_set) return fail(KU_ESTATE, "taxonomy not set");
  if (d_registers) *d_registers = ctx->cnt.registers;
  if (n_register_bytes) *n_register_bytes = (uint64_t)ctx->tax.n_slots * KU_HLL_M;
  if (d_n_kmers) *d_n_kmers = (uint64_t *)ctx->cnt.n_kmers;
  if (n_slots) *n_slots = ctx->tax.n_slots;
  if (d_n_reads) *d_n_reads = (uint64_t *)ctx->cnt.n_reads;
  if (n_nodes) *n_nodes = ctx->tax.n_nodes;
  return KU_OK;
}
