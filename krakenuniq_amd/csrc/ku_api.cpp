// ku_api.cpp -- C-ABI entry points (include/krakenuniq_amd.h): host-side database /
// taxonomy objects and the per-GPU context that owns the resident shard, the dense
// taxonomy tables and the per-taxon run state.  Compiled with hipcc together with
// ku_kernels.hip into libkrakenuniq_amd.so.  No CPU classification path exists
// here: every compute entry point needs a usable gfx950 device.  The C ABI spans six translation units, see ku_ctx.h.
#include "ku_ctx.h"

// ---------------------------------------------------------------------------- errors
static thread_local std::string g_last_error;
void ku_set_error(const std::string &s) { g_last_error = s; }

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

// ---------------------------------------------------------------------------- ku_ctx (the structs: ku_ctx.h)
// (what ku_mgpu.cpp may know of a context: declared in ku_internal.h)
hipStream_t ku_ctx_stream_of(ku_ctx *ctx) { return ctx->stream; }
unsigned long long *ku_ctx_exact_unique_of(ku_ctx *ctx) { return ctx->d_exact_unique; }
int ku_ctx_device_of(const ku_ctx *ctx) { return ctx->device; }
int ku_ctx_cus_of(const ku_ctx *ctx) { return ctx->n_cu; }
uint32_t ku_ctx_k_of(const ku_ctx *ctx) { return ctx->m.db.k; }

int ctx_activate(ku_ctx *ctx) {
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

void store_free(DbStore &d) {  // (callers that free a context's store also drop its count_taxons cache: ctx_drop_count_cache)
  if (d.d_table) (void)hipFree(d.d_table);
  if (d.db_owned && d.d_pairs) (void)hipFree(d.d_pairs);
  if (d.offsets_owned && d.d_offsets) (void)hipFree(d.d_offsets);
  d = DbStore{};
}
void ctx_drop_count_cache(ku_ctx *ctx) {
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
void ctx_free_sparse(ku_ctx *ctx) {
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
int store_upload(ku_ctx *ctx, DbStore &d, const ku_db *db, uint64_t bin_lo, uint64_t bin_hi, bool scan_values, hipStream_t stream) {
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

bool store_whole(const DbStore &d) { return d.db.bin_lo == 0 && d.db.bin_hi == (1ull << (2 * d.db.nt)); }

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
int store_finalize(ku_ctx *ctx, DbStore &d, hipStream_t stream, uint32_t *d_scalar) {
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
  d.slot_counts.clear();
  if (d.hash_layout && d.db.n_pairs) {
    unsigned long long *d_c = nullptr;
    const uint32_t ns = ctx->tax.n_slots;
    if (hipMalloc((void **)&d_c, (size_t)ns * 8) == hipSuccess) {
      std::vector<unsigned long long> h(ns);
      const bool ok = hipMemsetAsync(d_c, 0, (size_t)ns * 8, stream) == hipSuccess &&
                      ku_launch_count_slots(d.d_pairs, d.db.n_pairs, d_c, ns, stream) == KU_OK &&
                      hipMemcpyAsync(h.data(), d_c, (size_t)ns * 8, hipMemcpyDeviceToHost, stream) == hipSuccess &&
                      hipStreamSynchronize(stream) == hipSuccess;
      (void)hipFree(d_c);
      if (ok) d.slot_counts.swap(h);  // (not fatal: ku_ctx_count_taxons scans the table when they are missing)
      else (void)hipGetLastError();
    } else (void)hipGetLastError();
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

int sparse_pass(ku_ctx *ctx, const void *d_seqs, const uint64_t *d_off, const uint32_t *d_len, const uint64_t *h_off,
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
  if (d.db.table && d.slot_counts.size() == ns) {  // counted when the table was built
    const std::vector<unsigned long long> &h = d.slot_counts;
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
