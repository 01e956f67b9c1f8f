/* oracle/ku_oracle.c -- TEST INFRASTRUCTURE ONLY (see ku_oracle.h).
 *
 * Plain-C restatement of the KrakenUniq v1.0.4 classify hot path.  Every
 * function cites the reference file:line (under /root/reference/src) whose
 * behaviour it restates.  Nothing here is copied from the reference: the
 * containers (open-addressing tables instead of std::unordered_map/set) and the
 * control flow are this repo's own; only the arithmetic and the observable
 * behaviour follow the reference.  Pinned against oracle/_ref by
 * tests/test_oracle_golden.py.
 */
#define _GNU_SOURCE
#include "ku_oracle.h"

#include <fcntl.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define KO_NONE 0xFFFFFFFFu

/* ======================================================================== */
/* small open-addressing u32 -> u32 map (keys may be any value incl. 0)      */
/* ======================================================================== */
typedef struct {
  uint32_t *keys;
  uint32_t *vals;
  uint8_t *used;
  size_t cap; /* power of two */
  size_t n;
} u32map;

static uint32_t mix32(uint32_t x) {
  x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
  return x;
}
static void u32map_init(u32map *m, size_t cap_hint) {
  size_t cap = 16;
  while (cap < cap_hint * 2) cap <<= 1;
  m->keys = (uint32_t *)calloc(cap, 4);
  m->vals = (uint32_t *)calloc(cap, 4);
  m->used = (uint8_t *)calloc(cap, 1);
  m->cap = cap; m->n = 0;
}
static void u32map_free(u32map *m) { free(m->keys); free(m->vals); free(m->used); memset(m, 0, sizeof(*m)); }
static void u32map_clear(u32map *m) { memset(m->used, 0, m->cap); m->n = 0; }
static size_t u32map_slot(const u32map *m, uint32_t k) {
  size_t i = mix32(k) & (m->cap - 1);
  while (m->used[i] && m->keys[i] != k) i = (i + 1) & (m->cap - 1);
  return i;
}
static void u32map_grow(u32map *m);
/* returns pointer to value (inserted as 0 if new); *isnew set */
static uint32_t *u32map_get(u32map *m, uint32_t k, int *isnew) {
  if ((m->n + 1) * 2 > m->cap) u32map_grow(m);
  size_t i = u32map_slot(m, k);
  if (!m->used[i]) { m->used[i] = 1; m->keys[i] = k; m->vals[i] = 0; m->n++; if (isnew) *isnew = 1; }
  else if (isnew) *isnew = 0;
  return &m->vals[i];
}
static const uint32_t *u32map_find(const u32map *m, uint32_t k) {
  size_t i = u32map_slot(m, k);
  return m->used[i] ? &m->vals[i] : NULL;
}
static void u32map_grow(u32map *m) {
  u32map o = *m;
  u32map_init(m, o.cap);
  for (size_t i = 0; i < o.cap; ++i)
    if (o.used[i]) *u32map_get(m, o.keys[i], NULL) = o.vals[i];
  free(o.keys); free(o.vals); free(o.used);
}

/* ======================================================================== */
/* A1 scanner, A2 canonical, A3 bin key, hash                                */
/* ======================================================================== */

/* krakenutil.cpp:237-282: 2-bit code A=0 C=1 G=2 T=3 (case-insensitive), the
 * oldest base sits in the high bits, kmer_mask = ~0 >> (64-2k); a k-bit shift
 * register (mini_kmer_mask = ~0u >> (32-k)) carries one "ambiguous" bit per
 * base.  The reference's '\n'/'\r' skipping is not restated: its readers never
 * leave line terminators inside a sequence (seqreader.cpp:63-73,117) and with
 * them the reference reads past the string end. */
size_t ko_scan(const char *seq, size_t len, int k, uint64_t *fwd_out, uint8_t *ambig_out) {
  if (k <= 0 || k > 32 || len < (size_t)k) return 0; /* classify.cpp:913 */
  const uint64_t kmer_mask = ~0ULL >> (64 - 2 * k);
  const uint32_t mini_mask = ~0U >> (32 - k);
  uint64_t kmer = 0;
  uint32_t ambig = 0;
  size_t n = 0;
  for (size_t i = 0; i < len; ++i) {
    kmer <<= 2;
    ambig <<= 1;
    switch (seq[i]) {
      case 'A': case 'a': break;
      case 'C': case 'c': kmer |= 1; break;
      case 'G': case 'g': kmer |= 2; break;
      case 'T': case 't': kmer |= 3; break;
      default: ambig |= 1; break;
    }
    kmer &= kmer_mask;
    ambig &= mini_mask;
    if (i + 1 >= (size_t)k) {
      if (fwd_out) fwd_out[n] = kmer;
      if (ambig_out) ambig_out[n] = ambig != 0;
      ++n;
    }
  }
  return n;
}

/* krakendb.cpp:218-225: reverse the 2-bit groups, complement, drop the unused
 * high bits.  Restated with a per-base loop rather than the swap network. */
uint64_t ko_revcomp(uint64_t kmer, int n) {
  uint64_t r = 0;
  /* the reference reverses all 32 groups of the 64-bit word and then shifts
   * right by 64-2n, i.e. only the low n groups of the input survive */
  for (int i = 0; i < n; ++i) {
    r = (r << 2) | (3 - (kmer & 3));
    kmer >>= 2;
  }
  return r;
}

uint64_t ko_canonical(uint64_t kmer, int n) { /* krakendb.cpp:238-246 */
  uint64_t rc = ko_revcomp(kmer, n);
  return kmer < rc ? kmer : rc;
}

/* krakendb.cpp:182-215.  mask is computed in (32-bit) int in the reference, so
 * nt <= 15; XOR mask is 0 for the legacy KRAKIDX index (krakendb.cpp:203). */
uint64_t ko_bin_key(uint64_t kmer, int k, int nt, int idx_type) {
  const uint64_t INDEX2_XOR_MASK = 0xe37e28c4271b5a2dULL; /* krakendb.cpp:45 */
  uint64_t mask = (1ULL << (nt * 2)) - 1;
  uint64_t xor_mask = (idx_type == 1 ? 0 : INDEX2_XOR_MASK) & mask;
  uint64_t best = ~0ULL;
  for (int i = 0; i < k - nt + 1; ++i) {
    uint64_t t = xor_mask ^ ko_canonical(kmer & mask, nt);
    if (t < best) best = t;
    kmer >>= 2;
  }
  return best;
}

uint64_t ko_hash(uint64_t key) { /* hyperloglogplus.cpp:830-838 */
  key += 1;
  key ^= key >> 33; key *= 0xff51afd7ed558ccdULL;
  key ^= key >> 33; key *= 0xc4ceb9fe1a85ec53ULL;
  key ^= key >> 33;
  return key;
}

/* ======================================================================== */
/* A4-A6 database                                                            */
/* ======================================================================== */
struct ko_db {
  const uint8_t *pairs;
  const uint64_t *offsets;
  uint64_t key_ct;
  int k, nt, idx_type;
  uint64_t key_bits, key_len, pair_sz;
  void *map_kdb, *map_idx;
  size_t map_kdb_sz, map_idx_sz;
};

static void *map_file(const char *path, size_t *sz) {
  int fd = open(path, O_RDONLY);
  if (fd < 0) return NULL;
  struct stat st;
  if (fstat(fd, &st) != 0) { close(fd); return NULL; }
  void *p = mmap(NULL, (size_t)st.st_size, PROT_READ, MAP_PRIVATE, fd, 0);
  close(fd);
  if (p == MAP_FAILED) return NULL;
  *sz = (size_t)st.st_size;
  return p;
}

ko_db *ko_db_open(const char *kdb_path, const char *idx_path, char *err, size_t errlen) {
  size_t ksz = 0, isz = 0;
  uint8_t *kp = (uint8_t *)map_file(kdb_path, &ksz);
  if (!kp) { if (err) snprintf(err, errlen, "can't open %s", kdb_path); return NULL; }
  uint8_t *ip = (uint8_t *)map_file(idx_path, &isz);
  if (!ip) { munmap(kp, ksz); if (err) snprintf(err, errlen, "can't open %s", idx_path); return NULL; }
  ko_db *db = (ko_db *)calloc(1, sizeof(*db));
  db->map_kdb = kp; db->map_kdb_sz = ksz; db->map_idx = ip; db->map_idx_sz = isz;
  /* krakendb.cpp:67-77 */
  if (ksz < 72 || memcmp(kp, "JFLISTDN", 8) != 0) {
    if (err) snprintf(err, errlen, "database in improper format");
    ko_db_close(db); return NULL;
  }
  uint64_t val_len;
  memcpy(&db->key_bits, kp + 8, 8);
  memcpy(&val_len, kp + 16, 8);
  memcpy(&db->key_ct, kp + 48, 8);
  if (val_len != 4) {
    if (err) snprintf(err, errlen, "can only handle 4 byte DB values");
    ko_db_close(db); return NULL;
  }
  db->k = (int)(db->key_bits / 2);
  db->key_len = db->key_bits / 8 + !!(db->key_bits % 8);
  db->pair_sz = db->key_len + 4;
  size_t hdr = 72 + 2 * (4 + 8 * db->key_bits); /* krakendb.cpp:177 */
  db->pairs = kp + hdr;
  /* krakendb.cpp:534-544 */
  if (isz < 8) { if (err) snprintf(err, errlen, "illegal Kraken DB index format"); ko_db_close(db); return NULL; }
  if (memcmp(ip, "KRAKIDX", 7) == 0) db->idx_type = 1;
  else if (memcmp(ip, "KRAKIX2", 7) == 0) db->idx_type = 2;
  else { if (err) snprintf(err, errlen, "illegal Kraken DB index format"); ko_db_close(db); return NULL; }
  db->nt = ip[7];
  db->offsets = (const uint64_t *)(ip + 8);
  return db;
}

ko_db *ko_db_wrap(const void *pairs, uint64_t key_ct, int k, const uint64_t *offsets, int nt,
                  int idx_type) {
  ko_db *db = (ko_db *)calloc(1, sizeof(*db));
  db->pairs = (const uint8_t *)pairs; db->offsets = offsets; db->key_ct = key_ct;
  db->k = k; db->nt = nt; db->idx_type = idx_type;
  db->key_bits = 2 * (uint64_t)k;
  db->key_len = db->key_bits / 8 + !!(db->key_bits % 8);
  db->pair_sz = db->key_len + 4;
  return db;
}

void ko_db_close(ko_db *db) {
  if (!db) return;
  if (db->map_kdb) munmap(db->map_kdb, db->map_kdb_sz);
  if (db->map_idx) munmap(db->map_idx, db->map_idx_sz);
  free(db);
}
uint64_t ko_db_key_ct(const ko_db *db) { return db->key_ct; }
int ko_db_k(const ko_db *db) { return db->k; }
int ko_db_nt(const ko_db *db) { return db->nt; }
int ko_db_idx_type(const ko_db *db) { return db->idx_type; }
const uint64_t *ko_db_offsets(const ko_db *db) { return db->offsets; }
const uint8_t *ko_db_pairs(const ko_db *db) { return db->pairs; }

static inline uint64_t db_key_at(const ko_db *db, int64_t i) {
  uint64_t v = 0;
  memcpy(&v, db->pairs + db->pair_sz * (uint64_t)i, db->key_len);
  if (db->key_bits < 64) v &= (1ULL << db->key_bits) - 1; /* krakendb.cpp:284 */
  return v;
}
static inline uint32_t db_val_at(const ko_db *db, int64_t i) {
  uint32_t v;
  memcpy(&v, db->pairs + db->pair_sz * (uint64_t)i + db->key_len, 4);
  return v;
}

/* krakendb.cpp:279-299: bisect while the window is >= 16 wide, then scan.
 * Returns pair index or -1. */
static int64_t db_search(const ko_db *db, uint64_t kmer, int64_t min, int64_t max) {
  while (min + 15 <= max) {
    int64_t mid = min + (max - min) / 2;
    uint64_t c = db_key_at(db, mid);
    if (kmer > c) min = mid + 1;
    else if (kmer < c) max = mid - 1;
    else return mid;
  }
  for (int64_t mid = min; mid <= max; ++mid)
    if (db_key_at(db, mid) == kmer) return mid;
  return -1;
}

typedef struct { uint64_t bin; int64_t min, max; } ko_qstate; /* classify.cpp:115-120 */

/* krakendb.cpp:250-321 with the caller-side cache (retry_on_failure = true) */
static int64_t db_query_cached(const ko_db *db, uint64_t kmer, ko_qstate *st) {
  int64_t min, max;
  if (st->min <= st->max) { min = st->min; max = st->max; }
  else {
    uint64_t b = ko_bin_key(kmer, db->k, db->nt, db->idx_type);
    min = (int64_t)db->offsets[b];
    max = (int64_t)db->offsets[b + 1] - 1;
    st->bin = b; st->min = min; st->max = max;
  }
  int64_t pos = db_search(db, kmer, min, max);
  if (pos >= 0) return pos;
  uint64_t b = ko_bin_key(kmer, db->k, db->nt, db->idx_type);
  if (b == st->bin) return -1;
  min = (int64_t)db->offsets[b];
  max = (int64_t)db->offsets[b + 1] - 1;
  pos = db_search(db, kmer, min, max);
  st->bin = b; st->min = min; st->max = max;
  return pos;
}

int64_t ko_db_query(const ko_db *db, uint64_t kmer) { /* krakendb.cpp:324-326 */
  uint64_t b = ko_bin_key(kmer, db->k, db->nt, db->idx_type);
  int64_t pos = db_search(db, kmer, (int64_t)db->offsets[b], (int64_t)db->offsets[b + 1] - 1);
  return pos < 0 ? -1 : (int64_t)db_val_at(db, pos);
}

static int cmp_u32(const void *a, const void *b) {
  uint32_t x = *(const uint32_t *)a, y = *(const uint32_t *)b;
  return x < y ? -1 : x > y;
}

size_t ko_db_count_taxons(const ko_db *db, uint32_t *taxids, uint64_t *counts) { /* krakendb.cpp:90-113 */
  u32map m; u32map_init(&m, 1024);
  uint64_t *cnt = NULL; size_t ccap = 0;
  for (uint64_t i = 0; i < db->key_ct; ++i) {
    int isnew;
    uint32_t *slot = u32map_get(&m, db_val_at(db, (int64_t)i), &isnew);
    if (isnew) {
      if (m.n > ccap) { ccap = ccap ? ccap * 2 : 1024; cnt = (uint64_t *)realloc(cnt, ccap * 8); }
      *slot = (uint32_t)(m.n - 1);
      cnt[*slot] = 0;
    }
    cnt[*slot]++;
  }
  size_t n = m.n;
  if (taxids && counts) {
    uint32_t *ks = (uint32_t *)malloc(n * 4 + 4);
    size_t j = 0;
    for (size_t i = 0; i < m.cap; ++i) if (m.used[i]) ks[j++] = m.keys[i];
    qsort(ks, n, 4, cmp_u32); /* std::map iteration order */
    for (size_t i = 0; i < n; ++i) { taxids[i] = ks[i]; counts[i] = cnt[*u32map_find(&m, ks[i])]; }
    free(ks);
  }
  free(cnt); u32map_free(&m);
  return n;
}

/* ======================================================================== */
/* A10 taxonomy, A9 lca, A8 resolve_tree                                     */
/* ======================================================================== */
struct ko_tax {
  size_t n;
  uint32_t *ids;      /* as read */
  uint32_t *parents;  /* Parent_map value (0 for root/self/orphan) */
  uint32_t *file_parent;
  char **names;
  char **ranks;
  u32map index;       /* taxid -> row (entry 0 "unclassified" always present) */
};

static void tax_finish(ko_tax *t) {
  /* taxdb.hpp:411-433 createPointers + :383-398 getParentMap: the parent pointer
   * exists iff parent id != own id and the parent id has an entry */
  for (size_t i = 0; i < t->n; ++i) {
    uint32_t p = t->file_parent[i];
    if (t->ids[i] == 0) { t->parents[i] = 0; continue; }
    if (p != t->ids[i] && u32map_find(&t->index, p)) t->parents[i] = p;
    else t->parents[i] = 0;
  }
}

static void tax_push(ko_tax *t, size_t *cap, uint32_t id, uint32_t parent, const char *name, const char *rank) {
  int isnew;
  uint32_t *slot = u32map_get(&t->index, id, &isnew);
  if (!isnew) return; /* entries.insert keeps the first (taxdb.hpp:596) ... */
  if (t->n == *cap) {
    *cap = *cap ? *cap * 2 : 1024;
    t->ids = (uint32_t *)realloc(t->ids, *cap * 4);
    t->parents = (uint32_t *)realloc(t->parents, *cap * 4);
    t->file_parent = (uint32_t *)realloc(t->file_parent, *cap * 4);
    t->names = (char **)realloc(t->names, *cap * sizeof(char *));
    t->ranks = (char **)realloc(t->ranks, *cap * sizeof(char *));
  }
  *slot = (uint32_t)t->n;
  t->ids[t->n] = id; t->file_parent[t->n] = parent; t->parents[t->n] = 0;
  t->names[t->n] = strdup(name ? name : ""); t->ranks[t->n] = strdup(rank ? rank : "");
  t->n++;
}

/* taxdb.hpp:563-605: "id \t parent \t name \t rank-to-end-of-line" (classify
 * passes hasGenomeSizes=false, classify.cpp:218).  NB: parentMap[id] is
 * overwritten by later duplicates while entries keeps the first; duplicates do
 * not occur in real taxDB files and the two agree then. */
ko_tax *ko_tax_load(const char *path, char *err, size_t errlen) {
  FILE *f = fopen(path, "r");
  if (!f) { if (err) snprintf(err, errlen, "unable to open taxonomy index file %s", path); return NULL; }
  ko_tax *t = (ko_tax *)calloc(1, sizeof(*t));
  u32map_init(&t->index, 4096);
  size_t cap = 0;
  char *line = NULL; size_t lcap = 0; ssize_t ll;
  while ((ll = getline(&line, &lcap, f)) > 0) {
    if (line[ll - 1] == '\n') line[--ll] = 0;
    if (ll == 0) continue;
    char *p = line, *end;
    unsigned long id = strtoul(p, &end, 10); if (end == p) continue; p = end;
    unsigned long par = strtoul(p, &end, 10); if (end == p) continue; p = end;
    if (*p) ++p; /* inFile.get(): the tab */
    char *name = p;
    char *tab = strchr(p, '\t');
    const char *rank = "";
    if (tab) { *tab = 0; rank = tab + 1; }
    tax_push(t, &cap, (uint32_t)id, (uint32_t)par, name, rank);
  }
  free(line); fclose(f);
  tax_push(t, &cap, 0, 0, "unclassified", "no rank"); /* taxdb.hpp:599 */
  tax_finish(t);
  return t;
}

ko_tax *ko_tax_from_arrays(const uint32_t *ids, const uint32_t *parents, size_t n) {
  ko_tax *t = (ko_tax *)calloc(1, sizeof(*t));
  u32map_init(&t->index, n + 16);
  size_t cap = 0;
  for (size_t i = 0; i < n; ++i) tax_push(t, &cap, ids[i], parents[i], "", "");
  tax_push(t, &cap, 0, 0, "unclassified", "no rank");
  tax_finish(t);
  return t;
}

void ko_tax_free(ko_tax *t) {
  if (!t) return;
  for (size_t i = 0; i < t->n; ++i) { free(t->names[i]); free(t->ranks[i]); }
  free(t->ids); free(t->parents); free(t->file_parent); free(t->names); free(t->ranks);
  u32map_free(&t->index); free(t);
}
size_t ko_tax_size(const ko_tax *t) { return t->n; }

/* Parent_map lookup: key 0 is skipped by getParentMap (taxdb.hpp:388-389) */
uint32_t ko_tax_parent(const ko_tax *t, uint32_t taxid) {
  if (taxid == 0) return KO_NONE;
  const uint32_t *row = u32map_find(&t->index, taxid);
  return row ? t->parents[*row] : KO_NONE;
}

uint32_t ko_lca(const ko_tax *t, uint32_t a, uint32_t b) { /* krakenutil.cpp:90-118 */
  if (a == 0 || b == 0) return a ? a : b;
  uint32_t path[4096]; size_t np = 0;
  while (a > 1) {
    if (np < 4096) path[np++] = a;
    uint32_t p = ko_tax_parent(t, a);
    if (p == KO_NONE) break; /* "No parent for a" */
    a = p;
  }
  while (b > 1) {
    for (size_t i = 0; i < np; ++i) if (path[i] == b) return b;
    uint32_t p = ko_tax_parent(t, b);
    if (p == KO_NONE) break;
    b = p;
  }
  return 1;
}

uint32_t ko_resolve_tree(const ko_tax *t, const uint32_t *taxa, const uint32_t *counts, size_t n) {
  /* krakenutil.cpp:149-200.  The result does not depend on the iteration order
   * of hit_counts: the set of max-score taxa is collected and folded with lca()
   * in ascending taxid order (std::set). */
  if (n == 0) return 0;
  uint32_t *score = (uint32_t *)malloc(n * 4);
  uint32_t max_score = 0;
  for (size_t i = 0; i < n; ++i) {
    uint32_t node = taxa[i], s = 0;
    while (node > 0) {
      for (size_t j = 0; j < n; ++j) if (taxa[j] == node) { s += counts[j]; break; }
      uint32_t p = ko_tax_parent(t, node);
      if (p == KO_NONE) break;      /* "No parent for node recorded" */
      if (p == node) break;         /* "has itself as parent" */
      node = p;
    }
    score[i] = s;
    if (s > max_score) max_score = s;
  }
  /* taxa with score == max_score, ascending */
  uint32_t *tied = (uint32_t *)malloc(n * 4); size_t nt = 0;
  for (size_t i = 0; i < n; ++i) if (score[i] == max_score) tied[nt++] = taxa[i];
  qsort(tied, nt, 4, cmp_u32);
  uint32_t res = max_score == 0 ? 0 : tied[0];
  if (max_score != 0)
    for (size_t i = 1; i < nt; ++i) res = ko_lca(t, res, tied[i]);
  free(score); free(tied);
  return res;
}

/* ======================================================================== */
/* A13-A15 HyperLogLog++                                                     */
/* ======================================================================== */
#define KO_PPRIME 25
struct ko_hll {
  int p;
  uint32_t m;
  int sparse;
  uint64_t n_observed;
  uint8_t *M;            /* m registers when dense */
  /* sparse list = set of distinct encoded hashes (hyperloglogplus.hpp:48) */
  uint32_t *set;         /* open addressing, 0 = empty (see below) */
  size_t set_cap, set_n;
  int has_zero;          /* encoded value 0 tracked separately */
};

static inline int clz64(uint64_t x) { return x ? __builtin_clzll(x) : 64; }
static inline int clz32(uint32_t x) { return x ? __builtin_clz(x) : 32; }

/* hyperloglogplus.cpp:116-119,140-147 */
static inline uint32_t hll_index64(uint64_t h, int p) { return (uint32_t)(h >> (64 - p)); }
static inline uint8_t hll_rank64(uint64_t h, int p) {
  uint64_t bits = h << p;
  return (uint8_t)((bits == 0 ? 64 - p : clz64(bits)) + 1);
}
static inline uint8_t hll_rank32(uint32_t h, int p) {
  uint32_t bits = h << p;
  return (uint8_t)((bits == 0 ? 32 - p : clz32(bits)) + 1);
}
/* hyperloglogplus.cpp:181-204 */
static inline uint32_t hll_encode(uint64_t h, int p) {
  uint32_t idx = (uint32_t)((h >> (64 - KO_PPRIME)) << (32 - KO_PPRIME));
  if ((uint32_t)(idx << p) == 0) {
    uint8_t add = hll_rank64(h, KO_PPRIME);
    return idx | ((uint32_t)add << 1) | 1u;
  }
  return idx;
}
/* hyperloglogplus.cpp:152-161 (extractBits(v,7,1) = bits 6..1) */
static inline uint8_t hll_encoded_rank(uint32_t enc, int p) {
  if (enc & 1u) return (uint8_t)((KO_PPRIME - p) + ((enc >> 1) & 0x3f));
  return hll_rank32(enc, p);
}

static void set_init(ko_hll *h, size_t cap) {
  h->set_cap = cap; h->set_n = 0; h->has_zero = 0;
  h->set = (uint32_t *)calloc(cap, 4);
}
static int set_insert_raw(uint32_t *tab, size_t cap, uint32_t v) {
  size_t i = mix32(v) & (cap - 1);
  while (tab[i] && tab[i] != v) i = (i + 1) & (cap - 1);
  if (tab[i]) return 0;
  tab[i] = v; return 1;
}
static void set_insert(ko_hll *h, uint32_t v) {
  if (v == 0) { if (!h->has_zero) { h->has_zero = 1; } return; }
  if ((h->set_n + 1) * 2 > h->set_cap) {
    size_t ncap = h->set_cap * 2;
    uint32_t *nt = (uint32_t *)calloc(ncap, 4);
    for (size_t i = 0; i < h->set_cap; ++i) if (h->set[i]) set_insert_raw(nt, ncap, h->set[i]);
    free(h->set); h->set = nt; h->set_cap = ncap;
  }
  h->set_n += (size_t)set_insert_raw(h->set, h->set_cap, v);
}
static size_t set_size(const ko_hll *h) { return h->set_n + (size_t)h->has_zero; }

ko_hll *ko_hll_new(int p, int sparse) { /* hyperloglogplus.cpp:427-441 */
  if (p > 18 || p < 4) return NULL;
  ko_hll *h = (ko_hll *)calloc(1, sizeof(*h));
  h->p = p; h->m = 1u << p; h->sparse = sparse;
  if (sparse) set_init(h, 64);
  else h->M = (uint8_t *)calloc(h->m, 1);
  return h;
}
void ko_hll_free(ko_hll *h) { if (!h) return; free(h->M); free(h->set); free(h); }

static void hll_fold_encoded(ko_hll *h, uint32_t enc) { /* hyperloglogplus.cpp:559-577 */
  uint32_t idx = enc >> (32 - h->p);
  uint8_t r = hll_encoded_rank(enc, h->p);
  if (r > h->M[idx]) h->M[idx] = r;
}
static void hll_fold_set(ko_hll *dst, const ko_hll *src) {
  if (src->has_zero) hll_fold_encoded(dst, 0);
  for (size_t i = 0; i < src->set_cap; ++i) if (src->set[i]) hll_fold_encoded(dst, src->set[i]);
}
static void hll_to_dense(ko_hll *h) { /* hyperloglogplus.cpp:541-556 */
  if (!h->sparse) return;
  h->sparse = 0;
  h->M = (uint8_t *)calloc(h->m, 1);
  hll_fold_set(h, h);
  free(h->set); h->set = NULL; h->set_cap = h->set_n = 0; h->has_zero = 0;
}

void ko_hll_insert(ko_hll *h, uint64_t item) { /* hyperloglogplus.cpp:485-523 */
  ++h->n_observed;
  uint64_t hv = ko_hash(item);
  if (h->sparse && set_size(h) + 1 > h->m / 4) hll_to_dense(h);
  if (h->sparse) {
    set_insert(h, hll_encode(hv, h->p));
  } else {
    uint32_t idx = hll_index64(hv, h->p);
    uint8_t r = hll_rank64(hv, h->p);
    if (r > h->M[idx]) h->M[idx] = r;
  }
}

void ko_hll_merge(ko_hll *d, const ko_hll *s) { /* hyperloglogplus.cpp:586-665 */
  if (s->n_observed == 0) return;
  if (d->n_observed == 0) {
    /* adopt the other sketch wholesale */
    free(d->M); free(d->set); d->M = NULL; d->set = NULL;
    d->n_observed = s->n_observed; d->sparse = s->sparse;
    d->set_cap = d->set_n = 0; d->has_zero = 0;
    if (s->sparse) {
      d->set_cap = s->set_cap; d->set_n = s->set_n; d->has_zero = s->has_zero;
      d->set = (uint32_t *)malloc(s->set_cap * 4);
      memcpy(d->set, s->set, s->set_cap * 4);
    } else {
      d->M = (uint8_t *)malloc(d->m);
      memcpy(d->M, s->M, d->m);
    }
    return;
  }
  d->n_observed += s->n_observed;
  if (d->sparse && s->sparse) {
    /* plain set union, no size check (hyperloglogplus.cpp:601-604) */
    if (s->has_zero) set_insert(d, 0);
    for (size_t i = 0; i < s->set_cap; ++i) if (s->set[i]) set_insert(d, s->set[i]);
  } else if (s->sparse) {
    hll_fold_set(d, s);
  } else if (d->sparse) {
    d->sparse = 0;
    d->M = (uint8_t *)malloc(d->m);
    memcpy(d->M, s->M, d->m);
    hll_fold_set(d, d);
    free(d->set); d->set = NULL; d->set_cap = d->set_n = 0; d->has_zero = 0;
  } else {
    for (uint32_t i = 0; i < d->m; ++i) if (s->M[i] > d->M[i]) d->M[i] = s->M[i];
  }
}

/* hyperloglogplus.cpp:373-387 */
static double hll_sigma(double x) {
  if (x == 1.0) return INFINITY;
  double prev, sig = x, y = 1.0;
  do { prev = sig; x *= x; sig += x * y; y += y; } while (sig != prev);
  return sig;
}
/* hyperloglogplus.cpp:408-422 */
static double hll_tau(double x) {
  if (x == 0.0 || x == 1.0) return 0.0;
  double prev, y = 1.0, tau = 1 - x;
  do { prev = tau; x = sqrt(x); y /= 2.0; tau -= pow(1 - x, 2) * y; } while (tau != prev);
  return tau / 3.0;
}
/* hyperloglogplus.cpp:722-753 given the register histogram C[0..q+1] */
static uint64_t hll_ertl(const int *C, size_t q, double m, uint64_t n_observed, int use_n) {
  double den = m * hll_tau(1.0 - (double)C[q + 1] / m);
  for (int k = (int)q; k >= 1; --k) { den += C[k]; den *= 0.5; }
  den += m * hll_sigma((double)C[0] / m);
  double est = (m / (2.0 * log(2))) * m / den;
  if (use_n && (double)n_observed < est) return n_observed;
  return (uint64_t)round(est);
}

uint64_t ko_ertl_from_registers(const uint8_t *M, int p, uint64_t n_observed, int use_n) {
  int C[80] = {0};
  size_t q = 64 - (size_t)p;
  uint32_t m = 1u << p;
  for (uint32_t i = 0; i < m; ++i) C[M[i] < 79 ? M[i] : 79]++;
  return hll_ertl(C, q, (double)m, n_observed, use_n);
}

uint64_t ko_hll_cardinality(const ko_hll *h, int use_n) {
  if (!h->sparse) return ko_ertl_from_registers(h->M, h->p, h->n_observed, use_n);
  /* hyperloglogplus.cpp:356-366,726-729: m' = 2^25, q = 39, but the ranks in
   * the histogram are taken relative to p (getEncodedRank(...,pPrime,p)) */
  int C[80] = {0};
  size_t q = 64 - KO_PPRIME;
  int32_t m = 1 << KO_PPRIME;
  if (h->has_zero) { C[hll_encoded_rank(0, h->p)]++; --m; }
  for (size_t i = 0; i < h->set_cap; ++i)
    if (h->set[i]) { uint8_t r = hll_encoded_rank(h->set[i], h->p); C[r < 79 ? r : 79]++; --m; }
  C[0] = m;
  return hll_ertl(C, q, (double)(1 << KO_PPRIME), h->n_observed, use_n);
}

uint64_t ko_hll_n_observed(const ko_hll *h) { return h->n_observed; }
int ko_hll_is_sparse(const ko_hll *h) { return h->sparse; }
size_t ko_hll_sparse_size(const ko_hll *h) { return h->sparse ? set_size(h) : 0; }
size_t ko_hll_sparse_dump(const ko_hll *h, uint32_t *out, size_t cap) {
  if (!h->sparse) return 0;
  size_t n = set_size(h), j = 0;
  uint32_t *tmp = (uint32_t *)malloc((n + 1) * 4);
  if (h->has_zero) tmp[j++] = 0;
  for (size_t i = 0; i < h->set_cap; ++i) if (h->set[i]) tmp[j++] = h->set[i];
  qsort(tmp, n, 4, cmp_u32);
  memcpy(out, tmp, (n < cap ? n : cap) * 4);
  free(tmp);
  return n;
}
void ko_hll_registers(const ko_hll *h, uint8_t *out) {
  if (!h->sparse) { memcpy(out, h->M, h->m); return; }
  ko_hll tmp = *h; tmp.M = out; tmp.sparse = 0;
  memset(out, 0, h->m);
  hll_fold_set(&tmp, h);
}

/* ======================================================================== */
/* A7 classify_sequence, A11 hitlist, A12/A16 run with work units            */
/* ======================================================================== */
typedef struct {
  uint64_t n_reads, n_kmers;
  ko_hll *hll;
} ko_counts; /* readcounts.hpp:31-129 with CONTAINER = HLL(p=12, sparse) */

typedef struct {
  u32map idx;      /* taxid -> position in arrays */
  uint32_t *taxids;
  ko_counts *c;
  size_t n, cap;
} ko_cmap;

/* Test knob: 1 (default) = the reference's behaviour (sketches start sparse);
 * 0 = every sketch is dense from the start -- the "dense p=12 registers only"
 * model the GPU path implements (HLL section). */
static int g_hll_sparse = 1;
void ko_set_hll_sparse(int sparse) { g_hll_sparse = sparse; }

static void cmap_init(ko_cmap *m) { memset(m, 0, sizeof(*m)); u32map_init(&m->idx, 64); }
static ko_counts *cmap_get(ko_cmap *m, uint32_t taxid) { /* unordered_map::operator[] */
  int isnew;
  uint32_t *slot = u32map_get(&m->idx, taxid, &isnew);
  if (isnew) {
    if (m->n == m->cap) {
      m->cap = m->cap ? m->cap * 2 : 64;
      m->taxids = (uint32_t *)realloc(m->taxids, m->cap * 4);
      m->c = (ko_counts *)realloc(m->c, m->cap * sizeof(ko_counts));
    }
    *slot = (uint32_t)m->n;
    m->taxids[m->n] = taxid;
    m->c[m->n].n_reads = 0; m->c[m->n].n_kmers = 0;
    m->c[m->n].hll = ko_hll_new(12, g_hll_sparse); /* hyperloglogplus.hpp:87 defaults; -p is a no-op */
    m->n++;
  }
  return &m->c[*slot];
}
static void cmap_free(ko_cmap *m) {
  for (size_t i = 0; i < m->n; ++i) ko_hll_free(m->c[i].hll);
  free(m->taxids); free(m->c); u32map_free(&m->idx); memset(m, 0, sizeof(*m));
}
/* classify.cpp:542-544: taxon_counts[t] += local[t] */
static void cmap_merge(ko_cmap *dst, const ko_cmap *src) {
  for (size_t i = 0; i < src->n; ++i) {
    ko_counts *d = cmap_get(dst, src->taxids[i]);
    d->n_reads += src->c[i].n_reads;
    d->n_kmers += src->c[i].n_kmers;
    ko_hll_merge(d->hll, src->c[i].hll);
  }
}

/* one read; counts may be NULL (pure lookup) */
#define KO_MAX_DBS 8

/* ======================================================================== */
/* UID mapping (classify -I): resolve_uids3, uid_mapping.cpp:212-274
 *
 * With a UID database the values are UIDs: uid u names the taxid set found by
 * walking the {taxid, parent uid} blocks of the map file from block u - 1
 * (get_taxids_for_uid, uid_mapping.cpp:279-302).  A read's call is the taxid with
 * the most hits; ties go to the larger sum of count / |set| (double), then to
 * the LCA of the tied taxids.  The sums are accumulated, and ties are found, in
 * the ITERATION ORDER of std::unordered_map<uint32_t, ...> -- so this restates
 * that order too: libstdc++'s _Hashtable with the identity hash, buckets =
 * key % bucket_count, a singly linked node list in which a node enters at the
 * head of its bucket or, for an empty bucket, at the head of the whole list,
 * and the prime rehash policy (bucket counts 1, 13, 29, 59, 127, ...; checked
 * against the real container through oracle/ref_kat.cpp UMORDER).              */
/* ======================================================================== */
static const uint32_t ko_um_primes[] = { /* std::__detail::__prime_list (libstdc++), the part a read can reach */
  2,3,5,7,11,13,17,19,23,29,31,37,41,43,47,53,59,61,67,71,73,79,83,89,97,103,109,113,127,137,139,149,157,167,179,193,
  199,211,227,241,257,277,293,313,337,359,383,409,439,467,503,541,577,619,661,709,761,823,887,953,1031,1109,1193,1289,
  1381,1493,1613,1741,1879,2029,2179,2357,2549,2753,2971,3209,3469,3739,4027,4349,4703,5087,5503,5953,6427,6949,7517,
  8123,8783,9497,10273,11113,12011,12983,14033,15173,16411,17749,19183,20753,22447,24281,26267,28411,30727,33223,35933,
  38873,42043,45481,49201,53201,57557,62233,67307,72817,78779,85229,92203,99733,107897,116731,126271,136607,147793,
  159871,172933,187091,202409,218971,236897,256279,277261,299951,324503,351061,379787,410857,444487,480881,520241,
  562841,608903,658753,712697,771049,834181,902483,976369,1056323,1142821,1236397,1337629,1447153,1565659,1693859 };
#define KO_UM_NONE (-1)
#define KO_UM_BB (-2) /* &_M_before_begin */
typedef struct {
  uint32_t *key, *cnt; double *frac; int32_t *next;
  int32_t *bkt; size_t n, cap, n_bkt, next_resize; int32_t head;
} ko_um;
static void um_init(ko_um *m) {
  memset(m, 0, sizeof(*m));
  m->n_bkt = 1; m->bkt = (int32_t *)malloc(sizeof(int32_t)); m->bkt[0] = KO_UM_NONE; m->head = KO_UM_NONE;
}
static void um_free(ko_um *m) { free(m->key); free(m->cnt); free(m->frac); free(m->next); free(m->bkt); }
static int32_t um_next_of(const ko_um *m, int32_t node) { return node == KO_UM_BB ? m->head : m->next[node]; }
static void um_set_next(ko_um *m, int32_t node, int32_t v) { if (node == KO_UM_BB) m->head = v; else m->next[node] = v; }
static size_t um_next_bkt(ko_um *m, size_t n) { /* _Prime_rehash_policy::_M_next_bkt */
  static const unsigned char fast[] = { 2, 2, 2, 3, 5, 5, 7, 7, 11, 11, 11, 11, 13, 13 };
  if (n < sizeof(fast)) { if (n == 0) return 1; m->next_resize = fast[n]; return fast[n]; }
  size_t lo = 6, hi = sizeof(ko_um_primes) / sizeof(ko_um_primes[0]);
  while (lo < hi) { size_t mid = (lo + hi) / 2; if (ko_um_primes[mid] < n) lo = mid + 1; else hi = mid; }
  m->next_resize = ko_um_primes[lo];
  return ko_um_primes[lo];
}
static void um_rehash(ko_um *m, size_t n_new) { /* _M_rehash_aux(n, true_type) */
  int32_t *nb = (int32_t *)malloc(n_new * sizeof(int32_t));
  for (size_t i = 0; i < n_new; ++i) nb[i] = KO_UM_NONE;
  int32_t p = m->head; m->head = KO_UM_NONE;
  size_t bbegin_bkt = 0;
  while (p != KO_UM_NONE) {
    int32_t nx = m->next[p];
    size_t b = m->key[p] % n_new;
    if (nb[b] == KO_UM_NONE) {
      m->next[p] = m->head; m->head = p; nb[b] = KO_UM_BB;
      if (m->next[p] != KO_UM_NONE) nb[bbegin_bkt] = p;
      bbegin_bkt = b;
    } else {
      m->next[p] = um_next_of(m, nb[b]);
      um_set_next(m, nb[b], p);
    }
    p = nx;
  }
  free(m->bkt); m->bkt = nb; m->n_bkt = n_new;
}
static int32_t um_find(const ko_um *m, uint32_t k) {
  size_t b = k % m->n_bkt;
  int32_t prev = m->bkt[b];
  if (prev == KO_UM_NONE) return -1;
  int32_t p = um_next_of(m, prev);
  for (;;) {
    if (m->key[p] == k) return p;
    int32_t nx = m->next[p];
    if (nx == KO_UM_NONE || m->key[nx] % m->n_bkt != b) return -1;
    p = nx;
  }
}
static int32_t um_get(ko_um *m, uint32_t k) { /* operator[]: find or insert a zero-initialised value */
  int32_t f = um_find(m, k);
  if (f >= 0) return f;
  if (m->n + 1 > m->next_resize) { /* _M_need_rehash(n_bkt, n_elt, 1), max_load_factor 1 */
    size_t min_bkts = m->n + 1;
    if (!m->next_resize && min_bkts < 11) min_bkts = 11;
    if (min_bkts >= m->n_bkt) {
      size_t want = min_bkts + 1 > m->n_bkt * 2 ? min_bkts + 1 : m->n_bkt * 2;
      um_rehash(m, um_next_bkt(m, want));
    } else m->next_resize = m->n_bkt;
  }
  if (m->n == m->cap) {
    m->cap = m->cap ? 2 * m->cap : 64;
    m->key = (uint32_t *)realloc(m->key, m->cap * 4); m->cnt = (uint32_t *)realloc(m->cnt, m->cap * 4);
    m->frac = (double *)realloc(m->frac, m->cap * 8); m->next = (int32_t *)realloc(m->next, m->cap * 4);
  }
  int32_t node = (int32_t)m->n++;
  m->key[node] = k; m->cnt[node] = 0; m->frac[node] = 0.0;
  size_t b = k % m->n_bkt;
  if (m->bkt[b] != KO_UM_NONE) { /* _M_insert_bucket_begin */
    m->next[node] = um_next_of(m, m->bkt[b]);
    um_set_next(m, m->bkt[b], node);
  } else {
    m->next[node] = m->head; m->head = node;
    if (m->next[node] != KO_UM_NONE) m->bkt[m->key[m->next[node]] % m->n_bkt] = node;
    m->bkt[b] = KO_UM_BB;
  }
  return node;
}
/* iteration order of an unordered_map<uint32_t, T> after operator[] on keys[0..n) in that order */
size_t ko_umap_order(const uint32_t *keys, size_t n, uint32_t *out) {
  ko_um m; um_init(&m);
  for (size_t i = 0; i < n; ++i) um_get(&m, keys[i]);
  size_t j = 0;
  for (int32_t p = m.head; p != KO_UM_NONE; p = m.next[p]) out[j++] = m.key[p];
  um_free(&m);
  return j;
}

/* uids: the DB values of the read's unambiguous k-mers in k-mer order (zeros are skipped: classify.cpp:941
 * only counts hits); map: n_uid blocks {taxid, parent uid}.  A uid or parent beyond the map ends the walk
 * (the reference reads past its file there). */
uint32_t ko_resolve_uids3(const ko_tax *t, const uint32_t *uids, size_t n, const uint32_t *map, size_t n_uid) {
  ko_um hits; um_init(&hits);
  for (size_t i = 0; i < n; ++i)
    if (uids[i]) { const int32_t q = um_get(&hits, uids[i]); ++hits.cnt[q]; } /* (um_get may move the arrays) */
  if (hits.n == 0) { um_free(&hits); return 0; }
  ko_um tc; um_init(&tc);
  uint32_t *chain = NULL; size_t chain_cap = 0;
  for (int32_t p = hits.head; p != KO_UM_NONE; p = hits.next[p]) {
    size_t len = 0;
    for (uint32_t u = hits.key[p]; u != 0 && u <= n_uid;) {
      if (len == chain_cap) { chain_cap = chain_cap ? 2 * chain_cap : 16; chain = (uint32_t *)realloc(chain, chain_cap * 4); }
      chain[len++] = map[2 * (size_t)(u - 1)];
      u = map[2 * (size_t)(u - 1) + 1];
      if (len > n_uid) break; /* a cycle in a corrupt map */
    }
    if (!len) continue;
    const double frac = (double)hits.cnt[p] / (double)len;
    for (size_t i = 0; i < len; ++i) {
      int32_t q = um_get(&tc, chain[i]);
      tc.frac[q] += frac;
      tc.cnt[q] += hits.cnt[p];
    }
  }
  free(chain);
  uint32_t res = 0;
  if (tc.n) {
    uint32_t *best = (uint32_t *)calloc(tc.n, 4); size_t nb = 0;
    uint32_t max_count = 0; double max_frac = 0;
    for (int32_t p = tc.head; p != KO_UM_NONE; p = tc.next[p]) {
      if (tc.cnt[p] == max_count) {
        if (tc.frac[p] == max_frac) best[nb++] = tc.key[p];
        else if (tc.frac[p] > max_frac) { max_frac = tc.frac[p]; nb = 0; best[nb++] = tc.key[p]; }
      } else if (tc.cnt[p] > max_count) {
        nb = 0; best[nb++] = tc.key[p]; max_count = tc.cnt[p]; max_frac = tc.frac[p];
      }
    }
    res = best[0];
    for (size_t i = 1; i < nb; ++i) res = ko_lca(t, res, best[i]);
    free(best);
  }
  um_free(&hits); um_free(&tc);
  return res;
}

/* the run's UID map (ko_run_set_uid_map): NULL = plain taxid database */
typedef struct { const uint32_t *map; size_t n_uid; } ko_uidmap;

static uint32_t classify_one(const ko_db *const *dbs, int n_dbs, const ko_tax *tax, const char *seq, size_t len,
                             int quick, uint32_t min_hits, uint32_t *taxa_out, uint8_t *ambig_out, size_t *n_out,
                             uint32_t *hits_out, ko_cmap *counts, u32map *hit_counts, const ko_uidmap *um) {
  const int k = dbs[0]->k; /* classify.cpp:913: KrakenDatabases[0]->get_k(), all k equal (:199-208) */
  size_t n = 0;
  uint32_t taxon = 0, hits = 0;
  u32map_clear(hit_counts);
  uint32_t *hit_seq = NULL; size_t n_hit_seq = 0; /* UID mode: the hits in k-mer order (insertion order of hit_counts) */
  if (um && um->map && len >= (size_t)k) hit_seq = (uint32_t *)malloc((len - k + 1) * 4);
  if (len >= (size_t)k) { /* classify.cpp:913 */
    const uint64_t kmer_mask = ~0ULL >> (64 - 2 * k);
    const uint32_t mini_mask = ~0U >> (32 - k);
    uint64_t kmer = 0; uint32_t ambig = 0;
    ko_qstate st[KO_MAX_DBS]; /* classify.cpp:116,911: one cached bin per database */
    for (int d = 0; d < n_dbs; ++d) { st[d].bin = 0; st[d].min = 1; st[d].max = 0; }
    for (size_t i = 0; i < len; ++i) {
      kmer <<= 2; ambig <<= 1;
      switch (seq[i]) {
        case 'A': case 'a': break;
        case 'C': case 'c': kmer |= 1; break;
        case 'G': case 'g': kmer |= 2; break;
        case 'T': case 't': kmer |= 3; break;
        default: ambig |= 1; break;
      }
      kmer &= kmer_mask; ambig &= mini_mask;
      if (i + 1 < (size_t)k) continue;
      taxon = 0;
      int stop = 0;
      if (ambig) {
        if (ambig_out) ambig_out[n] = 1;
      } else {
        uint64_t canon = ko_canonical(kmer, k);
        if (ambig_out) ambig_out[n] = 0;
        for (int d = 0; d < n_dbs; ++d) { /* classify.cpp:928-936: the first database with the k-mer wins */
          int64_t pos = db_query_cached(dbs[d], canon, &st[d]);
          if (pos >= 0) { taxon = db_val_at(dbs[d], pos); break; }
        }
        if (counts) { /* classify.cpp:939: also when taxon == 0 */
          ko_counts *c = cmap_get(counts, taxon);
          ++c->n_kmers;
          ko_hll_insert(c->hll, canon);
        }
        if (taxon) {
          ++*u32map_get(hit_counts, taxon, NULL);
          if (hit_seq) hit_seq[n_hit_seq++] = taxon;
          if (quick && ++hits >= min_hits) stop = 1; /* classify.cpp:943-944: break before push */
        }
      }
      if (stop) break;
      if (taxa_out) taxa_out[n] = taxon;
      ++n;
    }
  }
  uint32_t call;
  if (um && um->map) { /* classify.cpp:953-960 (quick mode exits there) */
    call = ko_resolve_uids3(tax, hit_seq, n_hit_seq, um->map, um->n_uid);
    free(hit_seq);
  } else if (quick) call = hits >= min_hits ? taxon : 0; /* classify.cpp:962-963 */
  else {
    size_t nh = hit_counts->n;
    uint32_t *ts = (uint32_t *)malloc((nh + 1) * 4), *cs = (uint32_t *)malloc((nh + 1) * 4);
    size_t j = 0;
    for (size_t i = 0; i < hit_counts->cap; ++i)
      if (hit_counts->used[i]) { ts[j] = hit_counts->keys[i]; cs[j] = hit_counts->vals[i]; ++j; }
    call = ko_resolve_tree(tax, ts, cs, nh);
    free(ts); free(cs);
  }
  if (counts) cmap_get(counts, call)->n_reads++; /* classify.cpp:968 */
  if (n_out) *n_out = n;
  if (hits_out) *hits_out = hits;
  return call;
}

uint32_t ko_classify_read(const ko_db *db, const ko_tax *tax, const char *seq, size_t len, int quick,
                          uint32_t min_hits, uint32_t *taxa_out, uint8_t *ambig_out, size_t *n_out,
                          uint32_t *hits_out) {
  u32map hc; u32map_init(&hc, 64);
  uint32_t call = classify_one(&db, 1, tax, seq, len, quick, min_hits, taxa_out, ambig_out, n_out, hits_out,
                               NULL, &hc, NULL);
  u32map_free(&hc);
  return call;
}

size_t ko_hitlist_string(const uint32_t *taxa, const uint8_t *ambig, size_t n, char *buf) {
  /* classify.cpp:826-861 (+ "0:0" for an empty list, :994-995) */
  if (n == 0) { memcpy(buf, "0:0", 3); return 3; }
  char *p = buf;
  size_t i = 0;
  while (i < n) {
    size_t j = i + 1;
    if (ambig[i]) { while (j < n && ambig[j]) ++j; p += sprintf(p, "A:%zu", j - i); }
    else { while (j < n && !ambig[j] && taxa[j] == taxa[i]) ++j; p += sprintf(p, "%u:%zu", taxa[i], j - i); }
    if (j < n) *p++ = ' ';
    i = j;
  }
  return (size_t)(p - buf);
}

struct ko_run {
  const ko_db *dbs[KO_MAX_DBS]; int n_dbs; const ko_tax *tax;
  uint64_t unit_nt; int quick; uint32_t min_hits; int threads;
  ko_cmap global;
  uint64_t total_sequences, total_classified;
  /* sorted view */
  uint32_t *order; size_t order_n;
  ko_uidmap uid; /* classify -I */
};

ko_run *ko_run_new(const ko_db *db, const ko_tax *tax, uint64_t work_unit_nt, int quick, uint32_t min_hits,
                   int threads) {
  ko_run *r = (ko_run *)calloc(1, sizeof(*r));
  r->dbs[0] = db; r->n_dbs = 1; r->tax = tax; r->unit_nt = work_unit_nt ? work_unit_nt : 500000; /* classify.cpp:38 */
  r->quick = quick; r->min_hits = min_hits ? min_hits : 1; r->threads = threads > 0 ? threads : 1;
  cmap_init(&r->global);
  return r;
}
int ko_run_add_db(ko_run *r, const ko_db *db) { /* a further -d/-i pair, searched after the earlier ones */
  if (r->n_dbs >= KO_MAX_DBS || db->k != r->dbs[0]->k) return -1;
  r->dbs[r->n_dbs++] = db;
  return 0;
}
/* classify -I: the values of the (single) database are UIDs; map = the file's {taxid, parent uid} blocks */
void ko_run_set_uid_map(ko_run *r, const uint32_t *map, size_t n_uid) { r->uid.map = map; r->uid.n_uid = n_uid; }
void ko_run_free(ko_run *r) { if (!r) return; cmap_free(&r->global); free(r->order); free(r); }

void ko_run_classify(ko_run *r, const char *seqs, const uint64_t *off, const uint32_t *len, size_t n_reads,
                     uint32_t *calls, uint32_t *taxa_flat, uint8_t *ambig_flat, const uint64_t *taxa_off,
                     uint32_t *n_slots, uint32_t *hits) {
  /* classify.cpp:506-523: a unit takes reads until total_nt >= Work_unit_size;
   * a unit with total_nt == 0 ends processing (reads in it are dropped). */
  size_t *ustart = (size_t *)malloc((n_reads + 2) * sizeof(size_t));
  size_t nu = 0, i = 0;
  size_t n_valid = n_reads;
  while (i < n_reads) {
    uint64_t tot = 0; size_t s = i;
    while (i < n_reads && tot < r->unit_nt) tot += len[i++];
    if (tot == 0) { n_valid = s; break; }
    ustart[nu++] = s;
  }
  ustart[nu] = n_valid;
  free(r->order); r->order = NULL;
#ifdef _OPENMP
#pragma omp parallel num_threads(r->threads)
#endif
  {
    u32map hc; u32map_init(&hc, 64);
#ifdef _OPENMP
#pragma omp for schedule(dynamic, 1)
#endif
    for (size_t u = 0; u < nu; ++u) {
      ko_cmap local; cmap_init(&local); /* classify.cpp:525 */
      uint64_t ncls = 0;
      for (size_t j = ustart[u]; j < ustart[u + 1]; ++j) {
        size_t n = 0; uint32_t h = 0;
        uint32_t call = classify_one(r->dbs, r->n_dbs, r->tax, seqs + off[j], len[j], r->quick, r->min_hits,
                                     taxa_flat ? taxa_flat + taxa_off[j] : NULL,
                                     ambig_flat ? ambig_flat + taxa_off[j] : NULL, &n, &h, &local, &hc, &r->uid);
        if (calls) calls[j] = call;
        if (n_slots) n_slots[j] = (uint32_t)n;
        if (hits) hits[j] = h;
        ncls += call != 0;
      }
#ifdef _OPENMP
#pragma omp critical(ko_write_output)
#endif
      {
        r->total_classified += ncls;
        r->total_sequences += ustart[u + 1] - ustart[u];
        cmap_merge(&r->global, &local);
      }
      cmap_free(&local);
    }
    u32map_free(&hc);
  }
  free(ustart);
}

uint64_t ko_run_total_sequences(const ko_run *r) { return r->total_sequences; }
uint64_t ko_run_total_classified(const ko_run *r) { return r->total_classified; }

static const uint32_t *g_sort_keys;
static int cmp_by_taxid(const void *a, const void *b) {
  uint32_t x = g_sort_keys[*(const uint32_t *)a], y = g_sort_keys[*(const uint32_t *)b];
  return x < y ? -1 : x > y;
}
static void run_sort(ko_run *r) {
  if (r->order && r->order_n == r->global.n) return;
  free(r->order);
  r->order_n = r->global.n;
  r->order = (uint32_t *)malloc((r->order_n + 1) * 4);
  for (size_t i = 0; i < r->order_n; ++i) r->order[i] = (uint32_t)i;
  g_sort_keys = r->global.taxids;
  qsort(r->order, r->order_n, 4, cmp_by_taxid);
}
size_t ko_run_n_taxa(const ko_run *r) { return r->global.n; }
void ko_run_get(const ko_run *r, size_t i, uint32_t *taxid, uint64_t *n_reads, uint64_t *n_kmers,
                uint64_t *cardinality, int *is_sparse) {
  run_sort((ko_run *)r);
  size_t j = r->order[i];
  if (taxid) *taxid = r->global.taxids[j];
  if (n_reads) *n_reads = r->global.c[j].n_reads;
  if (n_kmers) *n_kmers = r->global.c[j].n_kmers;
  if (cardinality) *cardinality = ko_hll_cardinality(r->global.c[j].hll, 1);
  if (is_sparse) *is_sparse = r->global.c[j].hll->sparse;
}
const ko_hll *ko_run_sketch(const ko_run *r, size_t i) {
  run_sort((ko_run *)r);
  return r->global.c[r->order[i]].hll;
}

/* ======================================================================== */
/* A18 report                                                                */
/* ======================================================================== */
typedef struct { char *s; size_t n, cap; } sbuf;
static void sb_put(sbuf *b, const char *s, size_t n) {
  if (b->n + n + 1 > b->cap) { while (b->n + n + 1 > b->cap) b->cap = b->cap ? b->cap * 2 : 4096; b->s = (char *)realloc(b->s, b->cap); }
  memcpy(b->s + b->n, s, n); b->n += n; b->s[b->n] = 0;
}
static void sb_printf(sbuf *b, const char *fmt, ...) __attribute__((format(printf, 2, 3)));
#include <stdarg.h>
static void sb_printf(sbuf *b, const char *fmt, ...) {
  char tmp[512];
  va_list ap; va_start(ap, fmt);
  int n = vsnprintf(tmp, sizeof(tmp), fmt, ap);
  va_end(ap);
  if (n > 0) sb_put(b, tmp, (size_t)(n < (int)sizeof(tmp) ? n : (int)sizeof(tmp) - 1));
}

typedef struct {
  const ko_run *run; const ko_tax *tax;
  /* per taxonomy row */
  uint64_t *gsize, *gsize_children;
  uint64_t *clade_reads, *clade_kmers;
  ko_hll **clade_hll;
  uint8_t *has_clade;
  /* children lists */
  uint32_t *child_start, *child_list;
  sbuf out;
  uint64_t total_reads;
} report_ctx;

static int64_t tax_row(const ko_tax *t, uint32_t id) {
  const uint32_t *r = u32map_find(&t->index, id);
  return r ? (int64_t)*r : -1;
}
/* parent *row* following TaxonomyEntry::parent pointers (NULL -> -1) */
static int64_t tax_parent_row(const ko_tax *t, size_t row) {
  uint32_t id = t->ids[row], p = t->file_parent[row];
  if (p == id) return -1;
  return tax_row(t, p);
}

static report_ctx *g_rc;
static int cmp_children(const void *a, const void *b) {
  /* taxdb.hpp:1070: descending by (clade reads, clade kmers); ties -> ascending taxid (ours) */
  uint32_t x = *(const uint32_t *)a, y = *(const uint32_t *)b;
  if (g_rc->clade_reads[x] != g_rc->clade_reads[y]) return g_rc->clade_reads[x] > g_rc->clade_reads[y] ? -1 : 1;
  if (g_rc->clade_kmers[x] != g_rc->clade_kmers[y]) return g_rc->clade_kmers[x] > g_rc->clade_kmers[y] ? -1 : 1;
  return g_rc->tax->ids[x] < g_rc->tax->ids[y] ? -1 : g_rc->tax->ids[x] > g_rc->tax->ids[y];
}

static void report_node(report_ctx *rc, size_t row, unsigned depth) {
  /* taxdb.hpp:1049-1076 + printLine :1078-1123 */
  if (!rc->has_clade[row]) return;
  if (rc->clade_reads[row] == 0) return;
  const ko_tax *t = rc->tax;
  uint64_t uniq = ko_hll_cardinality(rc->clade_hll[row], 1);
  volatile double gs = (double)(rc->gsize[row] + rc->gsize_children[row]);
  uint64_t tax_reads = 0;
  const uint32_t *gi = u32map_find(&rc->run->global.idx, t->ids[row]);
  if (gi) tax_reads = rc->run->global.c[*gi].n_reads;
  sb_printf(&rc->out, "%.4g\t", 100.0 * (double)rc->clade_reads[row] / (double)rc->total_reads);
  sb_printf(&rc->out, "%llu\t%llu\t%llu\t", (unsigned long long)rc->clade_reads[row],
            (unsigned long long)tax_reads, (unsigned long long)uniq);
  volatile double kc = (double)rc->clade_kmers[row], un = (double)uniq;
  sb_printf(&rc->out, "%.3g\t", kc / un);
  if (gs == 0) sb_put(&rc->out, "NA\t", 3);
  else sb_printf(&rc->out, "%.4g\t", un / gs);
  if (t->ids[row] == 0xFFFFFFFFu) sb_put(&rc->out, "-1\t", 3);
  else sb_printf(&rc->out, "%d\t", (int32_t)t->ids[row]);
  sb_put(&rc->out, t->ranks[row], strlen(t->ranks[row]));
  sb_put(&rc->out, "\t", 1);
  for (unsigned d = 0; d < 2 * depth; ++d) sb_put(&rc->out, " ", 1);
  sb_put(&rc->out, t->names[row], strlen(t->names[row]));
  sb_put(&rc->out, "\n", 1);
  uint32_t s = rc->child_start[row], e = rc->child_start[row + 1];
  uint32_t *kids = (uint32_t *)malloc((e - s + 1) * 4); size_t nk = 0;
  for (uint32_t c = s; c < e; ++c) if (rc->has_clade[rc->child_list[c]]) kids[nk++] = rc->child_list[c];
  g_rc = rc;
  qsort(kids, nk, 4, cmp_children);
  for (size_t i = 0; i < nk; ++i) report_node(rc, kids[i], depth + 1);
  free(kids);
}

char *ko_run_report(const ko_run *r, const char *taxdb_path, const char *counts_path) {
  char err[256];
  ko_tax *t = ko_tax_load(taxdb_path, err, sizeof(err));
  if (!t) return NULL;
  report_ctx rc; memset(&rc, 0, sizeof(rc));
  rc.run = r; rc.tax = t;
  size_t n = t->n;
  rc.gsize = (uint64_t *)calloc(n, 8); rc.gsize_children = (uint64_t *)calloc(n, 8);
  rc.clade_reads = (uint64_t *)calloc(n, 8); rc.clade_kmers = (uint64_t *)calloc(n, 8);
  rc.clade_hll = (ko_hll **)calloc(n, sizeof(ko_hll *)); rc.has_clade = (uint8_t *)calloc(n, 1);
  /* children lists from parent pointers */
  rc.child_start = (uint32_t *)calloc(n + 2, 4); rc.child_list = (uint32_t *)calloc(n + 1, 4);
  for (size_t i = 0; i < n; ++i) { int64_t p = tax_parent_row(t, i); if (p >= 0) rc.child_start[p + 1]++; }
  for (size_t i = 0; i < n; ++i) rc.child_start[i + 1] += rc.child_start[i];
  {
    uint32_t *fill = (uint32_t *)malloc((n + 1) * 4);
    memcpy(fill, rc.child_start, (n + 1) * 4);
    for (size_t i = 0; i < n; ++i) { int64_t p = tax_parent_row(t, i); if (p >= 0) rc.child_list[fill[p]++] = (uint32_t)i; }
    free(fill);
  }
  /* taxdb.hpp:867-885 readGenomeSizes: "while(!eof) { in >> id >> size; set(id,size); }"
   * -> when the file ends with whitespace after the last number the failed
   * extraction leaves (id,size) unchanged and the LAST pair is applied twice. */
  /* several databases: one counts file each, read in order (classify.cpp:263-285); here '\n'-separated */
  char *paths = counts_path ? strdup(counts_path) : NULL;
  for (char *one = paths; one && *one;) {
    char *nl = strchr(one, '\n');
    if (nl) *nl = 0;
    FILE *f = fopen(one, "r");
    if (f) {
      fseek(f, 0, SEEK_END); long sz = ftell(f); fseek(f, 0, SEEK_SET);
      char *buf = (char *)malloc((size_t)sz + 1);
      size_t got = fread(buf, 1, (size_t)sz, f); buf[got] = 0; fclose(f);
      char *p = buf; unsigned long long id = 0, size = 0; int have = 0, trailing = 0;
      for (;;) {
        char *e1, *e2;
        unsigned long long a = strtoull(p, &e1, 10); if (e1 == p) break;
        unsigned long long b = strtoull(e1, &e2, 10); if (e2 == e1) break;
        id = a; size = b; have = 1; p = e2;
        trailing = (*p != 0);
        int64_t row = tax_row(t, (uint32_t)id);
        if (row < 0) continue; /* "No taxonomy entry for X!!" */
        rc.gsize[row] += size;
        for (int64_t q = tax_parent_row(t, (size_t)row); q >= 0; q = tax_parent_row(t, (size_t)q)) rc.gsize_children[q] += size;
      }
      if (have && trailing) {
        int64_t row = tax_row(t, (uint32_t)id);
        if (row >= 0) {
          rc.gsize[row] += size;
          for (int64_t q = tax_parent_row(t, (size_t)row); q >= 0; q = tax_parent_row(t, (size_t)q)) rc.gsize_children[q] += size;
        }
      }
      free(buf);
    }
    one = nl ? nl + 1 : NULL;
  }
  free(paths);
  /* taxdb.hpp:928-973: every taxon with counts contributes to itself and all ancestors */
  for (size_t i = 0; i < r->global.n; ++i) {
    int64_t row = tax_row(t, r->global.taxids[i]);
    if (row < 0) continue; /* "No entry for X in database!" */
    for (int64_t q = row; q >= 0; q = tax_parent_row(t, (size_t)q)) {
      if (!rc.has_clade[q]) { rc.has_clade[q] = 1; rc.clade_hll[q] = ko_hll_new(12, g_hll_sparse); }
      rc.clade_reads[q] += r->global.c[i].n_reads;
      rc.clade_kmers[q] += r->global.c[i].n_kmers;
      ko_hll_merge(rc.clade_hll[q], r->global.c[i].hll);
    }
  }
  const uint32_t roots[3] = {0, 1, 0xFFFFFFFFu};
  for (int i = 0; i < 3; ++i) { int64_t row = tax_row(t, roots[i]); if (row >= 0 && rc.has_clade[row]) rc.total_reads += rc.clade_reads[row]; }
  if (rc.total_reads != 0) {
    const char *hdr = "%\treads\ttaxReads\tkmers\tdup\tcov\ttaxID\trank\ttaxName\n"; /* classify.cpp:305-314 */
    sb_put(&rc.out, hdr, strlen(hdr));
    for (int i = 0; i < 3; ++i) { int64_t row = tax_row(t, roots[i]); if (row >= 0) report_node(&rc, (size_t)row, 0); }
  } else sb_put(&rc.out, "", 0);
  for (size_t i = 0; i < n; ++i) ko_hll_free(rc.clade_hll[i]);
  free(rc.gsize); free(rc.gsize_children); free(rc.clade_reads); free(rc.clade_kmers);
  free(rc.clade_hll); free(rc.has_clade); free(rc.child_start); free(rc.child_list);
  ko_tax_free(t);
  if (!rc.out.s) { rc.out.s = (char *)calloc(1, 1); }
  return rc.out.s;
}

void ko_free(void *p) { free(p); }
