// ku_host.cpp -- host-side arithmetic and text of the classify boundary: the Ertl
// cardinality estimator (double, once per report row), the Kraken output line and
// the report.  The reference does these on the host as well (SURVEY.md 8a A11, A15,
// A18); nothing here classifies reads.
#include "ku_host.h"

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>

// ---------------------------------------------------------------------------- HLL estimator
// sigma / tau series of Ertl's improved estimator (hyperloglogplus.cpp:373-387,408-422)
static double ertl_sigma(double x) {
  if (x == 1.0) return INFINITY;
  double prev, sig = x, y = 1.0;
  do {
    prev = sig;
    x *= x;
    sig += x * y;
    y += y;
  } while (sig != prev);
  return sig;
}
static double ertl_tau(double x) {
  if (x == 0.0 || x == 1.0) return 0.0;
  double prev, y = 1.0, t = 1 - x;
  do {
    prev = t;
    x = std::sqrt(x);
    y /= 2.0;
    t -= std::pow(1 - x, 2) * y;
  } while (t != prev);
  return t / 3.0;
}

// ertlCardinality (hyperloglogplus.cpp:722-753) from the register histogram C[0 .. q+1] of a sketch with m registers
static uint64_t ertl_from_histogram(const int *C, uint32_t q, double m, uint64_t n_observed) {
  double den = m * ertl_tau(1.0 - double(C[q + 1]) / m);
  for (int k = (int)q; k >= 1; --k) {
    den += C[k];
    den *= 0.5;
  }
  den += m * ertl_sigma(double(C[0]) / m);
  double est = (m / (2.0 * std::log(2))) * m / den;
  if (double(n_observed) < est) return n_observed;  // use_n_observed = true (hyperloglogplus.hpp:78)
  return (uint64_t)std::llround(est);
}

extern "C" uint64_t ku_hll_cardinality(const uint8_t *M, uint32_t p, uint64_t n_observed) {
  // dense registers: q = 64 - p, m = 2^p
  if (!M || p < 4 || p > 18) return 0;
  const uint32_t m = 1u << p, q = 64 - p;
  std::vector<int> C(q + 2, 0);
  for (uint32_t i = 0; i < m; ++i) C[std::min<uint32_t>(M[i], q + 1)]++;
  return ertl_from_histogram(C.data(), q, double(m), n_observed);
}

// The same estimator on a SPARSE sketch: the set of 32-bit encoded hashes at precision p' = 25
// (sparseRegisterHistogram, hyperloglogplus.cpp:356-366; cardinality :726-729): m = 2^25 virtual registers, q = 39,
// one histogram entry per encoded hash at its rank relative to p (getEncodedRank, :152-161), the rest are zero registers.
extern "C" uint64_t ku_hll_cardinality_sparse(const uint32_t *encoded, uint64_t n, uint64_t n_observed) {
  const uint32_t pp = 25, p = KU_HLL_P, q = 64 - pp;
  int C[80] = {0};
  int64_t m = 1ll << pp;
  for (uint64_t i = 0; i < n; ++i) {
    const uint32_t e = encoded[i];
    uint32_t r;
    if (e & 1u) r = (pp - p) + ((e >> 1) & 0x3Fu);
    else {
      const uint32_t bits = e << p;
      r = (bits == 0 ? 32 - p : (uint32_t)__builtin_clz(bits)) + 1;
    }
    C[r < 79 ? r : 79]++;
    --m;
  }
  C[0] = (int)m;
  return ertl_from_histogram(C, q, double(1 << pp), n_observed);
}

// ---------------------------------------------------------------------------- UID databases (classify -I)
struct ku_uid_map {
  std::vector<uint32_t> blocks;  // {taxid, parent uid} of uid i + 1 at 2 * i
};
extern "C" int ku_uid_map_from_blocks(const uint32_t *blocks, uint64_t n, ku_uid_map **out) {
  if (!out || (n && !blocks)) { ku_set_error("ku_uid_map_from_blocks: null argument"); return KU_EINVAL; }
  ku_uid_map *m = new ku_uid_map();
  m->blocks.assign(blocks, blocks + 2 * n);
  *out = m;
  return KU_OK;
}
extern "C" int ku_uid_map_open(const char *path, ku_uid_map **out) {
  if (!path || !out) { ku_set_error("ku_uid_map_open: null argument"); return KU_EINVAL; }
  *out = nullptr;
  FILE *f = fopen(path, "rb");
  if (!f) { ku_set_error(std::string("can't open ") + path); return KU_ENOINPUT; }
  fseek(f, 0, SEEK_END);
  const long sz = ftell(f);
  fseek(f, 0, SEEK_SET);
  if (sz < 0 || sz % 8) { fclose(f); ku_set_error("UID map: the file is not a sequence of {taxid, parent uid} blocks"); return KU_EDATA; }
  ku_uid_map *m = new ku_uid_map();
  m->blocks.resize((size_t)sz / 4);
  const size_t got = sz ? fread(m->blocks.data(), 4, (size_t)sz / 4, f) : 0;
  fclose(f);
  if (got != (size_t)sz / 4) { delete m; ku_set_error("UID map: short read"); return KU_EDATA; }
  *out = m;
  return KU_OK;
}
extern "C" void ku_uid_map_close(ku_uid_map *m) { delete m; }
extern "C" uint64_t ku_uid_map_size(const ku_uid_map *m) { return m ? m->blocks.size() / 2 : 0; }

// lca (krakenutil.cpp:90-118) on the host Parent_map: a taxid without an entry ends the walk ("No parent for ...")
static uint32_t host_lca(const ku_tax *t, uint32_t a, uint32_t b) {
  if (a == 0 || b == 0) return a ? a : b;
  std::vector<uint32_t> a_path;
  for (uint32_t guard = 0; a > 1 && guard < 4096; ++guard) {
    a_path.push_back(a);
    auto it = t->row.find(a);
    if (it == t->row.end() || a == 0) break;
    a = t->parent_map[it->second];
  }
  for (uint32_t guard = 0; b > 1 && guard < 4096; ++guard) {
    if (std::find(a_path.begin(), a_path.end(), b) != a_path.end()) return b;
    auto it = t->row.find(b);
    if (it == t->row.end()) break;
    b = t->parent_map[it->second];
  }
  return 1;
}

// resolve_uids3 (uid_mapping.cpp:212-274) for one read; `hits` was filled by hit_counts[uid]++ in k-mer order like
// classify_sequence does (classify.cpp:941): the containers and their insertion order are the reference's, so is the
// order the sums and the ties are formed in
static int resolve_uids_one(const ku_tax *tax, const ku_uid_map *map, const std::unordered_map<uint32_t, uint32_t> &hits, uint32_t *call) {
  *call = 0;
  if (hits.empty()) return KU_OK;
  const uint64_t n_uid = map->blocks.size() / 2;
  std::unordered_map<uint32_t, uint32_t> taxid_counts;
  std::unordered_map<uint32_t, double> frac_taxid_counts;
  std::vector<uint32_t> taxids;
  for (auto it = hits.begin(); it != hits.end(); ++it) {
    if (it->first == 0) continue;
    taxids.clear();
    for (uint32_t u = it->first; u != 0;) {  // get_taxids_for_uid (uid_mapping.cpp:279-302)
      if (u > n_uid || taxids.size() > n_uid) { ku_set_error("UID " + std::to_string(u) + " is not in the UID map"); return KU_EDATA; }
      taxids.push_back(map->blocks[2 * (size_t)(u - 1)]);
      u = map->blocks[2 * (size_t)(u - 1) + 1];
    }
    const double frac_count = (double)it->second / (double)taxids.size();
    for (size_t i = 0; i < taxids.size(); ++i) {
      frac_taxid_counts[taxids[i]] += frac_count;
      taxid_counts[taxids[i]] += it->second;
    }
  }
  if (taxid_counts.empty()) return KU_OK;
  std::vector<uint32_t> max_taxids;
  uint32_t max_count = 0;
  double max_frac_count = 0;
  for (auto it = taxid_counts.begin(); it != taxid_counts.end(); ++it) {
    if (it->second == max_count) {
      const double f = frac_taxid_counts[it->first];
      if (f == max_frac_count) max_taxids.push_back(it->first);
      else if (f > max_frac_count) { max_frac_count = f; max_taxids.assign(1, it->first); }
    } else if (it->second > max_count) {
      max_taxids.assign(1, it->first);
      max_count = it->second;
      max_frac_count = frac_taxid_counts[it->first];
    }
  }
  uint32_t max_taxon = max_taxids[0];
  for (size_t i = 1; i < max_taxids.size(); ++i) max_taxon = host_lca(tax, max_taxon, max_taxids[i]);
  *call = max_taxon;
  return KU_OK;
}

extern "C" int ku_resolve_uids(const ku_tax *tax, const ku_uid_map *map, const ku_run *runs, const uint64_t *run_off, const uint32_t *run_cnt,
                               const uint32_t *seq_len, uint64_t n_reads, uint32_t k, uint32_t n_threads, uint32_t *calls) {
  if (!tax || !map || (n_reads && (!run_off || !run_cnt || !seq_len || !calls))) { ku_set_error("ku_resolve_uids: null argument"); return KU_EINVAL; }
  if (n_threads == 0) n_threads = 1;
  if (n_reads < 4096) n_threads = 1;
  std::vector<int> status(n_threads, KU_OK);
  std::vector<std::string> msg(n_threads);
  auto work = [&](uint32_t t) {
    std::unordered_map<uint32_t, uint32_t> hits;
    for (uint64_t i = n_reads * t / n_threads; i < n_reads * (t + 1) / n_threads; ++i) {
      hits = std::unordered_map<uint32_t, uint32_t>();  // a fresh container per read, as in classify_sequence (:902)
      const uint32_t n = seq_len[i] >= k ? seq_len[i] - k + 1 : 0, c = run_cnt[i];
      for (uint32_t j = 0; j < c; ++j) {
        const ku_run &r = runs[run_off[i] + j];
        if (r.code == 0 || r.code == KU_AMBIG) continue;
        const uint32_t end = j + 1 < c ? runs[run_off[i] + j + 1].start : n;
        hits[r.code] += end - r.start;
      }
      const int st = resolve_uids_one(tax, map, hits, &calls[i]);
      if (st != KU_OK) { status[t] = st; msg[t] = "read " + std::to_string(i) + ": a code of the batch is not a UID of the map"; return; }
    }
  };
  if (n_threads == 1) work(0);
  else {
    std::vector<std::thread> team;
    for (uint32_t t = 0; t < n_threads; ++t) team.emplace_back(work, t);
    for (auto &th : team) th.join();
  }
  for (uint32_t t = 0; t < n_threads; ++t)
    if (status[t] != KU_OK) { ku_set_error(msg[t]); return status[t]; }
  return KU_OK;
}

// ---------------------------------------------------------------------------- Kraken lines
// decimal digits, two at a time from a table; the values of a Kraken line (taxids, lengths, run lengths) are 32-bit and
// mostly short: 32-bit arithmetic, digit count from comparisons, written front to back
static const char ku_digit_pairs[201] =
    "00010203040506070809101112131415161718192021222324252627282930313233343536373839404142434445464748495051525354555657585960616263"
    "646566676869707172737475767778798081828384858687888990919293949596979899";
static inline char *put_u32(char *p, uint32_t v) {
  const int n = v < 10 ? 1 : v < 100 ? 2 : v < 1000 ? 3 : v < 10000 ? 4 : v < 100000 ? 5 : v < 1000000 ? 6 : v < 10000000 ? 7 : v < 100000000 ? 8
                : v < 1000000000 ? 9 : 10;
  char *e = p + n;
  while (v >= 100) {
    const uint32_t q = v / 100, r = v - q * 100;
    e -= 2;
    memcpy(e, ku_digit_pairs + 2 * r, 2);
    v = q;
  }
  if (v >= 10) memcpy(e - 2, ku_digit_pairs + 2 * v, 2);
  else e[-1] = (char)('0' + v);
  return p + n;
}
static inline char *put_u64(char *p, uint64_t v) {
  if (v <= 0xFFFFFFFFull) return put_u32(p, (uint32_t)v);
  char tmp[24];
  int n = 0;
  do { tmp[n++] = (char)('0' + v % 10); v /= 10; } while (v);
  while (n) *p++ = tmp[--n];
  return p;
}

extern "C" size_t ku_hitlist_string(const uint32_t *taxa, size_t n, char *buf) {
  // classify.cpp:826-861; "0:0" for a read without k-mers (:994-995)
  if (n == 0) { memcpy(buf, "0:0", 3); return 3; }
  char *p = buf;
  size_t i = 0;
  while (i < n) {
    size_t j = i + 1;
    while (j < n && taxa[j] == taxa[i]) ++j;
    if (taxa[i] == KU_AMBIG) *p++ = 'A'; else p = put_u64(p, taxa[i]);
    *p++ = ':';
    p = put_u64(p, j - i);
    if (j < n) *p++ = ' ';
    i = j;
  }
  return (size_t)(p - buf);
}

extern "C" int ku_format_kraken(const char *seqs, const uint64_t *seq_off, const uint32_t *seq_len, uint64_t n_reads,
                                const char *ids, uint32_t k, const uint32_t *calls, const uint32_t *taxa,
                                const uint32_t *hits, uint32_t flags, char **out, size_t *out_len) {
  if (!out || !out_len || (n_reads && (!seq_off || !seq_len || !ids || !calls))) { ku_set_error("ku_format_kraken: null argument"); return KU_EINVAL; }
  const bool quick = flags & KU_P_QUICK, only_c = flags & KU_P_ONLY_CLASSIFIED, pseq = flags & KU_P_SEQUENCE;
  if ((!quick && !taxa) || (quick && !hits) || (pseq && !seqs)) { ku_set_error("ku_format_kraken: missing array for the requested columns"); return KU_EINVAL; }
  size_t cap = 1 << 16, len = 0;
  char *buf = (char *)malloc(cap);
  if (!buf) return KU_ENOMEM;
  const char *id = ids;
  for (uint64_t r = 0; r < n_reads; ++r) {
    size_t idl = strlen(id);
    const uint32_t L = seq_len[r];
    const size_t n = L >= k ? L - k + 1 : 0;
    size_t need = idl + 64 + 24 * n + (pseq ? L + 1 : 0);
    if (len + need > cap) {
      while (len + need > cap) cap *= 2;
      char *nb = (char *)realloc(buf, cap);
      if (!nb) { free(buf); return KU_ENOMEM; }
      buf = nb;
    }
    const uint32_t call = calls[r];
    if (!(call == 0 && only_c)) {  // classify.cpp:980-987
      char *p = buf + len;
      *p++ = call ? 'C' : 'U';
      *p++ = '\t';
      memcpy(p, id, idl); p += idl;
      *p++ = '\t';
      p = put_u64(p, call);
      *p++ = '\t';
      p = put_u64(p, L);
      *p++ = '\t';
      if (quick) { *p++ = 'Q'; *p++ = ':'; p = put_u64(p, hits[r]); }
      else p += ku_hitlist_string(taxa + seq_off[r], n, p);
      if (pseq) { *p++ = '\t'; memcpy(p, seqs + seq_off[r], L); p += L; }
      *p++ = '\n';
      len = (size_t)(p - buf);
    }
    id += idl + 1;
  }
  *out = buf;
  *out_len = len;
  return KU_OK;
}

extern "C" int ku_format_kraken_rle(const char *seqs, const uint64_t *seq_off, const uint32_t *seq_len, uint64_t n_reads,
                                    const char *ids, uint32_t k, const uint32_t *calls, const ku_run *runs,
                                    const uint64_t *run_off, const uint32_t *run_cnt, const uint32_t *hits,
                                    uint32_t flags, char **out, size_t *out_len) {
  if (!out || !out_len || (n_reads && (!seq_off || !seq_len || !ids || !calls))) { ku_set_error("ku_format_kraken_rle: null argument"); return KU_EINVAL; }
  const bool quick = flags & KU_P_QUICK, only_c = flags & KU_P_ONLY_CLASSIFIED, pseq = flags & KU_P_SEQUENCE;
  if ((!quick && n_reads && (!run_off || !run_cnt)) || (quick && !hits) || (pseq && !seqs)) { ku_set_error("ku_format_kraken_rle: missing array for the requested columns"); return KU_EINVAL; }
  // one allocation for the usual case (about 100 bytes per line; grows if the ids or hit lists are longer)
  size_t cap = std::max<size_t>(1 << 16, (size_t)n_reads * 128), len = 0;
  char *buf = (char *)malloc(cap);
  if (!buf) return KU_ENOMEM;
  const char *id = ids;
  for (uint64_t r = 0; r < n_reads; ++r) {
    const size_t idl = strlen(id);
    const uint32_t L = seq_len[r];
    const uint32_t n = L >= k ? L - k + 1 : 0;
    const uint32_t nr = quick ? 0 : run_cnt[r];
    if (nr == 0xFFFFFFFFu || (nr && !runs)) { free(buf); ku_set_error("ku_format_kraken_rle: read " + std::to_string(r) + " has no valid runs"); return KU_EINVAL; }
    size_t need = idl + 64 + 24 * (size_t)nr + (pseq ? L + 1 : 0);
    if (len + need > cap) {
      while (len + need > cap) cap *= 2;
      char *nb = (char *)realloc(buf, cap);
      if (!nb) { free(buf); return KU_ENOMEM; }
      buf = nb;
    }
    const uint32_t call = calls[r];
    if (!(call == 0 && only_c)) {
      char *p = buf + len;
      *p++ = call ? 'C' : 'U';
      *p++ = '\t';
      memcpy(p, id, idl); p += idl;
      *p++ = '\t';
      p = put_u64(p, call);
      *p++ = '\t';
      p = put_u64(p, L);
      *p++ = '\t';
      if (quick) { *p++ = 'Q'; *p++ = ':'; p = put_u64(p, hits[r]); }
      else if (n == 0) { memcpy(p, "0:0", 3); p += 3; }
      else {
        const ku_run *rr = runs + run_off[r];
        for (uint32_t j = 0; j < nr; ++j) {
          const uint32_t end = j + 1 < nr ? rr[j + 1].start : n;
          if (rr[j].code == KU_AMBIG) *p++ = 'A'; else p = put_u64(p, rr[j].code);
          *p++ = ':';
          p = put_u64(p, end - rr[j].start);
          if (j + 1 < nr) *p++ = ' ';
        }
      }
      if (pseq) { *p++ = '\t'; memcpy(p, seqs + seq_off[r], L); p += L; }
      *p++ = '\n';
      len = (size_t)(p - buf);
    }
    id += idl + 1;
  }
  *out = buf;
  *out_len = len;
  return KU_OK;
}

extern "C" void ku_free(void *p) { free(p); }

// ---------------------------------------------------------------------------- report
namespace {
struct Clade {
  uint64_t reads = 0, kmers = 0, uniq = 0;
  std::vector<uint8_t> regs;  // dense p=12 registers of the merged sketch (empty until first k-mer source)
  bool present = false;
  bool dense = false;         // a member's sketch was dense: the merged sketch is (hyperloglogplus.cpp:586-665)
  std::vector<uint32_t> set;  // else: union of the members' encoded hashes
};
struct Sb {
  std::string s;
  void printf(const char *fmt, ...) __attribute__((format(printf, 2, 3))) {
    char tmp[512];
    va_list ap;
    va_start(ap, fmt);
    int n = vsnprintf(tmp, sizeof(tmp), fmt, ap);
    va_end(ap);
    if (n > 0) s.append(tmp, (size_t)std::min<int>(n, (int)sizeof(tmp) - 1));
  }
};
}  // namespace

extern "C" int ku_report(const ku_tax *tax, const char *counts_path, const uint32_t *slot_taxid, const uint64_t *n_kmers,
                         const uint8_t *registers, uint64_t n_slots, const uint32_t *node_taxid, const uint64_t *n_reads,
                         uint64_t n_nodes, char **out, size_t *out_len) {
  return ku_report_multi(tax, counts_path ? &counts_path : nullptr, counts_path ? 1u : 0u, slot_taxid, n_kmers, registers,
                         n_slots, node_taxid, n_reads, n_nodes, out, out_len);
}

static int report_impl(const ku_tax *tax, const char *const *counts_paths, uint32_t n_paths, const uint32_t *slot_taxid,
                       const uint64_t *n_kmers, const uint8_t *registers, const uint64_t *unique, uint64_t n_slots,
                       const uint32_t *node_taxid, const uint64_t *n_reads, uint64_t n_nodes, char **out, size_t *out_len,
                       const uint8_t *slot_is_sparse = nullptr, const uint64_t *sparse_pairs = nullptr, uint64_t n_pairs = 0);

extern "C" int ku_report_sparse(const ku_tax *tax, const char *const *counts_paths, uint32_t n_paths, const uint32_t *slot_taxid,
                                const uint64_t *n_kmers, const uint8_t *registers, const uint8_t *slot_is_sparse,
                                const uint64_t *sparse_pairs, uint64_t n_pairs, uint64_t n_slots, const uint32_t *node_taxid,
                                const uint64_t *n_reads, uint64_t n_nodes, char **out, size_t *out_len) {
  if ((n_slots && (!registers || !slot_is_sparse)) || (n_pairs && !sparse_pairs)) { ku_set_error("ku_report_sparse: null argument"); return KU_EINVAL; }
  return report_impl(tax, counts_paths, n_paths, slot_taxid, n_kmers, registers, nullptr, n_slots, node_taxid, n_reads, n_nodes, out, out_len,
                     slot_is_sparse, sparse_pairs, n_pairs);
}

extern "C" int ku_report_multi(const ku_tax *tax, const char *const *counts_paths, uint32_t n_paths,
                               const uint32_t *slot_taxid, const uint64_t *n_kmers, const uint8_t *registers,
                               uint64_t n_slots, const uint32_t *node_taxid, const uint64_t *n_reads, uint64_t n_nodes,
                               char **out, size_t *out_len) {
  if (n_slots && !registers) { ku_set_error("ku_report: null argument"); return KU_EINVAL; }
  return report_impl(tax, counts_paths, n_paths, slot_taxid, n_kmers, registers, nullptr, n_slots, node_taxid, n_reads, n_nodes, out, out_len);
}

extern "C" int ku_report_exact(const ku_tax *tax, const char *const *counts_paths, uint32_t n_paths,
                               const uint32_t *slot_taxid, const uint64_t *n_kmers, const uint64_t *unique_kmers,
                               uint64_t n_slots, const uint32_t *node_taxid, const uint64_t *n_reads, uint64_t n_nodes,
                               char **out, size_t *out_len) {
  if (n_slots && !unique_kmers) { ku_set_error("ku_report_exact: null argument"); return KU_EINVAL; }
  return report_impl(tax, counts_paths, n_paths, slot_taxid, n_kmers, nullptr, unique_kmers, n_slots, node_taxid, n_reads, n_nodes, out, out_len);
}

// The report text from per-entry clade summaries (rows parallel to the taxDB entries): DFS from the roots, children by
// descending (reads, kmers) -- TaxReport::printReport (taxdb.hpp:1004-1123).  `clade_uniq` is the distinct k-mer count
// (estimate) of the clade's merged sketch.
extern "C" int ku_report_rows(const ku_tax *tax, const char *const *counts_paths, uint32_t n_paths, const uint8_t *present,
                              const uint64_t *clade_reads, const uint64_t *tax_reads, const uint64_t *clade_kmers,
                              const uint64_t *clade_uniq, uint64_t n_rows, char **out, size_t *out_len) {
  return ku_report_rows_cols(tax, counts_paths, n_paths, present, clade_reads, tax_reads, clade_kmers, clade_uniq, n_rows, 0u, out, out_len);
}

// ... with the reference's other column set: KU_R_NO_KMER_COLS = "% reads taxReads taxID rank taxName", what `classify -p 0`
// prints (HLL_PRECISION <= 0, classify.cpp:289,316-323); rows and their order are the same (children by reads, then by the
// number of k-mers: readcounts.hpp:90-98), clade_uniq is not looked at
extern "C" int ku_report_rows_cols(const ku_tax *tax, const char *const *counts_paths, uint32_t n_paths, const uint8_t *present,
                                   const uint64_t *clade_reads, const uint64_t *tax_reads, const uint64_t *clade_kmers,
                                   const uint64_t *clade_uniq, uint64_t n_rows, uint32_t flags, char **out, size_t *out_len) {
  if (!tax || !out || !out_len || (n_paths && !counts_paths)) { ku_set_error("ku_report_rows: null argument"); return KU_EINVAL; }
  const bool six = (flags & KU_R_NO_KMER_COLS) != 0;
  const size_t nt = tax->ids.size();
  if (n_rows != nt || (nt && (!present || !clade_reads || !tax_reads || !clade_kmers || (!six && !clade_uniq)))) {
    ku_set_error("ku_report_rows: the arrays must have one element per taxDB entry");
    return KU_EINVAL;
  }
  // genome sizes: readGenomeSizes (taxdb.hpp:867-885).  "while(!eof){in >> id >> size; set(id,size);}" applies
  // the LAST pair twice when the file ends in whitespace (the failed extraction leaves both values unchanged).
  std::vector<uint64_t> gsize(nt, 0), gchild(nt, 0);
  auto set_genome_size = [&](uint32_t id, uint64_t size) {  // taxdb.hpp:850-865
    auto it = tax->row.find(id);
    if (it == tax->row.end()) return;
    gsize[it->second] += size;
    for (int64_t q = tax->parent_row(it->second); q >= 0; q = tax->parent_row((size_t)q)) gchild[q] += size;
  };
  for (uint32_t pi = 0; pi < n_paths; ++pi) {  // one counts file per database, in order (classify.cpp:263-285)
    const char *counts_path = counts_paths[pi];
    if (!counts_path) continue;
    FILE *f = fopen(counts_path, "r");
    if (!f) { ku_set_error(std::string("unable to open file ") + counts_path); return KU_ENOINPUT; }
    std::string data;
    char tmp[65536];
    size_t got;
    while ((got = fread(tmp, 1, sizeof(tmp), f)) > 0) data.append(tmp, got);
    fclose(f);
    const char *p = data.c_str();
    unsigned long long id = 0, size = 0;
    bool have = false, trailing = false;
    for (;;) {
      char *e1, *e2;
      unsigned long long a = strtoull(p, &e1, 10);
      if (e1 == p) break;
      unsigned long long b = strtoull(e1, &e2, 10);
      if (e2 == e1) break;
      id = a; size = b; have = true; p = e2;
      trailing = *p != 0;
      set_genome_size((uint32_t)id, size);
    }
    if (have && trailing) set_genome_size((uint32_t)id, size);
  }
  // children lists
  std::vector<std::vector<uint32_t>> kids(nt);
  for (size_t i = 0; i < nt; ++i) { int64_t p = tax->parent_row(i); if (p >= 0) kids[p].push_back((uint32_t)i); }
  uint64_t total = 0;
  const uint32_t roots[3] = {0u, 1u, 0xFFFFFFFFu};  // taxdb.hpp:1004
  for (uint32_t r : roots) { auto it = tax->row.find(r); if (it != tax->row.end() && present[it->second]) total += clade_reads[it->second]; }
  Sb sb;
  if (total) {
    sb.s += six ? "%\treads\ttaxReads\ttaxID\trank\ttaxName\n"                       // classify.cpp:316-323
                : "%\treads\ttaxReads\tkmers\tdup\tcov\ttaxID\trank\ttaxName\n";  // classify.cpp:305-314
    // iterative DFS (taxdb.hpp:1049-1076)
    struct Frame { uint32_t row; unsigned depth; };
    std::vector<Frame> stack;
    for (int i = 2; i >= 0; --i) { auto it = tax->row.find(roots[i]); if (it != tax->row.end()) stack.push_back({it->second, 0}); }
    while (!stack.empty()) {
      Frame fr = stack.back();
      stack.pop_back();
      const uint32_t r = fr.row;
      if (!present[r] || clade_reads[r] == 0) continue;
      sb.printf("%.4g\t", 100.0 * double(clade_reads[r]) / double(total));
      if (six) {
        sb.printf("%llu\t%llu\t", (unsigned long long)clade_reads[r], (unsigned long long)tax_reads[r]);
      } else {
        const uint64_t uniq = clade_uniq[r];
        volatile double gs = double(gsize[r] + gchild[r]);
        volatile double kc = double(clade_kmers[r]), un = double(uniq);
        sb.printf("%llu\t%llu\t%llu\t", (unsigned long long)clade_reads[r], (unsigned long long)tax_reads[r], (unsigned long long)uniq);
        sb.printf("%.3g\t", kc / un);
        if (gs == 0) sb.s += "NA\t"; else sb.printf("%.4g\t", un / gs);
      }
      if (tax->ids[r] == 0xFFFFFFFFu) sb.s += "-1\t"; else sb.printf("%d\t", (int32_t)tax->ids[r]);
      sb.s += tax->ranks[r];
      sb.s += '\t';
      sb.s.append(2 * fr.depth, ' ');
      sb.s += tax->names[r];
      sb.s += '\n';
      std::vector<uint32_t> ch;
      for (uint32_t x : kids[r]) if (present[x]) ch.push_back(x);
      // descending (reads, kmers) (readcounts.hpp:90-98); ties: ascending taxid
      std::sort(ch.begin(), ch.end(), [&](uint32_t a, uint32_t b) {
        if (clade_reads[a] != clade_reads[b]) return clade_reads[a] > clade_reads[b];
        if (clade_kmers[a] != clade_kmers[b]) return clade_kmers[a] > clade_kmers[b];
        return tax->ids[a] < tax->ids[b];
      });
      for (auto it = ch.rbegin(); it != ch.rend(); ++it) stack.push_back({*it, fr.depth + 1});
    }
  }
  char *buf = (char *)malloc(sb.s.size() + 1);
  if (!buf) return KU_ENOMEM;
  memcpy(buf, sb.s.c_str(), sb.s.size() + 1);
  *out = buf;
  *out_len = sb.s.size();
  return KU_OK;
}

// HLL estimate from a register histogram computed on the device (ku_ctx_report): dense p = 12 sketches (bins 0 .. 53)
// and sparse p' = 25 ones (bins 1 .. 79 = ranks of the distinct encoded hashes; bin 0 is filled in here)
uint64_t ku_hll_estimate_hist(const uint32_t *bins, bool sparse, uint64_t n_observed) {
  int C[80];
  if (!sparse) {
    const uint32_t q = 64 - KU_HLL_P;
    for (uint32_t i = 0; i <= q + 1; ++i) C[i] = (int)bins[i];
    return ertl_from_histogram(C, q, double(KU_HLL_M), n_observed);
  }
  int64_t m = 1ll << 25;
  for (int i = 1; i < 80; ++i) { C[i] = (int)bins[i]; m -= bins[i]; }
  C[0] = (int)m;
  return ertl_from_histogram(C, 64 - 25, double(1 << 25), n_observed);
}

static int report_impl(const ku_tax *tax, const char *const *counts_paths, uint32_t n_paths, const uint32_t *slot_taxid,
                       const uint64_t *n_kmers, const uint8_t *registers, const uint64_t *unique, uint64_t n_slots,
                       const uint32_t *node_taxid, const uint64_t *n_reads, uint64_t n_nodes, char **out, size_t *out_len,
                       const uint8_t *slot_is_sparse, const uint64_t *sparse_pairs, uint64_t n_pairs) {
  if (!tax || !out || (n_paths && !counts_paths) || !out_len || (n_slots && (!slot_taxid || !n_kmers || (!registers && !unique))) || (n_nodes && (!node_taxid || !n_reads))) {
    ku_set_error("ku_report: null argument");
    return KU_EINVAL;
  }
  const size_t nt = tax->ids.size();
  // taxon_counts: taxid -> (n_reads, n_kmers, sketch); an entry exists when either count is non-zero
  // (exact mode, classifyExact: the sketch is a set of k-mers; a k-mer has one database value, so the sets of
  // different taxa are disjoint and a clade's distinct count is the sum of its members')
  struct TaxCount {
    uint64_t reads = 0, kmers = 0, uniq = 0;
    const uint8_t *regs = nullptr;
    bool sparse = false;                   // the taxon's sketch stayed in the sparse representation
    const uint32_t *enc = nullptr;         // ... its encoded hashes
    uint64_t n_enc = 0;
  };
  std::unordered_map<uint32_t, TaxCount> tc;
  // sparse sketches: (slot << 32 | encoded hash) pairs, grouped by slot
  std::vector<uint64_t> sorted_pairs;
  std::vector<uint32_t> enc_of;
  std::vector<uint64_t> enc_begin(n_slots + 1, 0);
  if (slot_is_sparse && !unique) {
    sorted_pairs.assign(sparse_pairs, sparse_pairs + n_pairs);
    std::sort(sorted_pairs.begin(), sorted_pairs.end());
    enc_of.resize(n_pairs);
    uint64_t at = 0;
    for (uint64_t s = 0; s < n_slots; ++s) {
      enc_begin[s] = at;
      while (at < n_pairs && (sorted_pairs[at] >> 32) == s) { enc_of[at] = (uint32_t)sorted_pairs[at]; ++at; }
    }
    enc_begin[n_slots] = at;
  }
  for (uint64_t s = 0; s < n_slots; ++s)
    if (n_kmers[s]) {
      auto &e = tc[slot_taxid[s]];
      e.kmers = n_kmers[s];
      if (unique) e.uniq = unique[s]; else e.regs = registers + s * KU_HLL_M;
      if (slot_is_sparse && !unique && slot_is_sparse[s]) {
        e.sparse = true;
        e.enc = enc_of.data() + enc_begin[s];
        e.n_enc = enc_begin[s + 1] - enc_begin[s];
      }
    }
  for (uint64_t i = 0; i < n_nodes; ++i)
    if (n_reads[i]) tc[node_taxid[i]].reads = n_reads[i];
  // clade roll-up (taxdb.hpp:928-973): every counted taxon contributes to itself and all ancestors
  std::vector<Clade> clade(nt);
  for (auto &kv : tc) {
    auto it = tax->row.find(kv.first);
    if (it == tax->row.end()) continue;  // "No entry for X in database!"
    for (int64_t q = it->second; q >= 0; q = tax->parent_row((size_t)q)) {
      Clade &c = clade[q];
      c.present = true;
      c.reads += kv.second.reads;
      c.kmers += kv.second.kmers;
      c.uniq += kv.second.uniq;
      if (kv.second.regs) {
        if (c.regs.empty()) c.regs.assign(kv.second.regs, kv.second.regs + KU_HLL_M);
        else for (int i = 0; i < KU_HLL_M; ++i) c.regs[i] = std::max(c.regs[i], kv.second.regs[i]);
        // HyperLogLogPlusMinus::merge (hyperloglogplus.cpp:586-665): a dense operand makes the result dense (the
        // registers above are exact either way: folding a sparse list into registers is lossless, :559-577)
        if (!kv.second.sparse) c.dense = true;
      }
    }
  }
  if (slot_is_sparse && !unique) {  // sparse + sparse = set union, whatever its size (:601-604)
    for (auto &kv : tc) {
      if (!kv.second.sparse || kv.second.n_enc == 0) continue;
      auto it = tax->row.find(kv.first);
      if (it == tax->row.end()) continue;
      for (int64_t q = it->second; q >= 0; q = tax->parent_row((size_t)q))
        if (!clade[q].dense) clade[q].set.insert(clade[q].set.end(), kv.second.enc, kv.second.enc + kv.second.n_enc);
    }
    for (Clade &c : clade)
      if (!c.dense && !c.set.empty()) {
        std::sort(c.set.begin(), c.set.end());
        c.set.erase(std::unique(c.set.begin(), c.set.end()), c.set.end());
      }
  }
  // estimator per present clade, then the text
  static const std::vector<uint8_t> zero_regs(KU_HLL_M, 0);
  std::vector<uint8_t> present(nt, 0);
  std::vector<uint64_t> c_reads(nt, 0), t_reads(nt, 0), c_kmers(nt, 0), c_uniq(nt, 0);
  for (size_t r = 0; r < nt; ++r) {
    const Clade &c = clade[r];
    if (!c.present) continue;
    present[r] = 1;
    c_reads[r] = c.reads;
    c_kmers[r] = c.kmers;
    auto ti = tc.find(tax->ids[r]);
    t_reads[r] = ti == tc.end() ? 0 : ti->second.reads;
    if (c.reads == 0) continue;  // not printed
    const uint8_t *regs = c.regs.empty() ? zero_regs.data() : c.regs.data();
    const bool sparse_sketch = slot_is_sparse && !unique && !c.dense;
    c_uniq[r] = unique ? c.uniq
                       : (sparse_sketch ? ku_hll_cardinality_sparse(c.set.data(), c.set.size(), c.kmers)
                                        : ku_hll_cardinality(regs, KU_HLL_P, c.kmers));
  }
  return ku_report_rows(tax, counts_paths, n_paths, present.data(), c_reads.data(), t_reads.data(), c_kmers.data(), c_uniq.data(), nt,
                        out, out_len);
}
