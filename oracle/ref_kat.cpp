// oracle/ref_kat.cpp -- TEST INFRASTRUCTURE ONLY.
//
// Known-answer harness: this file is OUR code; it is linked against the
// reference's own objects (krakendb.o krakenutil.o quickfile.o
// hyperloglogplus.o, compiled by oracle/Makefile from /root/reference/src where
// they lie) and exposes the reference's hot-path functions through a line
// protocol on stdin/stdout so tests/golden/make_golden.py can capture
// known-answer vectors.  The binary lands in oracle/_ref/ (git-ignored).
//
// Reference entry points exercised (file:line under /root/reference/src):
//   KmerScanner::next_kmer / ambig_kmer        krakenutil.cpp:237-282
//   KrakenDB::canonical_representation          krakendb.cpp:238-246
//   KrakenDB::reverse_complement                krakendb.cpp:218-235
//   KrakenDB::bin_key (1- and 2-arg)            krakendb.cpp:182-215
//   KrakenDB::kmer_query                        krakendb.cpp:250-326
//   kraken::lca / resolve_tree                  krakenutil.cpp:90-118,149-200
//   murmurhash3_finalizer                       hyperloglogplus.cpp:830-838
//   HyperLogLogPlusMinus<uint64_t>              hyperloglogplus.cpp:485-801
//   kraken::resolve_uids3 / get_taxids_for_uid  uid_mapping.cpp:212-311
//
// Protocol (one command per line, answers on one line each unless noted):
//   K <k>                      set k (once); builds an in-memory JFLISTDN header
//   IDX <type 1|2> <nt>        fabricate an index header (KRAKIDX / KRAKIX2)
//   SCAN <seq>                 -> "n" then n lines "fwdhex ambig canonhex"
//   CANON <hex> <n>            -> hex
//   RC <hex> <n>               -> hex
//   BINKEY <hex> <nt>          -> dec       (2-arg form, always XOR-scrambled)
//   BINKEY1 <hex>              -> dec       (1-arg form, uses IDX type/nt)
//   HASH <hex>                 -> hex
//   OPEN <kdb> <idx>           mmap a real DB
//   QUERY <hex>                -> taxid or -1   (stateless 1-arg kmer_query)
//   PARENT a:b a:b ...         set parent map
//   LCA a b                    -> taxid
//   RESOLVE t:c t:c ...        -> taxid
//   HNEW <id> <p> <sparse>     new sketch
//   HUSEN <id> <0|1>           use_n_observed
//   HINS <id> <hex>
//   HSEQ <id> <n> <multhex> <start>   insert (start+i)*mult, i in [0,n)
//   HMERGE <dst> <src>
//   HCARD <id>                 -> "ertl heule flajolet nobs sparse listsize"
//   HDUMP <id>                 -> "S v v v ..." sorted encoded list, or "D r r r ..." registers
//   UIDMAP t:p t:p ...         the UID-to-taxid map file as set_lcas -I writes it: block i = {taxid, parent uid} of uid i+1
//   UMORDER k k k ...          -> the keys in the iteration order of a std::unordered_map<uint32_t, uint32_t> they were
//                              put into (operator[]) in the order given
//   UIDTAXIDS <uid>            -> the taxids of the uid (get_taxids_for_uid)
//   UIDRESOLVE u u u ...       the non-zero DB values (UIDs) of a read's k-mers in k-mer order: hit_counts[u]++ in that
//                              order into a std::unordered_map as classify_sequence does (classify.cpp:939-942), then
//                              resolve_uids3 with the PARENT map -> taxid
#define private public   // harness only: dump sketch state (M / sparseList)
#include "hyperloglogplus.hpp"
#undef private
#include "krakendb.hpp"
#include "krakenutil.hpp"
#include "quickfile.hpp"
#include "uid_mapping.hpp"
#include <algorithm>
#include <cinttypes>
#include <iostream>
#include <map>
#include <sstream>

using namespace std;
using namespace kraken;

static char kdb_hdr[4096];
static char idx_hdr[16];
static KrakenDB *fake_db = NULL;
static KrakenDBIndex *fake_idx = NULL;
static KrakenDB *real_db = NULL;
static KrakenDBIndex *real_idx = NULL;
static QuickFile kdb_file, idx_file;
static unordered_map<uint32_t, uint32_t> parent_map;
static map<int, HyperLogLogPlusMinus<uint64_t> *> sketches;
static vector<uint32_t> uid_blob;  // {taxid, parent uid} pairs

static uint64_t hex64(const string &s) { return strtoull(s.c_str(), NULL, 16); }

static void parse_pairs(istringstream &ss, unordered_map<uint32_t, uint32_t> &m) {
  string tok;
  while (ss >> tok) {
    size_t c = tok.find(':');
    m[(uint32_t)strtoul(tok.substr(0, c).c_str(), NULL, 10)] =
        (uint32_t)strtoul(tok.substr(c + 1).c_str(), NULL, 10);
  }
}

int main() {
  ios::sync_with_stdio(false);
  string line;
  while (getline(cin, line)) {
    istringstream ss(line);
    string cmd;
    if (!(ss >> cmd)) continue;
    if (cmd == "K") {
      int k; ss >> k;
      memset(kdb_hdr, 0, sizeof(kdb_hdr));
      memcpy(kdb_hdr, "JFLISTDN", 8);
      uint64_t key_bits = 2 * k, val_len = 4, key_ct = 0;
      memcpy(kdb_hdr + 8, &key_bits, 8);
      memcpy(kdb_hdr + 16, &val_len, 8);
      memcpy(kdb_hdr + 48, &key_ct, 8);
      fake_db = new KrakenDB(kdb_hdr);
      KmerScanner::set_k(k);
      cout << "ok\n";
    } else if (cmd == "IDX") {
      int type, nt; ss >> type >> nt;
      memcpy(idx_hdr, type == 1 ? "KRAKIDX" : "KRAKIX2", 7);
      idx_hdr[7] = (char)nt;
      fake_idx = new KrakenDBIndex(idx_hdr);
      fake_db->set_index(fake_idx);
      cout << "ok\n";
    } else if (cmd == "SCAN") {
      string seq; ss >> seq;
      vector<string> out;
      if (seq.size() >= KmerScanner::get_k()) {
        KmerScanner scanner(seq);
        uint64_t *kp;
        while ((kp = scanner.next_kmer()) != NULL) {
          char buf[96];
          snprintf(buf, sizeof(buf), "%016" PRIx64 " %d %016" PRIx64, *kp,
                   (int)scanner.ambig_kmer(), fake_db->canonical_representation(*kp));
          out.push_back(buf);
        }
      }
      cout << out.size() << "\n";
      for (auto &s : out) cout << s << "\n";
    } else if (cmd == "CANON") {
      string h; int n; ss >> h >> n;
      printf("%016" PRIx64 "\n", fake_db->canonical_representation(hex64(h), n)); fflush(stdout);
    } else if (cmd == "RC") {
      string h; int n; ss >> h >> n;
      printf("%016" PRIx64 "\n", fake_db->reverse_complement(hex64(h), n)); fflush(stdout);
    } else if (cmd == "BINKEY") {
      string h; int nt; ss >> h >> nt;
      cout << fake_db->bin_key(hex64(h), nt) << "\n";
    } else if (cmd == "BINKEY1") {
      string h; ss >> h;
      cout << fake_db->bin_key(hex64(h)) << "\n";
    } else if (cmd == "HASH") {
      string h; ss >> h;
      printf("%016" PRIx64 "\n", murmurhash3_finalizer(hex64(h))); fflush(stdout);
    } else if (cmd == "OPEN") {
      string a, b; ss >> a >> b;
      kdb_file.open_file(a);
      idx_file.open_file(b);
      real_db = new KrakenDB(kdb_file.ptr());
      real_idx = new KrakenDBIndex(idx_file.ptr());
      real_db->set_index(real_idx);
      cout << real_db->get_key_ct() << " " << (int)real_db->get_k() << " "
           << (int)real_idx->indexed_nt() << " " << (int)real_idx->index_type() << "\n";
    } else if (cmd == "QUERY") {
      string h; ss >> h;
      uint32_t *v = real_db->kmer_query(hex64(h));
      if (v) cout << *v << "\n"; else cout << -1 << "\n";
    } else if (cmd == "PARENT") {
      parent_map.clear();
      parse_pairs(ss, parent_map);
      cout << "ok\n";
    } else if (cmd == "LCA") {
      uint32_t a, b; ss >> a >> b;
      cout << lca(parent_map, a, b) << "\n";
    } else if (cmd == "RESOLVE") {
      unordered_map<uint32_t, uint32_t> hits;
      parse_pairs(ss, hits);
      cout << resolve_tree(hits, parent_map) << "\n";
    } else if (cmd == "HNEW") {
      int id, p, sp; ss >> id >> p >> sp;
      sketches[id] = new HyperLogLogPlusMinus<uint64_t>(p, sp != 0);
      cout << "ok\n";
    } else if (cmd == "HUSEN") {
      int id, v; ss >> id >> v;
      sketches[id]->use_n_observed = (v != 0);
      cout << "ok\n";
    } else if (cmd == "HINS") {
      int id; string h; ss >> id >> h;
      sketches[id]->insert(hex64(h));
      cout << "ok\n";
    } else if (cmd == "HSEQ") {
      int id; uint64_t n, start; string m; ss >> id >> n >> m >> start;
      uint64_t mult = hex64(m);
      for (uint64_t i = 0; i < n; ++i) sketches[id]->insert((start + i) * mult);
      cout << "ok\n";
    } else if (cmd == "HMERGE") {
      int d, s; ss >> d >> s;
      sketches[d]->merge(*sketches[s]);
      cout << "ok\n";
    } else if (cmd == "HCARD") {
      int id; ss >> id;
      auto *h = sketches[id];
      cout << h->ertlCardinality() << " " << h->heuleCardinality() << " "
           << h->flajoletCardinality() << " " << h->nObserved() << " " << (int)h->sparse
           << " " << h->sparseList.size() << "\n";
    } else if (cmd == "HDUMP") {
      int id; ss >> id;
      auto *h = sketches[id];
      if (h->sparse) {
        vector<uint32_t> v(h->sparseList.begin(), h->sparseList.end());
        sort(v.begin(), v.end());
        cout << "S";
        for (auto x : v) cout << " " << x;
        cout << "\n";
      } else {
        cout << "D";
        for (auto x : h->M) cout << " " << (int)x;
        cout << "\n";
      }
    } else if (cmd == "UIDMAP") {
      uid_blob.clear();
      string tok;
      while (ss >> tok) {
        size_t c = tok.find(':');
        uid_blob.push_back((uint32_t)strtoul(tok.substr(0, c).c_str(), NULL, 10));
        uid_blob.push_back((uint32_t)strtoul(tok.substr(c + 1).c_str(), NULL, 10));
      }
      cout << "ok\n";
    } else if (cmd == "UMORDER") {
      unordered_map<uint32_t, uint32_t> m;
      uint32_t u;
      while (ss >> u) m[u]++;
      bool first = true;
      for (auto it = m.begin(); it != m.end(); ++it) { cout << (first ? "" : " ") << it->first; first = false; }
      cout << "\n";
    } else if (cmd == "UIDTAXIDS") {
      uint32_t u; ss >> u;
      vector<uint32_t> t = get_taxids_for_uid(u, (const char *)uid_blob.data());
      for (size_t i = 0; i < t.size(); ++i) cout << (i ? " " : "") << t[i];
      cout << "\n";
    } else if (cmd == "UIDRESOLVE") {
      unordered_map<uint32_t, uint32_t> hits;
      unordered_map<uint32_t, vector<uint32_t> > dict;
      uint32_t u;
      while (ss >> u) hits[u]++;
      cout << resolve_uids3(hits, parent_map, dict, (const char *)uid_blob.data(), uid_blob.size() * 4) << "\n";
    } else {
      cout << "ERR unknown command " << cmd << "\n";
    }
    cout.flush();
  }
  _exit(0);  // skip destructors (KrakenDB::~KrakenDB munmaps an unset pointer)
}
