// Drop-in for the reference's `set_lcas` (src/set_lcas.cpp): same getopt string, same files; the per-k-mer LCA fold
// runs on the GPU through ku_setlcas_*.  Built: -d -i -b -o -x -f / -F -m -c -T -R -E -p -v -a -A -I (-t and -M are accepted
// and have nothing left to do; -I: UID databases, the set numbering of the reference on one thread).  Unlike the reference without -M, the
// input database file is never modified when -o names another file.
#include <getopt.h>
#include <sysexits.h>
#include <unistd.h>

#include <cctype>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <map>
#include <sstream>
#include <string>
#include <unordered_map>
#include <vector>

#include "../../include/krakenuniq_amd.h"
#include "ku_seqio.h"

void ku_seqio::fatal(int code, const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  fprintf(stderr, "set_lcas: ");
  vfprintf(stderr, fmt, ap);
  fprintf(stderr, "\n");
  va_end(ap);
  exit(code);
}
using ku_seqio::fatal;

static int exit_code_of(int st) {
  switch (st) {
    case KU_EINVAL: return EX_USAGE;
    case KU_EDATA: return EX_DATAERR;
    case KU_ENOINPUT: return EX_NOINPUT;
    case KU_ENOMEM: return EX_OSERR;
    default: return EX_SOFTWARE;
  }
}
#define CHECK(expr)                                                  \
  do {                                                               \
    int st_ = (expr);                                                \
    if (st_ != KU_OK) fatal(exit_code_of(st_), "%s", ku_last_error()); \
  } while (0)

static void usage(int code) {
  fprintf(stderr,
          "Usage: set_lcas [options]\n\nOptions: (*mandatory)\n"
          "* -d filename      Kraken DB filename\n"
          "* -i filename      Kraken DB index filename\n"
          "* -b filename      Taxonomy DB file\n"
          "  -t #             Number of threads\n"
          "  -M               Copy DB to RAM during operation\n"
          "  -o filename      Output database to filename, instead of overwriting the input database\n"
          "  -x               K-mers not found in DB do not cause errors\n"
          "  -f filename      File to taxon map\n"
          "  -F filename      Multi-FASTA file with sequence data\n"
          "  -m filename      Sequence ID to taxon map\n"
          "  -a               Add taxonomy IDs (starting with 1000000001) for sequences to the Taxonomy DB\n"
          "  -A               Add taxonomy IDs for assemblies (third column in seqid2taxid.map) to the Taxonomy DB\n"
          "  -T               When a k-mer appears in a 'synthetic construct' sequence, force the taxID to be the\n"
          "                   'synthetic construct' taxID, instead of the LCA.\n"
          "  -R               Reset the taxID of the k-mers of the sequences to zero\n"
          "  -E #             Exclude sequences that are shorter than the threshold.\n"
          "  -c filename      Write the k-mer counts per taxon (database.kdb.counts)\n"
          "  -p               Pretend - do not write database back to disk\n"
          "  -v               Verbose output\n"
          "  -h               Print this message\n\n"
          "-F and -m must be specified together.  If -f is given, -F/-m are ignored.\n");
  exit(code);
}

// one FASTA record at a time through the classify executable's reader
struct Fasta {
  ku_seqio::Reader rd;
  ku_seqio::Batch bt;
  std::string header;
  explicit Fasta(const std::string &path) {
    bt.pinned = false;  // ku_setlcas_add takes ordinary host memory
    rd.open(path.c_str(), /*prefetch=*/true);
    rd.fastq = false;  // FastaReader regardless of the first byte (src/set_lcas.cpp:245,422)
  }
  ~Fasta() { bt.release(); }
  bool next(std::string &id, const char *&seq, size_t &len) {
    size_t n = 0, lo, hi;
    bt.clear();
    bt.begin_read();
    if (!ku_seqio::next_record(rd, bt, &header, nullptr, &n)) return false;
    bt.end_read();
    ku_seqio::split_id(header.data(), header.size(), lo, hi);
    id.assign(header, lo, hi - lo);
    seq = bt.seqs + bt.off.back();
    len = bt.len.back();
    return true;
  }
};

// taxDB as set_lcas needs it: readable, extendable (-a / -A add entries) and writable (TaxonomyDB, src/taxdb.hpp)
struct TaxTable {
  struct Entry { uint32_t file_parent; std::string name, rank; };
  std::map<uint32_t, Entry> entries;  // ascending = the order writeTaxonomyIndex prints (taxdb.hpp:533-547)
  void read(const std::string &path) {  // readTaxonomyIndex_ (taxdb.hpp:563-605): "id <ws> parent \t name \t rank"
    std::ifstream f(path);
    if (!f) fatal(EX_NOINPUT, "unable to open taxonomy index file %s", path.c_str());
    std::string line;
    while (std::getline(f, line)) {
      if (line.empty()) continue;
      char *end;
      const char *p = line.c_str();
      unsigned long id = strtoul(p, &end, 10);
      if (end == p) continue;
      p = end;
      unsigned long par = strtoul(p, &end, 10);
      if (end == p) continue;
      p = end;
      if (*p) ++p;
      const char *tab = strchr(p, '\t');
      Entry e{(uint32_t)par, tab ? std::string(p, tab - p) : std::string(p), tab ? std::string(tab + 1) : std::string()};
      if ((uint32_t)id > 1 && id == par) fatal(EX_DATAERR, "taxDB: the parent of %lu is itself", id);  // taxdb.hpp:583-586
      entries.emplace((uint32_t)id, e);
    }
    entries.emplace(0u, Entry{0u, "unclassified", "no rank"});  // taxdb.hpp:599
  }
  bool has(uint32_t id) const { return entries.count(id) != 0; }
  // Parent_map value (getParentMap, taxdb.hpp:383-398): the parent pointer's taxid, 0 without one
  uint32_t parent_of(uint32_t id) const {
    auto it = entries.find(id);
    if (it == entries.end()) return 0;
    const uint32_t p = it->second.file_parent;
    return (p != id && entries.count(p)) ? p : 0;
  }
  bool insert(uint32_t id, uint32_t parent, const std::string &rank, const std::string &name) {  // taxdb.hpp:713-734
    if (parent == id) return false;
    if (!entries.count(parent)) {
      fprintf(stderr, "ERROR with taxon [%u;%s;%s] - parent taxon %u not in database!\n", id, rank.c_str(), name.c_str(), parent);
      return false;
    }
    return entries.emplace(id, Entry{parent, name, rank}).second;
  }
  void write(const std::string &path) const {  // writeTaxonomyIndex: an entry without parent pointer prints its own id
    FILE *f = fopen(path.c_str(), "w");
    if (!f) fatal(EX_OSERR, "can't write %s", path.c_str());
    for (const auto &kv : entries) {
      const uint32_t p = parent_of(kv.first);
      const bool has_ptr = kv.second.file_parent != kv.first && entries.count(kv.second.file_parent);
      fprintf(f, "%u\t%u\t%s\t%s\n", kv.first, has_ptr ? p : kv.first, kv.second.name.c_str(), kv.second.rank.c_str());
    }
    fclose(f);
  }
};

static const uint32_t TID_HUMAN = 9606, TID_MOUSE = 10090;  // no sequence taxids for host genomes (src/set_lcas.cpp:83-85)

int main(int argc, char **argv) {
  std::string db_name, idx_name, out_name, taxdb_name, counts_name, file_map_name, id_map_name, fasta_name, uid_map_name;
  bool force_contaminant = false, reset = false, allow_extra = false, verbose = false, pretend = false;
  bool add_for_sequences = false, add_for_assembly = false;  // -a, -A (src/set_lcas.cpp:528-533)
  uint32_t min_size = 0;
  if (argc > 1 && strcmp(argv[1], "-h") == 0) usage(0);
  int opt;
  while ((opt = getopt(argc, argv, "f:d:i:t:n:m:F:xMTRvb:aApI:o:Sc:E:")) != -1) {
    switch (opt) {
      case 'f': file_map_name = optarg; break;
      case 'd': db_name = optarg; break;
      case 'i': idx_name = optarg; break;
      case 'F': fasta_name = optarg; break;
      case 'm': id_map_name = optarg; break;
      case 't': if (atoll(optarg) <= 0) fatal(EX_USAGE, "can't use nonpositive thread count"); break;
      case 'T': force_contaminant = true; break;
      case 'R': reset = true; break;
      case 'v': verbose = true; break;
      case 'x': allow_extra = true; break;
      case 'b': taxdb_name = optarg; break;
      case 'c': counts_name = optarg; break;
      case 'M': case 'n': case 'S': break;
      case 'o': out_name = optarg; break;
      case 'E': min_size = (uint32_t)atoi(optarg); break;
      case 'p': pretend = true; break;
      case 'a': add_for_sequences = true; break;
      case 'A': add_for_assembly = true; break;
      case 'I': uid_map_name = optarg; break;
      default: usage(EX_USAGE);
    }
  }
  if (db_name.empty() || idx_name.empty() || taxdb_name.empty()) usage(EX_USAGE);
  if (file_map_name.empty() && (fasta_name.empty() || id_map_name.empty())) usage(EX_USAGE);
  const bool one_fasta = file_map_name.empty();

  // KU_SETLCAS_DRY=1 (test hook): sequence-ID map, FASTA headers and taxDB handling only; nothing is computed or
  // written for the database
  const bool dry = getenv("KU_SETLCAS_DRY") != nullptr;
  TaxTable tt;
  tt.read(taxdb_name);
  uint32_t new_taxid = 1000000000;  // New_taxid_start (src/set_lcas.cpp:50)
  std::unordered_map<std::string, uint32_t> name_to_taxid;  // assembly names and sequence IDs share it (:210-236)
  auto get_new_taxid = [&](const std::string &name, uint32_t parent, const char *rank) -> uint32_t {  // :169-189
    auto it = name_to_taxid.find(name);
    if (it != name_to_taxid.end()) return it->second;
    const uint32_t id = ++new_taxid;  // consumed even when the insert fails
    if (!tt.insert(id, parent, rank, name)) return 0;
    name_to_taxid[name] = id;
    return id;
  };

  std::unordered_map<std::string, uint32_t> id_to_taxon;
  if (one_fasta) {  // read_seqid_to_taxid_map (src/set_lcas.cpp:191-237)
    fprintf(stderr, "Reading sequence ID to taxonomy ID mapping ... ");
    if (add_for_assembly || add_for_sequences) {
      for (const auto &kv : tt.entries)
        if (kv.first >= new_taxid) new_taxid = kv.first + 100;
      fprintf(stderr, "[starting new taxonomy IDs with %u]", new_taxid + 1);
    }
    std::ifstream mf(id_map_name);
    if (!mf) fatal(EX_NOINPUT, "can't open %s", id_map_name.c_str());
    std::string line, seq_id, name;
    while (std::getline(mf, line)) {
      if (line.empty()) break;
      std::istringstream iss(line);
      uint32_t taxid = 0;
      iss >> seq_id >> taxid;
      if (id_to_taxon.count(seq_id)) continue;  // a sequence ID seen before is ignored
      const uint32_t orig_taxid = taxid;
      if (add_for_assembly && iss.good()) {
        iss.get();
        std::getline(iss, name);
        if (!name.empty()) taxid = get_new_taxid(name, taxid, "assembly");
      }
      if (add_for_sequences && orig_taxid != TID_HUMAN && orig_taxid != TID_MOUSE) taxid = get_new_taxid(seq_id, taxid, "sequence");
      if (add_for_assembly || add_for_sequences) printf("%s\t%u\n", seq_id.c_str(), taxid);
      id_to_taxon[seq_id] = taxid;
    }
    if (id_to_taxon.empty()) fprintf(stderr, "Error: No ID mappings present!!\n");
    fprintf(stderr, " got %zu mappings.\n", id_to_taxon.size());
  }

  ku_db *db = nullptr;
  ku_tax *tax = nullptr;
  ku_setlcas *sl = nullptr;
  ku_db_info info{};
  if (!dry) {
    CHECK(ku_db_open(db_name.c_str(), idx_name.c_str(), &db));
    CHECK(ku_db_get_info(db, &info));
    std::vector<uint32_t> ids, parents;  // the (extended) taxonomy as the library takes it
    for (const auto &kv : tt.entries) { ids.push_back(kv.first); parents.push_back(kv.second.file_parent); }
    CHECK(ku_tax_from_arrays(ids.data(), parents.data(), ids.size(), &tax));
    const char *dev_env = getenv("KU_DEVICE");
    CHECK(ku_setlcas_open(dev_env ? atoi(dev_env) : 0, db, tax, (reset ? KU_SL_RESET : 0u) | (force_contaminant ? KU_SL_FORCE_CONTAMINANT : 0u) | (uid_map_name.empty() ? 0u : KU_SL_UIDS), &sl));
  }
  // Parent_map membership (src/set_lcas.cpp:313-318,338): taxids with an entry in taxDB
  auto in_taxonomy = [&](uint32_t taxid) { return taxid != 0 && tt.has(taxid); };

  if (one_fasta) {  // process_single_file (src/set_lcas.cpp:239-366)
    Fasta fa(fasta_name);
    const std::string prefix = "kraken:taxid|";
    std::string id;
    const char *seq;
    size_t len;
    uint32_t processed = 0, skipped = 0;
    while (fa.next(id, seq, len)) {
      if (len == 0) { ++skipped; continue; }
      uint32_t taxid = 0;
      auto it = id_to_taxon.find(id);
      if (it != id_to_taxon.end()) taxid = it->second;
      else {  // the ID without a ".<digits>" version suffix
        size_t pos = id.find_last_of('.');
        bool num = pos != std::string::npos;
        for (size_t i = pos + 1; num && i < id.size(); ++i) num = isdigit((unsigned char)id[i]) != 0;
        if (num && (it = id_to_taxon.find(id.substr(0, pos))) != id_to_taxon.end()) taxid = it->second;
      }
      bool from_header = false;
      if (taxid == 0 && id.compare(0, prefix.size(), prefix) == 0) {
        from_header = true;
        taxid = (uint32_t)strtol(id.c_str() + prefix.size(), nullptr, 10);
        if (taxid == 0) fprintf(stderr, "Error: taxonomy ID is zero for sequence '%s'?!\n", id.c_str());
      }
      if (taxid == 0) {
        fprintf(stderr, "Error! Didn't find taxonomy ID mapping for sequence %s!!\n", id.c_str());
        ++skipped;
        continue;
      }
      if (min_size > 0 && len < min_size) {
        fprintf(stderr, "Skipping sequence %s as it's too short (%zu)\n", id.c_str(), len);
        ++skipped;
        continue;
      }
      if (!in_taxonomy(taxid)) {
        fprintf(stderr, "Skipping sequence %s since taxonomy ID %u is not in taxonomy database!\n", id.c_str(), taxid);
        ++skipped;
        continue;
      }
      if (add_for_sequences) {  // the entry takes the sequence's header line as its name (src/set_lcas.cpp:321-330)
        const uint32_t par = tt.parent_of(taxid);
        if (taxid != TID_HUMAN && par != TID_HUMAN && taxid != TID_MOUSE && par != TID_MOUSE) {
          std::string h = fa.header;
          if (from_header) { const size_t b0 = h.find_first_not_of("\t "); if (b0 != std::string::npos) h = h.substr(b0); }
          tt.entries[taxid].name = h;
        }
      }
      if (sl) CHECK(ku_setlcas_add(sl, seq, len, taxid));
      ++processed;
      if (verbose) fprintf(stderr, "\rProcessed %u sequences", processed);
    }
    fprintf(stderr, "\rFinished processing %u sequences (skipping %u empty sequences, and 0 sequences with no taxonomy mapping)\n", processed, skipped);
  } else {  // process_files (src/set_lcas.cpp:368-409): one single-FASTA file per line, "<filename> <taxid>"
    fprintf(stderr, "Processing files in %s\n", file_map_name.c_str());
    std::ifstream mf(file_map_name);
    if (!mf) fatal(EX_NOINPUT, "can't open %s", file_map_name.c_str());
    std::string line, filename, id;
    uint32_t processed = 0;
    while (std::getline(mf, line)) {
      if (line.empty()) break;
      std::istringstream iss(line);
      uint32_t taxid = 0;
      iss >> filename >> taxid;
      Fasta fa(filename);
      const char *seq;
      size_t len;
      if (fa.next(id, seq, len) && sl) CHECK(ku_setlcas_add(sl, seq, len, taxid));  // the first record only
      ++processed;
    }
    fprintf(stderr, "\rFinished processing %u sequences\n", processed);
  }

  // the rewritten taxDB goes out only once the run cannot fail on a missing k-mer any more: the reference aborts
  // inside its processing loop, before src/set_lcas.cpp:171-177
  auto write_taxdb = [&] {
    if ((add_for_sequences || add_for_assembly) && !pretend) {
      fprintf(stderr, "Writing new TaxDB ...\n");
      tt.write(taxdb_name);
    }
  };
  if (dry) {
    write_taxdb();
    return 0;
  }
  std::vector<uint32_t> values(info.key_ct + 1);
  uint64_t n_missing = 0;
  CHECK(ku_setlcas_finish(sl, values.data(), &n_missing));
  if (!uid_map_name.empty()) {  // the UID-to-taxid map (src/uid_mapping.cpp:85-88: {taxid, parent UID} per UID, binary)
    const uint32_t *blocks = nullptr;
    uint64_t n_uids = 0;
    CHECK(ku_setlcas_uid_map(sl, &blocks, &n_uids));
    FILE *uf = fopen(uid_map_name.c_str(), "wb");
    if (!uf) fatal(EX_OSERR, "Something went wrong while creating the file %s", uid_map_name.c_str());
    if (n_uids && fwrite(blocks, 8, n_uids, uf) != n_uids) fatal(EX_OSERR, "can't write %s", uid_map_name.c_str());
    fclose(uf);
  }
  ku_setlcas_close(sl);
  if (n_missing && !allow_extra) fatal(EX_DATAERR, "kmer found in sequence that is not in database");
  if (n_missing && verbose) fprintf(stderr, "%llu kmers found in sequences that are not in database\n", (unsigned long long)n_missing);
  write_taxdb();

  if (!counts_name.empty()) {  // KrakenDB::count_taxons (src/krakendb.cpp:90-113)
    fprintf(stderr, "Writing kmer counts to %s...\n", counts_name.c_str());
    std::map<uint32_t, uint64_t> counts;
    for (uint64_t i = 0; i < info.key_ct; ++i) ++counts[values[i]];
    FILE *cf = fopen(counts_name.c_str(), "w");
    if (!cf) fatal(EX_OSERR, "can't write %s", counts_name.c_str());
    for (const auto &kv : counts) fprintf(cf, "%u\t%llu\n", kv.first, (unsigned long long)kv.second);
    fclose(cf);
  }
  if (!pretend) {
    // header and keys as they are, new values: read the whole input before (possibly) overwriting it
    const std::string target = out_name.empty() ? db_name : out_name;
    fprintf(stderr, "Writing database from RAM back to %s ...\n", target.c_str());
    FILE *in = fopen(db_name.c_str(), "rb");
    if (!in) fatal(EX_NOINPUT, "can't open %s", db_name.c_str());
    fseek(in, 0, SEEK_END);
    const size_t sz = (size_t)ftell(in);
    fseek(in, 0, SEEK_SET);
    std::vector<char> dat(sz);
    if (fread(dat.data(), 1, sz, in) != sz) fatal(EX_OSERR, "can't read %s", db_name.c_str());
    fclose(in);
    const size_t ps = info.key_len + 4, hdr = 72 + 2 * (4 + 8 * 2 * (size_t)info.k);  // krakendb.cpp:177
    if (sz < hdr || info.key_ct > (sz - hdr) / ps) fatal(EX_DATAERR, "database file truncated");
    for (uint64_t i = 0; i < info.key_ct; ++i) memcpy(dat.data() + hdr + i * ps + info.key_len, &values[i], 4);
    ku_db_close(db);  // unmap before writing over the file
    db = nullptr;
    // never truncate the only copy of a database: the new image goes to a temporary file next to the target, is
    // flushed to disk, and only then takes the target's name
    const std::string tmp_name = target + ".tmp";
    FILE *out = fopen(tmp_name.c_str(), "wb");
    if (!out) fatal(EX_OSERR, "can't write %s", tmp_name.c_str());
    const bool ok = fwrite(dat.data(), 1, sz, out) == sz && fflush(out) == 0 && fsync(fileno(out)) == 0;
    if (fclose(out) != 0 || !ok) {
      remove(tmp_name.c_str());
      fatal(EX_OSERR, "can't write %s", tmp_name.c_str());
    }
    if (rename(tmp_name.c_str(), target.c_str()) != 0) {
      remove(tmp_name.c_str());
      fatal(EX_OSERR, "can't move %s over %s", tmp_name.c_str(), target.c_str());
    }
  }
  if (db) ku_db_close(db);
  ku_tax_close(tax);
  return 0;
}
