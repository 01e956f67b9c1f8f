// Drop-in for the reference's `set_lcas` (src/set_lcas.cpp): same getopt string, same files; the per-k-mer LCA fold
// runs on the GPU through ku_setlcas_*.  Built: -d -i -b -o -x -f / -F -m -c -T -R -E -p -v (-t and -M are accepted and
// have nothing left to do).  Not built (EX_SOFTWARE): -a / -A (new taxids for sequences / assemblies, which rewrite
// taxDB) and -I (UID databases).  Unlike the reference without -M, the input database file is never modified when -o
// names another file.
#include <getopt.h>
#include <sysexits.h>

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

int main(int argc, char **argv) {
  std::string db_name, idx_name, out_name, taxdb_name, counts_name, file_map_name, id_map_name, fasta_name;
  bool force_contaminant = false, reset = false, allow_extra = false, verbose = false, pretend = false;
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
      case 'a': case 'A': fatal(EX_SOFTWARE, "adding taxonomy IDs for sequences / assemblies (-a, -A) is not built into the MI355X set_lcas");
      case 'I': fatal(EX_SOFTWARE, "UID databases (-I) are not built into the MI355X set_lcas");
      default: usage(EX_USAGE);
    }
  }
  if (db_name.empty() || idx_name.empty() || taxdb_name.empty()) usage(EX_USAGE);
  if (file_map_name.empty() && (fasta_name.empty() || id_map_name.empty())) usage(EX_USAGE);
  const bool one_fasta = file_map_name.empty();

  ku_db *db = nullptr;
  ku_tax *tax = nullptr;
  CHECK(ku_db_open(db_name.c_str(), idx_name.c_str(), &db));
  CHECK(ku_tax_open(taxdb_name.c_str(), &tax));
  ku_db_info info;
  CHECK(ku_db_get_info(db, &info));
  ku_setlcas *sl = nullptr;
  const char *dev_env = getenv("KU_DEVICE");
  CHECK(ku_setlcas_open(dev_env ? atoi(dev_env) : 0, db, tax, (reset ? KU_SL_RESET : 0u) | (force_contaminant ? KU_SL_FORCE_CONTAMINANT : 0u), &sl));
  // Parent_map membership (src/set_lcas.cpp:313-318,338): taxids with an entry in taxDB
  auto in_taxonomy = [&](uint32_t taxid) { return taxid != 0 && ku_tax_parent(tax, taxid) != KU_AMBIG; };

  if (one_fasta) {  // process_single_file (src/set_lcas.cpp:239-366)
    fprintf(stderr, "Reading sequence ID to taxonomy ID mapping ... ");
    std::unordered_map<std::string, uint32_t> id_to_taxon;
    {
      std::ifstream mf(id_map_name);
      if (!mf) fatal(EX_NOINPUT, "can't open %s", id_map_name.c_str());
      std::string line, seq_id;
      while (std::getline(mf, line)) {
        if (line.empty()) break;
        std::istringstream iss(line);
        uint32_t taxid = 0;
        iss >> seq_id >> taxid;
        id_to_taxon.emplace(seq_id, taxid);  // a sequence ID seen before is ignored
      }
    }
    if (id_to_taxon.empty()) fprintf(stderr, "Error: No ID mappings present!!\n");
    fprintf(stderr, " got %zu mappings.\n", id_to_taxon.size());
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
      if (taxid == 0 && id.compare(0, prefix.size(), prefix) == 0) {
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
      CHECK(ku_setlcas_add(sl, seq, len, taxid));
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
      if (fa.next(id, seq, len)) CHECK(ku_setlcas_add(sl, seq, len, taxid));  // the first record only
      ++processed;
    }
    fprintf(stderr, "\rFinished processing %u sequences\n", processed);
  }

  std::vector<uint32_t> values(info.key_ct + 1);
  uint64_t n_missing = 0;
  CHECK(ku_setlcas_finish(sl, values.data(), &n_missing));
  ku_setlcas_close(sl);
  if (n_missing && !allow_extra) fatal(EX_DATAERR, "kmer found in sequence that is not in database");
  if (n_missing && verbose) fprintf(stderr, "%llu kmers found in sequences that are not in database\n", (unsigned long long)n_missing);

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
    if (sz < hdr + info.key_ct * ps) fatal(EX_DATAERR, "database file truncated");
    for (uint64_t i = 0; i < info.key_ct; ++i) memcpy(dat.data() + hdr + i * ps + info.key_len, &values[i], 4);
    ku_db_close(db);  // unmap before writing over the file
    db = nullptr;
    FILE *out = fopen(target.c_str(), "wb");
    if (!out || fwrite(dat.data(), 1, sz, out) != sz) fatal(EX_OSERR, "can't write %s", target.c_str());
    fclose(out);
  }
  if (db) ku_db_close(db);
  ku_tax_close(tax);
  return 0;
}
