// Drop-in for the reference's `db_sort` (src/db_sort.cpp): same getopt string "n:d:o:i:t:zM", same files in and
// out; the binning, sorting and indexing run on the GPU through ku_db_sort_files (-t and -M are accepted and have
// nothing left to do).
#include <getopt.h>
#include <sysexits.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>

#include "../../include/krakenuniq_amd.h"

static void usage(int code) {
  fprintf(stderr, "Usage: db_sort [-z] [-M] [-t threads] [-n nt] <-d input db> <-o output db> <-i output idx>\n");
  exit(code);
}

int main(int argc, char **argv) {
  std::string in, out, idx;
  long long nt = 15;  // Bin_key_nt default (src/db_sort.cpp:28)
  bool zero = false;
  if (argc > 1 && strcmp(argv[1], "-h") == 0) usage(0);
  int opt;
  while ((opt = getopt(argc, argv, "n:d:o:i:t:zM")) != -1) {
    switch (opt) {
      case 'n':
        nt = atoll(optarg);
        if (nt < 1 || nt > 31) { fprintf(stderr, "db_sort: bin key length out of range\n"); return EX_USAGE; }
        break;
      case 'd': in = optarg; break;
      case 'o': out = optarg; break;
      case 'i': idx = optarg; break;
      case 'M': break;
      case 't':
        if (atoll(optarg) <= 0) { fprintf(stderr, "db_sort: can't use nonpositive thread count\n"); return EX_USAGE; }
        break;
      case 'z': zero = true; break;
      default: usage(EX_USAGE);
    }
  }
  if (in.empty() || out.empty() || idx.empty()) usage(EX_USAGE);
  fprintf(stderr, "db_sort: Getting database into memory ...");
  const char *dev_env = getenv("KU_DEVICE");
  int st = ku_db_sort_files(dev_env ? atoi(dev_env) : 0, in.c_str(), out.c_str(), idx.c_str(), (uint32_t)nt, zero ? 1 : 0);
  if (st != KU_OK) {
    fprintf(stderr, "\ndb_sort: %s\n", ku_last_error());
    return st == KU_EINVAL ? EX_USAGE : st == KU_EDATA ? EX_DATAERR : st == KU_ENOINPUT ? EX_NOINPUT
           : st == KU_ENOMEM ? EX_OSERR : EX_SOFTWARE;
  }
  fprintf(stderr, "db_sort: Sorting complete - writing database to disk ...\n");
  return 0;
}
