// ku_host.h -- host-side (no device code) pieces behind the C ABI: taxonomy object,
// HLL estimator, Kraken line formatting, report.  Not part of the public ABI.
#pragma once
#include <stdint.h>

#include <string>
#include <unordered_map>
#include <vector>

#include "../../include/krakenuniq_amd.h"

// taxDB in memory: entries as read + Parent_map (taxdb.hpp:563-605,411-433,383-398)
struct ku_tax {
  std::vector<uint32_t> ids;          // in file order; entry 0 "unclassified" always present
  std::vector<uint32_t> file_parent;  // parent column as read
  std::vector<uint32_t> parent_map;   // Parent_map value: 0 = no parent pointer (root / self / orphan)
  std::vector<std::string> names, ranks;
  std::unordered_map<uint32_t, uint32_t> row;  // taxid -> index
  void add(uint32_t id, uint32_t parent, const std::string &name, const std::string &rank) {
    if (row.count(id)) return;  // entries.insert keeps the first (taxdb.hpp:596)
    row[id] = (uint32_t)ids.size();
    ids.push_back(id);
    file_parent.push_back(parent);
    parent_map.push_back(0);
    names.push_back(name);
    ranks.push_back(rank);
  }
  // createPointers: the parent pointer exists iff parent id != own id and the parent has an entry
  void finish() {
    for (size_t i = 0; i < ids.size(); ++i) {
      uint32_t p = file_parent[i];
      parent_map[i] = (ids[i] != 0 && p != ids[i] && row.count(p)) ? p : 0;
    }
  }
  // row of the parent entry following TaxonomyEntry::parent, -1 if none
  int64_t parent_row(size_t i) const {
    uint32_t p = file_parent[i];
    if (p == ids[i]) return -1;
    auto it = row.find(p);
    return it == row.end() ? -1 : (int64_t)it->second;
  }
};

void ku_set_error(const std::string &s);
// Ertl estimate from a register histogram (80 bins) built on the device: dense p = 12 or sparse p' = 25 sketch
uint64_t ku_hll_estimate_hist(const uint32_t *bins, bool sparse, uint64_t n_observed);
