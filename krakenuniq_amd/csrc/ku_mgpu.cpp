// ku_mgpu.cpp -- the classify hot path over several GPUs: host C++ over the C ABI + RCCL (include/krakenuniq_amd.h,
// "several GPUs").  One host thread per rank; the database is sharded by minimizer-bin range (the reference's
// --preload-size partitioning, krakendb.cpp:430-526, laid out in space instead of in time).  Default exchange: OWNER
// ROUTING (rank_step_routed) -- a rank scans its slice of the reads, runs of k-mers travel as 16-byte records to the rank
// that owns their bin, one slot per k-mer comes back.  Kept beside it: the position-wise exchange -- read batches are
// broadcast, per-k-mer slots are max-reduced ("non-zero wins", classify.cpp:445-452) and scattered over the read dimension.
// Either way every rank resolves its slice (classify.cpp:676-785), the per-taxon state is reduced at the end of the run.
//
// Two exchange back ends behind one small interface (Comm):
//   RCCL      ncclBroadcast / grouped ncclReduce (a reduce-scatter whose slices end on read boundaries) / ncclAllReduce /
//             ncclAllGather on the rank's stream.  librccl is bound with dlopen at the first use, so single-GPU users
//             of the library never load it and a process that already holds an RCCL (PyTorch ships one) shares it.
//   same-process copies + merge kernels, host barriers between the ranks' threads: ranks that share a device (RCCL
//             admits one rank per device) -- the form the 1-GPU test box can run -- or KU_MGPU_NO_RCCL.
#include <dlfcn.h>
#include <pthread.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <set>
#include <string>
#include <thread>
#include <vector>

#include "ku_internal.h"

static int mfail(int code, const std::string &msg) {
  ku_set_error(msg);
  return code;
}
#define M_HIP(expr)                                                                                          \
  do {                                                                                                       \
    hipError_t e_ = (expr);                                                                                  \
    if (e_ != hipSuccess)                                                                                    \
      return mfail(e_ == hipErrorOutOfMemory ? KU_ENOMEM : KU_EHIP, std::string(#expr) + ": " + hipGetErrorString(e_)); \
  } while (0)
#define M_TRY(expr)             \
  do {                          \
    int s_ = (expr);            \
    if (s_ != KU_OK) return s_; \
  } while (0)

// ---------------------------------------------------------------------------- RCCL, bound at run time
namespace {
struct Rccl {
  void *h = nullptr;
  decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
  decltype(&ncclCommInitRank) CommInitRank = nullptr;
  decltype(&ncclCommInitAll) CommInitAll = nullptr;
  decltype(&ncclCommDestroy) CommDestroy = nullptr;
  decltype(&ncclBroadcast) Broadcast = nullptr;
  decltype(&ncclReduce) Reduce = nullptr;
  decltype(&ncclSend) Send = nullptr;
  decltype(&ncclRecv) Recv = nullptr;
  decltype(&ncclAllReduce) AllReduce = nullptr;
  decltype(&ncclAllGather) AllGather = nullptr;
  decltype(&ncclGroupStart) GroupStart = nullptr;
  decltype(&ncclGroupEnd) GroupEnd = nullptr;
  decltype(&ncclGetErrorString) GetErrorString = nullptr;
};
Rccl g_rccl;
pthread_once_t g_rccl_once = PTHREAD_ONCE_INIT;
std::string g_rccl_err;

void rccl_load() {
#ifdef KU_TEST_HOOKS
  // Test builds only (tests/rccl_shim/Makefile links libkrakenuniq_amd_testhooks.so with -DKU_TEST_HOOKS; the product library does
  // not look at the variable): KU_RCCL_LIB=<path> binds that library instead -- the peer paths between processes on a 1-GPU box;
  // RTLD_LOCAL + the handle-first lookup of dlsym keep it apart from an RCCL the process already holds
  if (const char *over = getenv("KU_RCCL_LIB")) {
    g_rccl.h = dlopen(over, RTLD_NOW | RTLD_LOCAL);
    if (!g_rccl.h) {
      g_rccl_err = std::string("cannot load KU_RCCL_LIB=") + over + ": " + dlerror();
      return;
    }
  }
#endif
  for (const char *name : {"librccl.so.1", "/opt/rocm/lib/librccl.so.1", "librccl.so"}) {
    if (g_rccl.h) break;
    g_rccl.h = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
  }
  if (!g_rccl.h) {
    g_rccl_err = std::string("cannot load librccl: ") + dlerror();
    return;
  }
#define KU_SYM(field, sym)                                                     \
  g_rccl.field = reinterpret_cast<decltype(g_rccl.field)>(dlsym(g_rccl.h, sym)); \
  if (!g_rccl.field) g_rccl_err = std::string("librccl lacks ") + sym;
  KU_SYM(GetUniqueId, "ncclGetUniqueId")
  KU_SYM(CommInitRank, "ncclCommInitRank")
  KU_SYM(CommInitAll, "ncclCommInitAll")
  KU_SYM(CommDestroy, "ncclCommDestroy")
  KU_SYM(Broadcast, "ncclBroadcast")
  KU_SYM(Reduce, "ncclReduce")
  KU_SYM(Send, "ncclSend")
  KU_SYM(Recv, "ncclRecv")
  KU_SYM(AllReduce, "ncclAllReduce")
  KU_SYM(AllGather, "ncclAllGather")
  KU_SYM(GroupStart, "ncclGroupStart")
  KU_SYM(GroupEnd, "ncclGroupEnd")
  KU_SYM(GetErrorString, "ncclGetErrorString")
#undef KU_SYM
}
int rccl_ready() {
  pthread_once(&g_rccl_once, rccl_load);
  if (!g_rccl_err.empty()) return mfail(KU_EHIP, g_rccl_err);
  return KU_OK;
}
#define M_NCCL(expr)                                                                                     \
  do {                                                                                                   \
    ncclResult_t r_ = (expr);                                                                            \
    if (r_ != ncclSuccess) return mfail(KU_EHIP, std::string(#expr) + ": " + g_rccl.GetErrorString(r_)); \
  } while (0)

struct DBuf {
  void *p = nullptr;
  size_t cap = 0;
  int reserve(size_t bytes) {
    if (bytes <= cap) return KU_OK;
    if (p) (void)hipFree(p);
    p = nullptr;
    cap = 0;
    const size_t want = bytes + bytes / 8 + 256;
    if (hipMalloc(&p, want) != hipSuccess) {
      p = nullptr;
      return mfail(KU_ENOMEM, "device memory for a multi-GPU batch buffer");
    }
    cap = want;
    return KU_OK;
  }
  void release() {
    if (p) (void)hipFree(p);
    p = nullptr;
    cap = 0;
  }
};

// state the ranks of one process share for the same-process exchange
struct Shared {
  pthread_barrier_t bar;
  bool bar_init = false;
  std::vector<const void *> ptr_a, ptr_b, ptr_c;  // published per local rank
  std::vector<std::vector<uint32_t>> vals;
  std::vector<std::vector<uint64_t>> u64s;
  std::vector<const uint64_t *> offs;  // host offset tables published next to ptr_a
  std::vector<int> dev;
  std::atomic<int> failed{0};
};
}  // namespace

struct ku_mgpu {
  uint32_t world = 0, first_rank = 0, n_local = 0, flags = 0;
  bool use_rccl = false;
  struct Rank {
    uint32_t rank = 0, local = 0;
    int device = 0;
    ku_ctx *ctx = nullptr;
    ncclComm_t comm = nullptr;
    DBuf seqs, off, len, taxa, calls, hits, runs, roff, rcnt, scratch, small;
    // owner routing, two sets (the rounds of a step alternate between them, each set on a stream of its own): the send
    // queues (records) + their k-mer numbering, what this rank received + its numbering, the slots it found, the slots that
    // came back, scratch of the prefix sums, the small device tables of a round
    struct RouteSet {
      DBuf q_rec, q_kb, r_rec, r_kb, r_slots, ret_slots, pfx_work, rt_dev;
    } rs[2];
    // ku_mgpu_set_timing: event pairs of the last routed step, [stage 0 scan, 1 owner, 2 resolve], and what it received
    std::vector<hipEvent_t> tev_pool;
    std::vector<std::pair<int, std::pair<hipEvent_t, hipEvent_t>>> tev;
    size_t tev_used = 0;
    double t_rounds = 0, t_rec = 0, t_kmers = 0;
    hipStream_t aux = nullptr;            // the second set's stream
    hipEvent_t ev_a = nullptr, ev_b = nullptr, ev_r = nullptr;
    uint64_t n_runs = 0;   // runs of the last host batch still in `runs`
    uint64_t run_base = 0; // where they start in the caller's array
  };
  std::vector<Rank> ranks;
  Shared sh;
  bool loaded = false, tax_set = false;
  bool reduced = false;  // ku_mgpu_reduce_state ran and no batch was classified since: a second call would add the sums again
  // HyperLogLog++ sparse-mode emulation over the group (ku_mgpu_enable_sparse): every rank runs it on whole work units
  bool sparse = false;
  bool sparse_gave_up = false;       // the emulation was given up for the whole group (a rank ran out of memory)
  uint64_t unit_nt = 0, acc_nt = 0;  // -u, and the nt of the unit that is still open (classify.cpp:510-521)
  int open_rank = -1;                // the rank whose context holds that unit's state
  bool exact = false;                // classifyExact on the sharded group (ku_mgpu_enable_exact)
  // owner routing of the sharded path (set up with the taxonomy): every rank's minimizer range
  bool route = false;
  std::vector<uint64_t> own_lo, own_hi;
  bool timing = false;
};

namespace {
void barrier(ku_mgpu *m) {
  if (m->n_local > 1) pthread_barrier_wait(&m->sh.bar);
}

// run fn(rank) on one host thread per local rank (inline for a single rank); the first failing status wins
int run_all(ku_mgpu *m, const std::function<int(ku_mgpu::Rank &)> &fn) {
  std::vector<int> st(m->n_local, KU_OK);
  std::vector<std::string> msg(m->n_local);
  auto body = [&](uint32_t i) {
    ku_mgpu::Rank &r = m->ranks[i];
    if (hipSetDevice(r.device) != hipSuccess) {
      st[i] = KU_EHIP;
      msg[i] = "hipSetDevice failed";
      m->sh.failed.store(1);
      return;
    }
    st[i] = fn(r);
    if (st[i] != KU_OK) {
      msg[i] = ku_last_error();
      m->sh.failed.store(1);
    }
  };
  m->sh.failed.store(0);
  if (m->n_local == 1) {
    body(0);
  } else {
    std::vector<std::thread> team;
    for (uint32_t i = 0; i < m->n_local; ++i) team.emplace_back(body, i);
    for (auto &t : team) t.join();
  }
  for (uint32_t i = 0; i < m->n_local; ++i)
    if (st[i] != KU_OK) return mfail(st[i], "rank " + std::to_string(m->ranks[i].rank) + ": " + msg[i]);
  return KU_OK;
}

// ---- the exchange primitives.  Same-process variants always pass every barrier, whatever the local status, so that
// a failing rank cannot leave the others waiting; the status is checked behind the last barrier.
int copy_from_peer(ku_mgpu *m, ku_mgpu::Rank &r, void *dst, const void *src, size_t bytes, hipStream_t s) {
  (void)m;
  (void)r;
  if (bytes == 0) return KU_OK;
  M_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDefault, s));
  return KU_OK;
}

// same-process exchange: a rank that already failed says so before the first barrier, everybody leaves with an error
// behind the last one
void gate_in(ku_mgpu *m, int st) {
  if (st != KU_OK) m->sh.failed.store(1);
}
int gate_out(ku_mgpu *m, int st) {
  if (st == KU_OK && m->sh.failed.load()) return mfail(KU_ESTATE, "another rank of the group failed");
  return st;
}
bool comm_noop(const ku_mgpu *m) { return m->world == 1 && !m->use_rccl; }
// RCCL path: a rank that failed before a collective must not leave the others waiting inside it.  The ranks of one
// process agree on skipping it through the shared flag (several processes cannot, short of another collective: there a
// failed rank ends its process and the launcher takes the job down).
int rccl_gate(ku_mgpu *m, int st) {
  if (m->n_local > 1) {
    gate_in(m, st);
    barrier(m);
    st = gate_out(m, st);
  }
  return st;
}

int comm_broadcast(ku_mgpu *m, ku_mgpu::Rank &r, int st, void *buf, size_t bytes, hipStream_t s) {
  if (comm_noop(m) || bytes == 0) return st;
  if (m->use_rccl) {
    st = rccl_gate(m, st);
    if (st != KU_OK) return st;
    M_NCCL(g_rccl.Broadcast(buf, buf, bytes, ncclUint8, 0, r.comm, s));
    return KU_OK;
  }
  if (st == KU_OK && hipStreamSynchronize(s) != hipSuccess) st = mfail(KU_EHIP, "stream synchronisation failed");
  m->sh.ptr_a[r.local] = buf;
  gate_in(m, st);
  barrier(m);
  if (st == KU_OK && !m->sh.failed.load() && r.local != 0) {
    st = copy_from_peer(m, r, buf, m->sh.ptr_a[0], bytes, s);
    if (st == KU_OK && hipStreamSynchronize(s) != hipSuccess) st = mfail(KU_EHIP, "broadcast copy failed");
  }
  gate_in(m, st);
  barrier(m);
  return gate_out(m, st);
}

// in place: afterwards rank r holds max over all ranks of taxa[pos[r] .. pos[r+1])
int comm_reduce_slices_max(ku_mgpu *m, ku_mgpu::Rank &r, int st, uint32_t *taxa, const uint64_t *pos, hipStream_t s) {
  if (comm_noop(m)) return st;
  if (m->use_rccl) {
    st = rccl_gate(m, st);
    if (st != KU_OK) return st;
    const uint64_t lo = pos[r.rank], n_mine = pos[r.rank + 1] - lo;
    static const bool ring_reduce = getenv("KU_MGPU_EXCHANGE") && !strcmp(getenv("KU_MGPU_EXCHANGE"), "reduce");
    if (!ring_reduce) {
      // all-to-all over the point-to-point xGMI links: every rank sends slice q of its array straight to rank q (each
      // pair of GPUs has its own link, so the world - 1 transfers of a rank run side by side) and folds the world - 1
      // slices it receives into its own with the merge kernel.  A ring / tree ncclReduce per slice would push every
      // slice through every link.
      const uint32_t peers = m->world - 1;
      M_TRY(r.scratch.reserve(std::max<uint64_t>(1, (uint64_t)peers * n_mine) * 4));
      uint32_t *stage = (uint32_t *)r.scratch.p;
      M_NCCL(g_rccl.GroupStart());
      ncclResult_t e = ncclSuccess;
      uint32_t slot = 0;
      for (uint32_t q = 0; q < m->world && e == ncclSuccess; ++q) {
        if (q == r.rank) continue;
        const uint64_t n_q = pos[q + 1] - pos[q];
        if (n_q) e = g_rccl.Send(taxa + pos[q], n_q, ncclUint32, (int)q, r.comm, s);
        if (e == ncclSuccess && n_mine) e = g_rccl.Recv(stage + (uint64_t)slot * n_mine, n_mine, ncclUint32, (int)q, r.comm, s);
        ++slot;
      }
      if (e != ncclSuccess) {
        (void)g_rccl.GroupEnd();
        return mfail(KU_EHIP, std::string("ncclSend / ncclRecv: ") + g_rccl.GetErrorString(e));
      }
      M_NCCL(g_rccl.GroupEnd());
      for (uint32_t i = 0; i < peers && n_mine; ++i) M_TRY(ku_launch_merge_max_u32(taxa + lo, stage + (uint64_t)i * n_mine, n_mine, s));
      return KU_OK;
    }
    M_NCCL(g_rccl.GroupStart());
    for (uint32_t q = 0; q < m->world; ++q) {
      const uint64_t n = pos[q + 1] - pos[q];
      if (n == 0) continue;
      ncclResult_t e = g_rccl.Reduce(taxa + pos[q], taxa + pos[q], n, ncclUint32, ncclMax, (int)q, r.comm, s);
      if (e != ncclSuccess) {
        (void)g_rccl.GroupEnd();
        return mfail(KU_EHIP, std::string("ncclReduce: ") + g_rccl.GetErrorString(e));
      }
    }
    M_NCCL(g_rccl.GroupEnd());
    return KU_OK;
  }
  if (st == KU_OK && hipStreamSynchronize(s) != hipSuccess) st = mfail(KU_EHIP, "lookup failed");
  m->sh.ptr_a[r.local] = taxa;
  gate_in(m, st);
  barrier(m);
  const uint64_t lo = pos[r.rank], n = pos[r.rank + 1] - lo;
  if (!m->sh.failed.load()) {
    for (uint32_t q = 0; q < m->n_local && st == KU_OK && n; ++q) {
      if (q == r.local) continue;
      const uint32_t *src = (const uint32_t *)m->sh.ptr_a[q] + lo;
      if (m->sh.dev[q] != r.device) {  // stage the peer's slice on this device first
        st = r.scratch.reserve(n * 4);
        if (st == KU_OK) st = copy_from_peer(m, r, r.scratch.p, src, n * 4, s);
        src = (const uint32_t *)r.scratch.p;
      }
      if (st == KU_OK) st = ku_launch_merge_max_u32(taxa + lo, src, n, s);
    }
    if (st == KU_OK && hipStreamSynchronize(s) != hipSuccess) st = mfail(KU_EHIP, "slot merge failed");
  }
  gate_in(m, st);
  barrier(m);  // the peers may overwrite their arrays again
  return gate_out(m, st);
}

int comm_allreduce_state(ku_mgpu *m, ku_mgpu::Rank &r, hipStream_t s) {
  if (comm_noop(m)) return KU_OK;
  uint8_t *regs = nullptr;
  uint64_t n_regs = 0, n_slots = 0, n_nodes = 0, *nk = nullptr, *nr = nullptr;
  int st = ku_counts_device_ptrs(r.ctx, &regs, &n_regs, &nk, &n_slots, &nr, &n_nodes);
  if (m->use_rccl) {
    st = rccl_gate(m, st);
    if (st != KU_OK) return st;
    M_NCCL(g_rccl.GroupStart());
    ncclResult_t e1 = g_rccl.AllReduce(regs, regs, n_regs, ncclUint8, ncclMax, r.comm, s);
    ncclResult_t e2 = g_rccl.AllReduce(nk, nk, n_slots, ncclUint64, ncclSum, r.comm, s);
    ncclResult_t e3 = g_rccl.AllReduce(nr, nr, n_nodes, ncclUint64, ncclSum, r.comm, s);
    M_NCCL(g_rccl.GroupEnd());
    if (e1 != ncclSuccess || e2 != ncclSuccess || e3 != ncclSuccess) return mfail(KU_EHIP, "ncclAllReduce of the per-taxon state failed");
    return KU_OK;
  }
  if (st == KU_OK && hipStreamSynchronize(s) != hipSuccess) st = mfail(KU_EHIP, "stream synchronisation failed");
  m->sh.ptr_a[r.local] = regs;
  m->sh.ptr_b[r.local] = nk;
  m->sh.ptr_c[r.local] = nr;
  gate_in(m, st);
  barrier(m);
  if (r.local == 0 && !m->sh.failed.load()) {  // fold everybody into the first rank ...
    for (uint32_t q = 1; q < m->n_local && st == KU_OK; ++q) {
      const void *a = m->sh.ptr_a[q], *b = m->sh.ptr_b[q], *c = m->sh.ptr_c[q];
      if (m->sh.dev[q] != r.device) {
        st = r.scratch.reserve(n_regs + (n_slots + n_nodes) * 8);
        uint8_t *sp = (uint8_t *)r.scratch.p;
        if (st == KU_OK) st = copy_from_peer(m, r, sp, a, n_regs, s);
        if (st == KU_OK) st = copy_from_peer(m, r, sp + n_regs, b, n_slots * 8, s);
        if (st == KU_OK) st = copy_from_peer(m, r, sp + n_regs + n_slots * 8, c, n_nodes * 8, s);
        a = sp;
        b = sp + n_regs;
        c = sp + n_regs + n_slots * 8;
      }
      if (st == KU_OK) st = ku_launch_merge_max_u8(regs, (const uint8_t *)a, n_regs, s);
      if (st == KU_OK) st = ku_launch_merge_add_u64((unsigned long long *)nk, (const unsigned long long *)b, n_slots, s);
      if (st == KU_OK) st = ku_launch_merge_add_u64((unsigned long long *)nr, (const unsigned long long *)c, n_nodes, s);
      if (st == KU_OK && hipStreamSynchronize(s) != hipSuccess) st = mfail(KU_EHIP, "state merge failed");
    }
  }
  gate_in(m, st);
  barrier(m);
  if (r.local != 0 && st == KU_OK && !m->sh.failed.load()) {  // ... and hand the result back
    st = copy_from_peer(m, r, regs, m->sh.ptr_a[0], n_regs, s);
    if (st == KU_OK) st = copy_from_peer(m, r, nk, m->sh.ptr_b[0], n_slots * 8, s);
    if (st == KU_OK) st = copy_from_peer(m, r, nr, m->sh.ptr_c[0], n_nodes * 8, s);
    if (st == KU_OK && hipStreamSynchronize(s) != hipSuccess) st = mfail(KU_EHIP, "state copy failed");
  }
  gate_in(m, st);
  barrier(m);
  return gate_out(m, st);
}

// classifyExact over a sharded group: the ranks' k-mer sets are disjoint (a k-mer lives where its bin lives), the
// first-insertion counters per slot add up
int comm_allreduce_exact(ku_mgpu *m, ku_mgpu::Rank &r, hipStream_t s) {
  if (comm_noop(m)) return KU_OK;
  unsigned long long *uq = ku_ctx_exact_unique_of(r.ctx);
  ku_counts_dims d{};
  int st = uq ? ku_counts_dims_get(r.ctx, &d) : mfail(KU_ESTATE, "exact counting is not enabled on every rank");
  if (m->use_rccl) {
    st = rccl_gate(m, st);
    if (st != KU_OK) return st;
    M_NCCL(g_rccl.AllReduce(uq, uq, d.n_slots, ncclUint64, ncclSum, r.comm, s));
    return KU_OK;
  }
  if (st == KU_OK && hipStreamSynchronize(s) != hipSuccess) st = mfail(KU_EHIP, "stream synchronisation failed");
  m->sh.ptr_a[r.local] = uq;
  gate_in(m, st);
  barrier(m);
  if (r.local == 0 && !m->sh.failed.load()) {
    for (uint32_t q = 1; q < m->n_local && st == KU_OK; ++q) {
      const void *a = m->sh.ptr_a[q];
      if (m->sh.dev[q] != r.device) {
        st = r.scratch.reserve(d.n_slots * 8);
        if (st == KU_OK) st = copy_from_peer(m, r, r.scratch.p, a, d.n_slots * 8, s);
        a = r.scratch.p;
      }
      if (st == KU_OK) st = ku_launch_merge_add_u64(uq, (const unsigned long long *)a, d.n_slots, s);
      if (st == KU_OK && hipStreamSynchronize(s) != hipSuccess) st = mfail(KU_EHIP, "exact counter merge failed");
    }
  }
  gate_in(m, st);
  barrier(m);
  if (r.local != 0 && st == KU_OK && !m->sh.failed.load()) {
    st = copy_from_peer(m, r, uq, m->sh.ptr_a[0], d.n_slots * 8, s);
    if (st == KU_OK && hipStreamSynchronize(s) != hipSuccess) st = mfail(KU_EHIP, "exact counter copy failed");
  }
  gate_in(m, st);
  barrier(m);
  return gate_out(m, st);
}

// union of every rank's ascending value list
int comm_allgather_values(ku_mgpu *m, ku_mgpu::Rank &r, int st, const std::vector<uint32_t> &mine, std::vector<uint32_t> &all) {
  all = mine;
  if (comm_noop(m)) return st;
  if (m->use_rccl) {
    st = rccl_gate(m, st);
    if (st != KU_OK) return st;
    hipStream_t s = ku_ctx_stream_of(r.ctx);
    M_TRY(r.small.reserve(8ull * m->world + 8));
    unsigned long long mine_n = mine.size();
    unsigned long long *d_n = (unsigned long long *)r.small.p;
    M_HIP(hipMemcpyAsync(d_n + m->world, &mine_n, 8, hipMemcpyHostToDevice, s));
    M_NCCL(g_rccl.AllGather(d_n + m->world, d_n, 1, ncclUint64, r.comm, s));
    std::vector<unsigned long long> counts(m->world);
    M_HIP(hipMemcpyAsync(counts.data(), d_n, 8ull * m->world, hipMemcpyDeviceToHost, s));
    M_HIP(hipStreamSynchronize(s));
    const uint64_t mx = std::max<uint64_t>(1, *std::max_element(counts.begin(), counts.end()));
    M_TRY(r.scratch.reserve(4 * mx * (m->world + 1)));
    uint32_t *d_all = (uint32_t *)r.scratch.p, *d_mine = d_all + mx * m->world;
    M_HIP(hipMemsetAsync(d_mine, 0, 4 * mx, s));
    if (!mine.empty()) M_HIP(hipMemcpyAsync(d_mine, mine.data(), 4 * mine.size(), hipMemcpyHostToDevice, s));
    M_NCCL(g_rccl.AllGather(d_mine, d_all, mx, ncclUint32, r.comm, s));
    std::vector<uint32_t> buf(mx * m->world);
    M_HIP(hipMemcpyAsync(buf.data(), d_all, 4 * mx * m->world, hipMemcpyDeviceToHost, s));
    M_HIP(hipStreamSynchronize(s));
    all.clear();
    for (uint32_t q = 0; q < m->world; ++q) all.insert(all.end(), buf.begin() + q * mx, buf.begin() + q * mx + counts[q]);
  } else {
    m->sh.vals[r.local] = mine;
    gate_in(m, st);
    barrier(m);
    all.clear();
    for (uint32_t q = 0; q < m->n_local; ++q) all.insert(all.end(), m->sh.vals[q].begin(), m->sh.vals[q].end());
    barrier(m);
    st = gate_out(m, st);
  }
  std::sort(all.begin(), all.end());
  all.erase(std::unique(all.begin(), all.end()), all.end());
  return st;
}

// every rank's `n` numbers, rank-major, on every rank
int comm_allgather_u64(ku_mgpu *m, ku_mgpu::Rank &r, int st, const uint64_t *mine, uint32_t n, std::vector<uint64_t> &all, hipStream_t s) {
  all.assign((size_t)m->world * n, 0);
  if (comm_noop(m)) { std::copy(mine, mine + n, all.begin()); return st; }
  if (m->use_rccl) {
    st = rccl_gate(m, st);
    if (st != KU_OK) return st;
    M_TRY(r.small.reserve(8ull * n * (m->world + 1) + 64));
    unsigned long long *d_all = (unsigned long long *)r.small.p, *d_mine = d_all + (size_t)n * m->world;
    M_HIP(hipMemcpyAsync(d_mine, mine, 8ull * n, hipMemcpyHostToDevice, s));
    M_NCCL(g_rccl.AllGather(d_mine, d_all, n, ncclUint64, r.comm, s));
    M_HIP(hipMemcpyAsync(all.data(), d_all, 8ull * n * m->world, hipMemcpyDeviceToHost, s));
    M_HIP(hipStreamSynchronize(s));
    return KU_OK;
  }
  m->sh.u64s[r.local].assign(mine, mine + n);
  gate_in(m, st);
  barrier(m);
  for (uint32_t q = 0; q < m->n_local; ++q) std::copy(m->sh.u64s[q].begin(), m->sh.u64s[q].end(), all.begin() + (size_t)q * n);
  barrier(m);
  return gate_out(m, st);
}

// the same for numbers that lie in DEVICE memory (written by work queued on s): ONE host round trip -- over RCCL the
// all-gather runs on the device and one copy brings everything back
int comm_allgather_u64_dev(ku_mgpu *m, ku_mgpu::Rank &r, int st, const unsigned long long *d_mine, uint32_t n, std::vector<uint64_t> &all,
                           hipStream_t s) {
  all.assign((size_t)m->world * n, 0);
  if (m->use_rccl && !comm_noop(m)) {
    st = rccl_gate(m, st);
    if (st != KU_OK) return st;
    M_TRY(r.small.reserve(8ull * n * m->world + 64));
    M_NCCL(g_rccl.AllGather(d_mine, r.small.p, n, ncclUint64, r.comm, s));
    M_HIP(hipMemcpyAsync(all.data(), r.small.p, 8ull * n * m->world, hipMemcpyDeviceToHost, s));
    M_HIP(hipStreamSynchronize(s));
    return KU_OK;
  }
  std::vector<uint64_t> mine(n, 0);
  if (st == KU_OK && (hipMemcpyAsync(mine.data(), d_mine, 8ull * n, hipMemcpyDeviceToHost, s) != hipSuccess || hipStreamSynchronize(s) != hipSuccess))
    st = mfail(KU_EHIP, "routing counts copy failed");
  if (comm_noop(m)) { std::copy(mine.begin(), mine.end(), all.begin()); return st; }
  m->sh.u64s[r.local] = mine;
  gate_in(m, st);
  barrier(m);
  for (uint32_t q = 0; q < m->n_local; ++q) std::copy(m->sh.u64s[q].begin(), m->sh.u64s[q].end(), all.begin() + (size_t)q * n);
  barrier(m);
  return gate_out(m, st);
}

// all-to-all of variable-size segments: rank r sends send[send_off[q] .. send_off[q + 1]) (elements of `elem` bytes) to rank
// q and receives rank p's segment for it at recv[recv_off[p] ..).  Over xGMI every pair of GPUs has its own link: the
// world - 1 transfers of a rank run side by side (grouped ncclSend / ncclRecv).
int comm_alltoallv(ku_mgpu *m, ku_mgpu::Rank &r, int st, const void *send, const uint64_t *send_at, const uint64_t *send_n, void *recv,
                   const uint64_t *recv_at, const uint64_t *recv_n, size_t elem, hipStream_t s) {
  if (comm_noop(m)) {
    if (st == KU_OK && send_n[0] &&
        hipMemcpyAsync((char *)recv + recv_at[0] * elem, (const char *)send + send_at[0] * elem, send_n[0] * elem, hipMemcpyDeviceToDevice, s) != hipSuccess)
      st = mfail(KU_EHIP, "local copy failed");
    return st;
  }
  if (m->use_rccl) {
    st = rccl_gate(m, st);
    if (st != KU_OK) return st;
    M_NCCL(g_rccl.GroupStart());
    ncclResult_t e = ncclSuccess;
    for (uint32_t q = 0; q < m->world && e == ncclSuccess; ++q) {
      if (send_n[q]) e = g_rccl.Send((const char *)send + send_at[q] * elem, send_n[q] * elem, ncclUint8, (int)q, r.comm, s);
      if (e == ncclSuccess && recv_n[q]) e = g_rccl.Recv((char *)recv + recv_at[q] * elem, recv_n[q] * elem, ncclUint8, (int)q, r.comm, s);
    }
    if (e != ncclSuccess) {
      (void)g_rccl.GroupEnd();
      return mfail(KU_EHIP, std::string("ncclSend / ncclRecv: ") + g_rccl.GetErrorString(e));
    }
    M_NCCL(g_rccl.GroupEnd());
    return KU_OK;
  }
  if (st == KU_OK && hipStreamSynchronize(s) != hipSuccess) st = mfail(KU_EHIP, "stream synchronisation failed");
  m->sh.ptr_a[r.local] = send;
  m->sh.offs[r.local] = send_at;
  gate_in(m, st);
  barrier(m);
  if (!m->sh.failed.load()) {
    for (uint32_t p = 0; p < m->n_local && st == KU_OK; ++p)  // pull: rank p's segment for this rank
      if (recv_n[p])
        st = copy_from_peer(m, r, (char *)recv + recv_at[p] * elem, (const char *)m->sh.ptr_a[p] + m->sh.offs[p][r.rank] * elem, recv_n[p] * elem, s);
    if (st == KU_OK && hipStreamSynchronize(s) != hipSuccess) st = mfail(KU_EHIP, "all-to-all copy failed");
  }
  gate_in(m, st);
  barrier(m);  // the peers may reuse their send buffers
  return gate_out(m, st);
}

// rank 0 holds buf[0 .. bounds[world]) (elements of `elem` bytes); rank q receives its slice [bounds[q], bounds[q + 1]) in place
int comm_scatter_slices(ku_mgpu *m, ku_mgpu::Rank &r, int st, void *buf, const uint64_t *bounds, size_t elem, hipStream_t s) {
  if (comm_noop(m)) return st;
  const uint64_t lo = bounds[r.rank], n = bounds[r.rank + 1] - lo;
  if (m->use_rccl) {
    st = rccl_gate(m, st);
    if (st != KU_OK) return st;
    M_NCCL(g_rccl.GroupStart());
    ncclResult_t e = ncclSuccess;
    if (r.rank == 0) {
      for (uint32_t q = 1; q < m->world && e == ncclSuccess; ++q) {
        const uint64_t nq = bounds[q + 1] - bounds[q];
        if (nq) e = g_rccl.Send((const char *)buf + bounds[q] * elem, nq * elem, ncclUint8, (int)q, r.comm, s);
      }
    } else if (n) e = g_rccl.Recv((char *)buf + lo * elem, n * elem, ncclUint8, 0, r.comm, s);
    if (e != ncclSuccess) {
      (void)g_rccl.GroupEnd();
      return mfail(KU_EHIP, std::string("ncclSend / ncclRecv: ") + g_rccl.GetErrorString(e));
    }
    M_NCCL(g_rccl.GroupEnd());
    return KU_OK;
  }
  if (st == KU_OK && hipStreamSynchronize(s) != hipSuccess) st = mfail(KU_EHIP, "stream synchronisation failed");
  m->sh.ptr_a[r.local] = buf;
  gate_in(m, st);
  barrier(m);
  if (st == KU_OK && !m->sh.failed.load() && r.local != 0 && n) {
    st = copy_from_peer(m, r, (char *)buf + lo * elem, (const char *)m->sh.ptr_a[0] + lo * elem, n * elem, s);
    if (st == KU_OK && hipStreamSynchronize(s) != hipSuccess) st = mfail(KU_EHIP, "scatter copy failed");
  }
  gate_in(m, st);
  barrier(m);
  return gate_out(m, st);
}

bool single_process(const ku_mgpu *m) { return m->first_rank == 0 && m->n_local == m->world; }
}  // namespace

// ---------------------------------------------------------------------------- life cycle
extern "C" int ku_mgpu_unique_id(uint8_t *id) {
  if (!id) return mfail(KU_EINVAL, "ku_mgpu_unique_id: null argument");
  M_TRY(rccl_ready());
  static_assert(sizeof(ncclUniqueId) == KU_MGPU_ID_BYTES, "RCCL unique id size");
  ncclUniqueId u;
  M_NCCL(g_rccl.GetUniqueId(&u));
  memcpy(id, &u, sizeof u);
  return KU_OK;
}

extern "C" void ku_mgpu_destroy(ku_mgpu *m) {
  if (!m) return;
  for (auto &r : m->ranks) {
    (void)hipSetDevice(r.device);
    if (r.ctx) (void)ku_ctx_synchronize(r.ctx);
    if (r.comm && g_rccl.CommDestroy) (void)g_rccl.CommDestroy(r.comm);
    for (DBuf *b : {&r.seqs, &r.off, &r.len, &r.taxa, &r.calls, &r.hits, &r.runs, &r.roff, &r.rcnt, &r.scratch, &r.small})
      b->release();
    for (auto &t : r.rs)
      for (DBuf *b : {&t.q_rec, &t.q_kb, &t.r_rec, &t.r_kb, &t.r_slots, &t.ret_slots, &t.pfx_work, &t.rt_dev}) b->release();
    if (r.aux) (void)hipStreamDestroy(r.aux);
    for (hipEvent_t e : r.tev_pool) (void)hipEventDestroy(e);
    for (hipEvent_t e : {r.ev_a, r.ev_b, r.ev_r})
      if (e) (void)hipEventDestroy(e);
    if (r.ctx) ku_ctx_destroy(r.ctx);
  }
  if (m->sh.bar_init) pthread_barrier_destroy(&m->sh.bar);
  delete m;
}

extern "C" int ku_mgpu_create(const int *devices, uint32_t n_local, uint32_t first_rank, uint32_t world, const uint8_t *id,
                              uint32_t flags, ku_mgpu **out) {
  if (!out || !devices || n_local == 0 || world == 0 || first_rank + n_local > world)
    return mfail(KU_EINVAL, "ku_mgpu_create: bad rank layout");
  *out = nullptr;
  const bool all_here = first_rank == 0 && n_local == world;
  if (!all_here && !id) return mfail(KU_EINVAL, "ku_mgpu_create: a world that spans processes needs the unique id of rank 0");
  if (!all_here && n_local != 1) return mfail(KU_EUNSUP, "ku_mgpu_create: one rank per process, or all ranks in one process");
  ku_mgpu *m = new ku_mgpu();
  m->world = world;
  m->first_rank = first_rank;
  m->n_local = n_local;
  m->flags = flags;
  m->ranks.resize(n_local);
  m->sh.ptr_a.assign(n_local, nullptr);
  m->sh.ptr_b.assign(n_local, nullptr);
  m->sh.ptr_c.assign(n_local, nullptr);
  m->sh.vals.resize(n_local);
  m->sh.u64s.resize(n_local);
  m->sh.offs.assign(n_local, nullptr);
  m->sh.dev.assign(devices, devices + n_local);
  std::set<int> distinct(devices, devices + n_local);
  // KU_MGPU_FORCE_RCCL=1 takes the RCCL calls even for a world of one rank (a way to exercise them on a 1-GPU box)
  m->use_rccl = (world > 1 || getenv("KU_MGPU_FORCE_RCCL")) &&
                (!all_here || (distinct.size() == n_local && !(flags & KU_MGPU_NO_RCCL) && !getenv("KU_MGPU_NO_RCCL")));
  if (n_local > 1) {
    if (pthread_barrier_init(&m->sh.bar, nullptr, n_local) != 0) { delete m; return mfail(KU_ENOMEM, "pthread_barrier_init"); }
    m->sh.bar_init = true;
  }
  for (uint32_t i = 0; i < n_local; ++i) {
    auto &r = m->ranks[i];
    r.rank = first_rank + i;
    r.local = i;
    r.device = devices[i];
    int st = ku_ctx_create(devices[i], &r.ctx);
    if (st != KU_OK) { ku_mgpu_destroy(m); return st; }
  }
  if (m->use_rccl) {
    int st = rccl_ready();
    if (st != KU_OK) { ku_mgpu_destroy(m); return st; }
    if (all_here && !(id && n_local == 1)) {
      std::vector<ncclComm_t> comms(n_local);
      ncclResult_t e = g_rccl.CommInitAll(comms.data(), (int)n_local, devices);
      if (e != ncclSuccess) { ku_mgpu_destroy(m); return mfail(KU_EHIP, std::string("ncclCommInitAll: ") + g_rccl.GetErrorString(e)); }
      for (uint32_t i = 0; i < n_local; ++i) m->ranks[i].comm = comms[i];
    } else {
      ncclUniqueId u;
      memcpy(&u, id, sizeof u);
      if (hipSetDevice(devices[0]) != hipSuccess) { ku_mgpu_destroy(m); return mfail(KU_EHIP, "hipSetDevice failed"); }
      ncclResult_t e = g_rccl.CommInitRank(&m->ranks[0].comm, (int)world, u, (int)first_rank);
      if (e != ncclSuccess) { ku_mgpu_destroy(m); return mfail(KU_EHIP, std::string("ncclCommInitRank: ") + g_rccl.GetErrorString(e)); }
    }
  }
  *out = m;
  return KU_OK;
}

extern "C" ku_ctx *ku_mgpu_ctx(ku_mgpu *m, uint32_t local_index) {
  return (m && local_index < m->n_local) ? m->ranks[local_index].ctx : nullptr;
}
extern "C" int ku_mgpu_uses_rccl(const ku_mgpu *m) { return m && m->use_rccl ? 1 : 0; }
extern "C" int ku_mgpu_uses_routing(const ku_mgpu *m) { return m && m->route && !m->exact ? 1 : 0; }

extern "C" int ku_mgpu_set_timing(ku_mgpu *m, int on) {
  if (!m) return mfail(KU_EINVAL, "ku_mgpu_set_timing: null argument");
  m->timing = on != 0;
  return KU_OK;
}
extern "C" int ku_mgpu_step_times(ku_mgpu *m, uint32_t local_index, double *out) {
  if (!m || !out || local_index >= m->n_local) return mfail(KU_EINVAL, "ku_mgpu_step_times: bad argument");
  ku_mgpu::Rank &r = m->ranks[local_index];
  for (int i = 0; i < 6; ++i) out[i] = 0;
  if (hipSetDevice(r.device) != hipSuccess) return mfail(KU_EHIP, "hipSetDevice failed");
  for (auto &t : r.tev) {
    float ms = 0;
    if (hipEventSynchronize(t.second.second) != hipSuccess || hipEventElapsedTime(&ms, t.second.first, t.second.second) != hipSuccess)
      return mfail(KU_EHIP, "ku_mgpu_step_times: an event of the last step is not complete");
    if (t.first >= 0 && t.first < 3) out[t.first] += ms;
  }
  out[3] = r.t_rounds;
  out[4] = r.t_rec;
  out[5] = r.t_kmers;
  return KU_OK;
}

extern "C" int ku_mgpu_set_taxonomy(ku_mgpu *m, const ku_tax *tax) {
  if (!m || !tax) return mfail(KU_EINVAL, "ku_mgpu_set_taxonomy: null argument");
  M_TRY(run_all(m, [&](ku_mgpu::Rank &r) -> int {
    uint64_t n = 0;
    int st = ku_ctx_db_values(r.ctx, nullptr, &n);
    std::vector<uint32_t> mine(st == KU_OK ? n : 0), all;
    if (st == KU_OK && n) st = ku_ctx_db_values(r.ctx, mine.data(), &n);
    st = comm_allgather_values(m, r, st, mine, all);
    if (st != KU_OK) return st;
    st = ku_ctx_set_taxonomy(r.ctx, tax, all.data(), all.size());
    // owner routing needs every rank's minimizer range and the probe table everywhere
    uint64_t info[4] = {0, 0, 0, 0};
    int is_hash = 0, single = 0;
    if (st == KU_OK) st = ku_ctx_route_info(r.ctx, &info[0], &info[1], &is_hash, &single);
    info[2] = (uint64_t)(is_hash && single);
    std::vector<uint64_t> allinfo;
    st = comm_allgather_u64(m, r, st, info, 4, allinfo, ku_ctx_stream_of(r.ctx));
    if (st == KU_OK && r.local == 0) {
      m->own_lo.assign(m->world, 0);
      m->own_hi.assign(m->world, 0);
      bool ok = true;
      for (uint32_t q = 0; q < m->world; ++q) { m->own_lo[q] = allinfo[4 * q]; m->own_hi[q] = allinfo[4 * q + 1]; ok = ok && allinfo[4 * q + 2]; }
      // the scan finds a bin's owner by bisection: the ranges must ascend with the rank (a shard plan's do) and not overlap
      for (uint32_t q = 0; q + 1 < m->world; ++q) ok = ok && m->own_lo[q] <= m->own_hi[q] && m->own_hi[q] <= m->own_lo[q + 1];
      if (std::getenv("KU_ROUTE_DEBUG")) {
        fprintf(stderr, "[ku_route] rank %u: hash+single everywhere / ascending ranges: %d;", r.rank, (int)ok);
        for (uint32_t q = 0; q < m->world; ++q) fprintf(stderr, " [%llu, %llu) flag %llu", (unsigned long long)m->own_lo[q], (unsigned long long)m->own_hi[q], (unsigned long long)allinfo[4 * q + 2]);
        fprintf(stderr, "\n");
      }
      const char *ex = getenv("KU_MGPU_EXCHANGE");
      // (KU_MGPU_FORCE_ROUTE=1: also a world of one rank takes the routed path -- scan, records, owner kernel, gather on one
      // stream, nothing on the wire: the device work of a routed step in one clean kernel trace)
      m->route = ok && (m->world > 1 || std::getenv("KU_MGPU_FORCE_ROUTE")) && m->world <= 64 && !(m->flags & KU_MGPU_REPLICAS) && !(ex && (!strcmp(ex, "slots") || !strcmp(ex, "reduce")));
    }
    return st;
  }));
  m->tax_set = true;
  return KU_OK;
}

extern "C" int ku_mgpu_load_dbs(ku_mgpu *m, const ku_db *const *dbs, uint32_t n_dbs, const ku_tax *tax) {
  if (!m || !dbs || !n_dbs || !tax) return mfail(KU_EINVAL, "ku_mgpu_load_dbs: null argument");
  if (n_dbs == 1) return ku_mgpu_load(m, dbs[0], tax);
  // several databases are searched one after the other per k-mer, the first hit wins (classify.cpp:928-936): with the
  // first database cut into shards a later one could not tell whether another rank had found the k-mer already -- the
  // reference's own chunk mode only searches the first database (classify.cpp:639).  Replicas hold everything.
  if (!(m->flags & KU_MGPU_REPLICAS)) return mfail(KU_EUNSUP, "several databases on several GPUs need the replicas mode (KU_MGPU_REPLICAS)");
  ku_db_info info;
  M_TRY(ku_db_get_info(dbs[0], &info));
  M_TRY(run_all(m, [&](ku_mgpu::Rank &r) -> int {
    M_TRY(ku_ctx_load_db(r.ctx, dbs[0], 0, info.n_bins));
    for (uint32_t i = 1; i < n_dbs; ++i) M_TRY(ku_ctx_add_db(r.ctx, dbs[i]));
    return KU_OK;
  }));
  m->loaded = true;
  return ku_mgpu_set_taxonomy(m, tax);
}

extern "C" int ku_mgpu_load(ku_mgpu *m, const ku_db *db, const ku_tax *tax) {
  if (!m || !db || !tax) return mfail(KU_EINVAL, "ku_mgpu_load: null argument");
  ku_db_info info;
  M_TRY(ku_db_get_info(db, &info));
  std::vector<uint64_t> bounds(m->world + 1);
  M_TRY(ku_db_shard_plan(db, m->world, bounds.data()));
  const bool replicas = (m->flags & KU_MGPU_REPLICAS) != 0;
  M_TRY(run_all(m, [&](ku_mgpu::Rank &r) -> int {
    return replicas ? ku_ctx_load_db(r.ctx, db, 0, info.n_bins) : ku_ctx_load_db(r.ctx, db, bounds[r.rank], bounds[r.rank + 1]);
  }));
  m->loaded = true;
  return ku_mgpu_set_taxonomy(m, tax);
}

// ---------------------------------------------------------------------------- report modes over the group
extern "C" int ku_mgpu_enable_sparse(ku_mgpu *m, uint64_t work_unit_nt, uint32_t global_log2) {
  if (!m) return mfail(KU_EINVAL, "ku_mgpu_enable_sparse: null argument");
  if (!m->tax_set) return mfail(KU_ESTATE, "ku_mgpu_enable_sparse: load the database and the taxonomy first");
  if (!single_process(m)) return mfail(KU_EUNSUP, "the sparse-mode emulation over several GPUs needs the group in one process");
  if (work_unit_nt == 0) return mfail(KU_EUNSUP, "ku_mgpu_enable_sparse: a work unit size is needed (the ranks take whole units)");
  M_TRY(run_all(m, [&](ku_mgpu::Rank &r) -> int { return ku_ctx_enable_sparse(r.ctx, work_unit_nt, global_log2); }));
  m->sparse = true;
  m->unit_nt = work_unit_nt;
  m->acc_nt = 0;
  m->open_rank = -1;
  return KU_OK;
}
extern "C" int ku_mgpu_sparse_close_unit(ku_mgpu *m) {
  if (!m) return mfail(KU_EINVAL, "ku_mgpu_sparse_close_unit: null argument");
  if (!m->sparse) return KU_OK;
  if (m->open_rank >= 0) {
    ku_mgpu::Rank &r = m->ranks[m->open_rank];
    if (hipSetDevice(r.device) != hipSuccess) return mfail(KU_EHIP, "hipSetDevice failed");
    M_TRY(ku_sparse_close_unit(r.ctx));
  }
  m->acc_nt = 0;
  m->open_rank = -1;
  return KU_OK;
}
extern "C" int ku_mgpu_sparse_state(const ku_mgpu *m) {
  if (!m) return 0;
  int worst = m->sparse_gave_up ? 2 : (m->sparse ? 1 : 0);
  for (const auto &r : m->ranks) {
    const int s = ku_ctx_sparse_state(r.ctx);
    if (s == 2) worst = 2;
  }
  return worst;
}
extern "C" int ku_mgpu_enable_exact(ku_mgpu *m, uint32_t capacity_log2) {
  if (!m) return mfail(KU_EINVAL, "ku_mgpu_enable_exact: null argument");
  if (!m->tax_set) return mfail(KU_ESTATE, "ku_mgpu_enable_exact: load the database and the taxonomy first");
  // owner-computes: a k-mer is put into the set of the rank that owns its minimizer bin, so the ranks' sets are disjoint
  // and the distinct counts add up.  Replicas would see the same k-mer on several ranks.
  if (m->flags & KU_MGPU_REPLICAS) return mfail(KU_EUNSUP, "exact counting over several GPUs needs the sharded mode (the k-mer set is partitioned by owner)");
  M_TRY(run_all(m, [&](ku_mgpu::Rank &r) -> int { return ku_ctx_enable_exact(r.ctx, capacity_log2); }));
  m->exact = true;
  return KU_OK;
}

// ---------------------------------------------------------------------------- one batch
namespace {
// the sharded batch on one rank, everything on stream s: broadcast, lookup of the owned k-mers, slot merge, resolve
int rank_step_sharded(ku_mgpu *m, ku_mgpu::Rank &r, int st, void *d_seqs, uint64_t *d_off, uint32_t *d_len, uint32_t *d_calls,
                      uint32_t *d_taxa, uint32_t *d_hits, uint64_t n_bytes, uint64_t n_reads, const uint64_t *rb,
                      const uint64_t *pos, const ku_opts &opts, hipStream_t s, const uint64_t *h_off = nullptr,
                      const uint32_t *h_len = nullptr) {
  st = comm_broadcast(m, r, st, d_seqs, n_bytes, s);
  st = comm_broadcast(m, r, st, d_off, n_reads * 8, s);
  st = comm_broadcast(m, r, st, d_len, n_reads * 4, s);
  ku_opts lo = opts;
  lo.flags = (opts.flags & ~KU_F_MERGE_CHUNK) | KU_F_KEEP_SLOTS;
  if (st == KU_OK && m->exact) {
    // classifyExact: the k-mers this rank OWNS go into its set.  The lookup runs as a chunk pass over an array preset with
    // a mark (positions of other owners keep it), the set takes what is not marked, the marks become 0 for the exchange
    st = ku_exact_owned_step(r.ctx, d_seqs, d_off, d_len, n_reads, n_bytes, &lo, d_taxa, s);
  } else if (st == KU_OK) st = ku_lookup_device(r.ctx, d_seqs, n_bytes, &lo, d_taxa, s);
  st = comm_reduce_slices_max(m, r, st, d_taxa, pos, s);
  if (st != KU_OK) return st;
  const uint64_t r0 = rb[r.rank], nr = rb[r.rank + 1] - r0;
  ku_opts ro = opts;
  ro.flags &= ~(KU_F_KEEP_SLOTS | KU_F_MERGE_CHUNK);
  if (nr == 0) return KU_OK;
  if (m->sparse && h_len && !(opts.flags & KU_F_NO_COUNTS))  // the rank's slice is whole work units: its merged slots feed the emulation
    M_TRY(ku_ctx_sparse_pass_slots(r.ctx, d_seqs, d_off + r0, d_len + r0, h_off + r0, h_len + r0, nr, n_bytes, d_taxa,
                                   (opts.flags & KU_F_QUICK) ? std::max(1u, opts.min_hits) : 0u, s));
  return ku_resolve_device(r.ctx, d_seqs, d_off + r0, d_len + r0, nr, &ro, d_calls + r0, d_taxa, d_hits ? d_hits + r0 : nullptr, s);
}

// The owner-routed sharded batch on one rank: the rank scans only ITS slice of the reads, every run of
// k-mers that share a minimizer occurrence travels as one 16-byte record to the rank that owns the bin, one slot per k-mer
// comes back.  The slice is taken in ROUNDS of whole reads (about KU_ROUTE_ROUND positions each; every rank runs the same
// number of rounds: it owns bins in all of them), and a round is four stages:
//   S  scan -> records in W queues, tickets in the per-k-mer array; prefix sum of the records' k-mer counts, per-queue totals
//   G  all-gather of {records, k-mers} per queue: the ONE host round trip of a round (a queue that overflowed: rescan)
//   X  all-to-all of the records (16 B each); prefix sum over what arrived; owner kernel: probe + HLL + n_kmers
//   Y  all-to-all of the slots (4 B per k-mer); tickets -> slots fused with the resolve stage of the round's reads
// The scan and the resolve stage are bound by instruction issue and latency, the owner kernel by the random line rate of the
// memory system.  Consecutive rounds alternate between two buffer sets on two streams, issued
// S(0) G(0) { X(r)  S(r+1)  G(r+1)  Y(r) }: the owner kernel of round r runs beside the scan of round r + 1 and the resolve
// of round r - 1, and the copies of one round's exchange under the kernels of the other.  Measured (one rank forced
// through this path, 10 M reads): two streams 41.2 ms against 43.0 on one -- whole kernels side by side hide each other far
// less than the stages inside the fused single-GPU kernel do -- and every round costs ~0.5-1 ms (padding of the queues,
// launches, the host round trip), so rounds are LARGE (256 M positions: a 10 M-read step over eight ranks is one round);
// what the rounds are for is bounded buffers, and a record index that fits a ticket, whatever the size of a slice.
//   have_slice: the rank's slice of seqs / off / len is in place already (host batches: every rank uploads its own part);
//   else rank 0 holds the whole batch and the slices are scattered first (device batches).
struct RouteRound {
  uint64_t a = 0, sb = 0;    // first position (relative to the slice) and bytes to scan
  uint64_t ra = 0, rn = 0;   // the round's reads (relative to the slice's first read)
  uint64_t cap = 0;
  uint32_t chunk = 0;
  std::vector<uint64_t> all, send_at, send_n, ret_at, ret_n, recv_at, recv_n, rk_at, rk_n;
  uint64_t n_recv = 0, k_recv = 0, k_send = 0;
};

int rank_step_routed(ku_mgpu *m, ku_mgpu::Rank &r, int st, void *d_seqs, uint64_t *d_off, uint32_t *d_len, uint32_t *d_calls,
                     uint32_t *d_taxa, uint32_t *d_hits, uint64_t n_bytes, uint64_t n_reads, const uint64_t *rb, const uint64_t *pos,
                     const ku_opts &opts, hipStream_t s, const uint64_t *h_off, const uint32_t *h_len, bool have_slice) {
  const uint32_t W = m->world;
  const uint64_t r0 = rb[r.rank], nr = rb[r.rank + 1] - r0, p0 = pos[r.rank], nb = pos[r.rank + 1] - p0;
  (void)n_reads;
  if (!have_slice) {
    st = comm_scatter_slices(m, r, st, d_seqs, pos, 1, s);
    st = comm_scatter_slices(m, r, st, d_off, rb, 8, s);
    st = comm_scatter_slices(m, r, st, d_len, rb, 4, s);
  }
  // ---- rounds: every rank takes part in every round, so their number follows from the largest slice
  uint64_t round_pos = 256ull << 20;
  if (const char *e = std::getenv("KU_ROUTE_ROUND")) round_pos = std::max<uint64_t>(256, std::strtoull(e, nullptr, 10));
  uint64_t nb_max = 0;
  for (uint32_t q = 0; q < W; ++q) nb_max = std::max(nb_max, pos[q + 1] - pos[q]);
  const uint32_t R = (uint32_t)std::min<uint64_t>(4096, std::max<uint64_t>(1, (nb_max + round_pos - 1) / round_pos));
  const bool counts = !(opts.flags & (KU_F_NO_COUNTS | KU_F_QUICK));  // quick mode books the scanned prefix in the resolve stage
  const bool emulate = m->sparse && h_len && !(opts.flags & KU_F_NO_COUNTS);
  ku_opts ro = opts;
  ro.flags &= ~(KU_F_KEEP_SLOTS | KU_F_MERGE_CHUNK);
  // the resolve stage inside the rounds (the fused kernel's ROUTE instance: slot through the ticket, hit counts,
  // resolve_tree, call, taxids in place); quick mode, reads beyond 65535 k-mers and the sparse-sketch emulation (it wants the
  // slots in the array) turn the tickets into slots per round and resolve the slice behind the last one
  bool fused = !emulate && st == KU_OK && ku_ctx_route_resolve_prepare(r.ctx, &ro, s) == KU_OK;
  // a round ends on a read boundary: the first position of every round's first read
  std::vector<RouteRound> rd(R);
  {
    std::vector<uint64_t> cut(R + 1, nb);
    cut[0] = 0;
    std::vector<uint64_t> rc(R + 1);
    for (uint32_t i = 0; i <= R; ++i) rc[i] = nr * i / R;
    if (R > 1 && nr) {
      if (h_off) {
        for (uint32_t i = 1; i < R; ++i) cut[i] = rc[i] < nr ? h_off[r0 + rc[i]] - p0 : nb;
      } else {
        std::vector<uint64_t> tmp(R, 0);
        for (uint32_t i = 1; i < R && st == KU_OK; ++i)
          if (rc[i] < nr && hipMemcpyAsync(&tmp[i], d_off + r0 + rc[i], 8, hipMemcpyDeviceToHost, s) != hipSuccess) st = mfail(KU_EHIP, "read offsets copy failed");
        if (st == KU_OK && hipStreamSynchronize(s) != hipSuccess) st = mfail(KU_EHIP, "read offsets copy failed");
        for (uint32_t i = 1; i < R; ++i) cut[i] = rc[i] < nr ? tmp[i] - p0 : nb;
      }
      for (uint32_t i = 1; i < R; ++i)
        if (st == KU_OK && (cut[i] < cut[i - 1] || cut[i] > nb)) st = mfail(KU_EINVAL, "owner routing: the reads of a slice must be in buffer order");
      // Equal numbers of READS per round are equal numbers of positions only when the reads are about equally long.  With
      // contigs among short reads a round could hold far more than round_pos positions -- its queues more records than a
      // ticket can number.  Then the rounds are cut by POSITION: round i starts at the first read at or behind
      // i / R of the slice (on the host when the offsets are there, else by a bisection over the device array: R - 1 values
      // per step, one synchronisation per step).
      uint64_t widest = 0;
      for (uint32_t i = 0; i < R && st == KU_OK; ++i) widest = std::max(widest, cut[i + 1] - cut[i]);
      if (st == KU_OK && widest > round_pos + round_pos / 2) {
        std::vector<uint64_t> lo(R, 0), hi(R, nr);  // first read with offset >= target_i lies in [lo, hi]
        std::vector<uint64_t> target(R, 0);
        for (uint32_t i = 1; i < R; ++i) target[i] = p0 + nb / R * i;
        if (h_off) {
          for (uint32_t i = 1; i < R; ++i) lo[i] = (uint64_t)(std::lower_bound(h_off + r0, h_off + r0 + nr, target[i]) - (h_off + r0));
        } else {
          std::vector<uint64_t> val(R, 0);
          for (;;) {
            bool any = false;
            for (uint32_t i = 1; i < R && st == KU_OK; ++i)
              if (lo[i] < hi[i]) {
                any = true;
                if (hipMemcpyAsync(&val[i], d_off + r0 + (lo[i] + hi[i]) / 2, 8, hipMemcpyDeviceToHost, s) != hipSuccess) st = mfail(KU_EHIP, "read offsets copy failed");
              }
            if (!any || st != KU_OK) break;
            if (hipStreamSynchronize(s) != hipSuccess) { st = mfail(KU_EHIP, "read offsets copy failed"); break; }
            for (uint32_t i = 1; i < R; ++i)
              if (lo[i] < hi[i]) {
                const uint64_t mid = (lo[i] + hi[i]) / 2;
                if (val[i] < target[i]) lo[i] = mid + 1; else hi[i] = mid;
              }
          }
          if (st == KU_OK) {  // the offsets of the reads found
            for (uint32_t i = 1; i < R && st == KU_OK; ++i)
              if (lo[i] < nr && hipMemcpyAsync(&val[i], d_off + r0 + lo[i], 8, hipMemcpyDeviceToHost, s) != hipSuccess) st = mfail(KU_EHIP, "read offsets copy failed");
            if (st == KU_OK && hipStreamSynchronize(s) != hipSuccess) st = mfail(KU_EHIP, "read offsets copy failed");
            for (uint32_t i = 1; i < R; ++i) cut[i] = lo[i] < nr ? val[i] - p0 : nb;
          }
        }
        for (uint32_t i = 1; i < R && st == KU_OK; ++i) {
          lo[i] = std::max(lo[i], lo[i - 1]);
          rc[i] = lo[i];
          if (h_off) cut[i] = rc[i] < nr ? h_off[r0 + rc[i]] - p0 : nb;
        }
        for (uint32_t i = 1; i < R; ++i)
          if (st == KU_OK && (cut[i] < cut[i - 1] || cut[i] > nb)) st = mfail(KU_EINVAL, "owner routing: the reads of a slice must be in buffer order");
      }
    }
    for (uint32_t i = 0; i < R; ++i) {
      rd[i].a = cut[i];
      rd[i].sb = st == KU_OK ? cut[i + 1] - cut[i] : 0;
      rd[i].ra = rc[i];
      rd[i].rn = rc[i + 1] - rc[i];
    }
  }
  // ---- the second set's stream follows the caller's up to here
  hipStream_t str[2] = {s, s};
  // (with the stage timing on, everything stays on one stream: kernels side by side would stretch each other's event pairs.
  // Over RCCL too, unless KU_ROUTE_TWO_STREAMS=1: collectives of one communicator issued from two streams are ordered by the
  // library, but that path has never run on more than one physical GPU here, and what the second stream buys is ~4 %.)
  if (R > 1 && !std::getenv("KU_ROUTE_ONE_STREAM") && !m->timing && (!m->use_rccl || std::getenv("KU_ROUTE_TWO_STREAMS"))) {
    if (!r.aux && hipStreamCreateWithFlags(&r.aux, hipStreamNonBlocking) != hipSuccess) { r.aux = nullptr; st = st == KU_OK ? mfail(KU_EHIP, "stream creation failed") : st; }
    for (hipEvent_t *e : {&r.ev_a, &r.ev_b, &r.ev_r})
      if (!*e && hipEventCreateWithFlags(e, hipEventDisableTiming) != hipSuccess) { *e = nullptr; st = st == KU_OK ? mfail(KU_EHIP, "event creation failed") : st; }
    if (st == KU_OK) {
      if (hipEventRecord(r.ev_a, s) != hipSuccess || hipStreamWaitEvent(r.aux, r.ev_a, 0) != hipSuccess) st = mfail(KU_EHIP, "stream fork failed");
      str[1] = r.aux;
    }
  }
  const uint32_t n_info = 2 * W + 1;
  struct Tables { uint64_t *lo, *hi; unsigned long long *cursor, *info; } tb[2] = {};
  for (int i = 0; i < 2; ++i) {
    auto &t = r.rs[i];
    if (st == KU_OK) st = t.rt_dev.reserve((2ull * W + (uint64_t)W * KU_ROUTE_CURSOR_STRIDE + n_info + 8) * 8);
    if (st != KU_OK) break;
    tb[i].lo = (uint64_t *)t.rt_dev.p;
    tb[i].hi = tb[i].lo + W;
    tb[i].cursor = (unsigned long long *)(tb[i].hi + W);
    tb[i].info = tb[i].cursor + (size_t)W * KU_ROUTE_CURSOR_STRIDE;
    if (hipMemcpyAsync(tb[i].lo, m->own_lo.data(), 8ull * W, hipMemcpyHostToDevice, str[i]) != hipSuccess ||
        hipMemcpyAsync(tb[i].hi, m->own_hi.data(), 8ull * W, hipMemcpyHostToDevice, str[i]) != hipSuccess)
      st = mfail(KU_EHIP, "routing tables upload failed");
  }
  // ku_mgpu_set_timing: stage `tag` between two events on stream q
  if (m->timing) { r.tev.clear(); r.tev_used = 0; r.t_rounds = R; r.t_rec = r.t_kmers = 0; }
  auto t_begin = [&](int tag, hipStream_t q) {
    if (!m->timing || st != KU_OK) return;
    while (r.tev_pool.size() < r.tev_used + 2) {
      hipEvent_t e = nullptr;
      if (hipEventCreate(&e) != hipSuccess) return;
      r.tev_pool.push_back(e);
    }
    hipEvent_t e0 = r.tev_pool[r.tev_used], e1 = r.tev_pool[r.tev_used + 1];
    r.tev_used += 2;
    (void)hipEventRecord(e0, q);
    r.tev.push_back({tag, {e0, e1}});
  };
  auto t_end = [&](hipStream_t q) {
    if (!m->timing || r.tev.empty()) return;
    (void)hipEventRecord(r.tev.back().second.second, q);
  };
  bool resolved_before = false;  // a resolve kernel was queued: the next one (other stream) waits for it (they share the context's spill workspace)

  // S: scan + prefix of round i on its set's stream (first attempt: the default room per queue)
  auto scan_round = [&](uint32_t i) {
    RouteRound &x = rd[i];
    auto &t = r.rs[i & 1];
    hipStream_t q = str[i & 1];
    const uint64_t n_q = (uint64_t)W * x.cap;
    if (st == KU_OK && n_q > KU_ROUTE_MAX_RECORDS) st = mfail(KU_EUNSUP, "owner routing: more records per rank and round than a ticket can number");
    if (st == KU_OK && (t.q_rec.reserve(n_q * 16) || t.q_kb.reserve((n_q + 1) * 4) || t.pfx_work.reserve(ku_route_prefix_work_bytes(n_q))))
      st = mfail(KU_ENOMEM, "device memory for the routing queues");
    if (st == KU_OK && hipMemsetAsync(tb[i & 1].cursor, 0, 8ull * W * KU_ROUTE_CURSOR_STRIDE, q) != hipSuccess) st = mfail(KU_EHIP, "routing cursors reset failed");
    if (st != KU_OK) return;
    t_begin(0, q);
    KuRouteDev rt{};
    rt.own_lo = tb[i & 1].lo; rt.own_hi = tb[i & 1].hi; rt.cursor = tb[i & 1].cursor;
    rt.q_rec = (uint4 *)t.q_rec.p;
    rt.cap = x.cap; rt.world = W; rt.chunk = x.chunk;
    if (x.sb) st = ku_ctx_route_scan(r.ctx, (const char *)d_seqs + p0 + x.a, x.sb, d_taxa + p0 + x.a, rt, q);
    if (st == KU_OK) {
      st = ku_launch_route_prefix(t.q_rec.p, n_q, x.cap, tb[i & 1].cursor, (uint32_t *)t.q_kb.p, tb[i & 1].info, t.pfx_work.p, q);
      if (st != KU_OK) st = mfail(st, "routing prefix kernels failed");
    }
    t_end(q);
  };
  auto stage_S = [&](uint32_t i) {
    RouteRound &x = rd[i];
    // One scanning pass: every owner's queue gets room for about 2.5 times its fair share of the records a slice of random
    // sequence makes (a record per ~10 positions; the shard bounds are quantiles of the database's k-mers, reads follow
    // the database -- but the owners of the high bins get up to twice the records per k-mer: a large minimizer does not
    // stay one for long), plus the chunk every block of the scan may leave partly used; the cursors come back as the
    // true totals.  A queue that did not hold its total (low-complexity reads, a batch from one corner of the minimizer
    // space) sends the rank through a second pass with queues of the size the first one counted -- the results do not
    // depend on which it was (KU_ROUTE_CAP: records per queue, for the tests).
    const unsigned grid = x.sb ? ku_route_scan_grid(x.sb, ku_ctx_cus_of(r.ctx)) : 1u;
    // (a block claims `chunk` records of a queue at a time: with few owners larger claims keep the adds on one cursor apart)
    x.chunk = KU_ROUTE_CHUNK * std::max(1u, 8u / W);
    x.cap = x.sb / (4ull * W) + ((uint64_t)grid + 1) * x.chunk;
    if (const char *e = std::getenv("KU_ROUTE_CAP")) x.cap = std::max<uint64_t>(1, std::strtoull(e, nullptr, 10));
    x.cap = (x.cap + x.chunk - 1) / x.chunk * x.chunk;
    scan_round(i);
  };
  // G: every rank learns how much it gets from whom (records and k-mers), and whether anybody's queues overflowed (then
  // the ranks concerned scan again and everybody gathers again: all see the same numbers, so all agree)
  auto stage_G = [&](uint32_t i) -> int {
    RouteRound &x = rd[i];
    auto &t = r.rs[i & 1];
    hipStream_t q = str[i & 1];
    for (int attempt = 0;; ++attempt) {
      st = comm_allgather_u64_dev(m, r, st, tb[i & 1].info, n_info, x.all, q);
      if (st != KU_OK) return st;
      if (std::getenv("KU_ROUTE_DEBUG") && r.rank == 0) {
        fprintf(stderr, "[ku_route] round %u/%u attempt %d: sb %llu reads %llu", i, R, attempt, (unsigned long long)x.sb, (unsigned long long)x.rn);
        for (uint32_t qq = 0; qq < W; ++qq) {
          fprintf(stderr, " | rank %u cap %llu:", qq, (unsigned long long)x.all[(size_t)qq * n_info + 2 * W]);
          for (uint32_t o = 0; o < W; ++o)
            fprintf(stderr, " %llu/%llu", (unsigned long long)x.all[(size_t)qq * n_info + 2 * o], (unsigned long long)x.all[(size_t)qq * n_info + 2 * o + 1]);
        }
        fprintf(stderr, "\n");
      }
      bool over_any = false, over_mine = false;
      uint64_t mine_max = 0;
      for (uint32_t qq = 0; qq < W; ++qq)
        for (uint32_t o = 0; o < W; ++o) {
          const bool ov = x.all[(size_t)qq * n_info + 2 * o] > x.all[(size_t)qq * n_info + 2 * W];
          over_any |= ov;
          if (qq == r.rank) { over_mine |= ov; mine_max = std::max(mine_max, x.all[(size_t)qq * n_info + 2 * o]); }
        }
      if (!over_any) break;
      if (attempt >= 1) return mfail(KU_ESTATE, "owner routing: a queue overflowed twice");
      if (over_mine) {
        x.cap = (mine_max + x.chunk - 1) / x.chunk * x.chunk;
        scan_round(i);
      }
    }
    // segment tables: what goes where (records / k-mers), what comes from whom
    const uint64_t *mine = x.all.data() + (size_t)r.rank * n_info;
    for (auto *v : {&x.send_at, &x.send_n, &x.ret_at, &x.ret_n, &x.recv_at, &x.recv_n, &x.rk_at, &x.rk_n}) v->assign(W, 0);
    x.n_recv = x.k_recv = x.k_send = 0;
    for (uint32_t qq = 0; qq < W; ++qq) {
      x.send_at[qq] = (uint64_t)qq * x.cap;
      x.send_n[qq] = mine[2 * qq];
      x.ret_at[qq] = x.k_send;
      x.ret_n[qq] = mine[2 * qq + 1];
      x.k_send += x.ret_n[qq];
      x.recv_at[qq] = x.n_recv;
      x.recv_n[qq] = x.all[(size_t)qq * n_info + 2 * r.rank];
      x.n_recv += x.recv_n[qq];
      x.rk_at[qq] = x.k_recv;
      x.rk_n[qq] = x.all[(size_t)qq * n_info + 2 * r.rank + 1];
      x.k_recv += x.rk_n[qq];
    }
    if (x.n_recv >= (1ull << 32) || x.k_recv >= (1ull << 32) || x.k_send >= (1ull << 32)) st = mfail(KU_EUNSUP, "owner routing: more than 2^32 k-mers per rank and round");
    if (st == KU_OK && (t.r_rec.reserve(std::max<uint64_t>(x.n_recv, 1) * 16) || t.r_kb.reserve((x.n_recv + 1) * 4) ||
                        t.pfx_work.reserve(ku_route_prefix_work_bytes(std::max(x.n_recv, (uint64_t)W * x.cap))) ||
                        t.r_slots.reserve(std::max<uint64_t>(x.k_recv, 1) * 4) || t.ret_slots.reserve(std::max<uint64_t>(x.k_send, 1) * 4)))
      st = mfail(KU_ENOMEM, "device memory for the routing queues");
    return st;
  };
  // X: records to their owners (16 B per run of k-mers), probe + accounting there
  auto stage_X = [&](uint32_t i) {
    RouteRound &x = rd[i];
    auto &t = r.rs[i & 1];
    hipStream_t q = str[i & 1];
    st = comm_alltoallv(m, r, st, t.q_rec.p, x.send_at.data(), x.send_n.data(), t.r_rec.p, x.recv_at.data(), x.recv_n.data(), 16, q);
    t_begin(1, q);
    if (m->timing) { r.t_rec += (double)x.n_recv; r.t_kmers += (double)x.k_recv; }
    if (st == KU_OK) {
      st = ku_launch_route_prefix(t.r_rec.p, x.n_recv, 0, nullptr, (uint32_t *)t.r_kb.p, nullptr, t.pfx_work.p, q);
      if (st != KU_OK) st = mfail(st, "routing prefix kernels failed");
    }
    if (st == KU_OK) st = ku_ctx_route_owner(r.ctx, t.r_rec.p, x.n_recv, (const uint32_t *)t.r_kb.p, (uint32_t *)t.r_slots.p, counts, q);
    t_end(q);
  };
  // Y: slots back (4 B per k-mer), into place: resolve stage of the round's reads, or tickets -> slots
  auto stage_Y = [&](uint32_t i) {
    RouteRound &x = rd[i];
    auto &t = r.rs[i & 1];
    hipStream_t q = str[i & 1];
    st = comm_alltoallv(m, r, st, t.r_slots.p, x.rk_at.data(), x.rk_n.data(), t.ret_slots.p, x.ret_at.data(), x.ret_n.data(), 4, q);
    if (st != KU_OK) return;
    t_begin(2, q);
    if (fused && x.rn) {
      if (resolved_before && str[0] != str[1] && hipStreamWaitEvent(q, r.ev_r, 0) != hipSuccess) { st = mfail(KU_EHIP, "stream wait failed"); return; }
      const uint64_t f = r0 + x.ra;
      st = ku_ctx_route_resolve(r.ctx, d_off + f, d_len + f, x.rn, &ro, d_calls + f, d_taxa, d_hits ? d_hits + f : nullptr,
                                (const uint32_t *)t.q_kb.p, (const uint32_t *)t.ret_slots.p, q);
      if (st == KU_OK && str[0] != str[1] && hipEventRecord(r.ev_r, q) != hipSuccess) st = mfail(KU_EHIP, "event record failed");
      resolved_before = true;
    } else if (!fused && x.sb) {
      st = ku_launch_route_gather(d_taxa + p0 + x.a, x.sb, (const uint32_t *)t.q_kb.p, (const uint32_t *)t.ret_slots.p, q);
      if (st != KU_OK) st = mfail(st, "routing gather kernel failed");
    }
    t_end(q);
  };

  stage_S(0);
  M_TRY(stage_G(0));
  for (uint32_t i = 0; i < R; ++i) {
    stage_X(i);                            // ... ends with the owner kernel queued on this round's stream
    if (i + 1 < R) stage_S(i + 1);         // the next round's scan beside it, on the other stream
    if (i + 1 < R) M_TRY(stage_G(i + 1));  // (host: waits for that scan)
    stage_Y(i);                            // (host: waits for the owner kernel) slots back, resolve stage queued
  }
  // the caller's stream takes over again
  if (str[1] != s && r.ev_b) {
    // (KU_TEST_DROP_STREAM_JOIN, test builds only (-DKU_TEST_HOOKS): the join is left out, i.e. the bug an asynchronous collective library exposes
    // and a synchronous stand-in hides; tests/test_gpu_rccl_shim.py checks that the stand-in of round 5 makes the step fail)
#ifdef KU_TEST_HOOKS
    static const bool drop_join = std::getenv("KU_TEST_DROP_STREAM_JOIN") != nullptr;
#else
    constexpr bool drop_join = false;
#endif
    if (hipEventRecord(r.ev_b, str[1]) != hipSuccess || (!drop_join && hipStreamWaitEvent(s, r.ev_b, 0) != hipSuccess))
      st = st == KU_OK ? mfail(KU_EHIP, "stream join failed") : st;
  }
  if (st != KU_OK) return st;
  if (nr == 0 || fused) return KU_OK;
  if (emulate)
    M_TRY(ku_ctx_sparse_pass_slots(r.ctx, d_seqs, d_off + r0, d_len + r0, h_off + r0, h_len + r0, nr, n_bytes, d_taxa,
                                   (opts.flags & KU_F_QUICK) ? std::max(1u, opts.min_hits) : 0u, s));
  t_begin(2, s);
  st = ku_resolve_device(r.ctx, d_seqs, d_off + r0, d_len + r0, nr, &ro, d_calls + r0, d_taxa, d_hits ? d_hits + r0 : nullptr, s);
  t_end(s);
  return st;
}
}  // namespace

extern "C" int ku_mgpu_step_device(ku_mgpu *m, const ku_mgpu_dev_batch *local, uint64_t n_bytes, uint64_t n_reads,
                                   const uint64_t *read_bounds, const uint64_t *pos_bounds, const ku_opts *opts) {
  if (!m || !local || !read_bounds || !pos_bounds) return mfail(KU_EINVAL, "ku_mgpu_step_device: null argument");
  if (!m->tax_set) return mfail(KU_ESTATE, "ku_mgpu_step_device: load the database and the taxonomy first");
  if (m->flags & KU_MGPU_REPLICAS) return mfail(KU_EINVAL, "ku_mgpu_step_device is the sharded step; replicas classify through their own contexts");
  const ku_opts o = opts ? *opts : ku_opts{0, 1, 0, 0};
  m->reduced = false;
  return run_all(m, [&](ku_mgpu::Rank &r) -> int {
    const ku_mgpu_dev_batch &b = local[r.local];
    if (n_bytes && (!b.d_seqs || !b.d_taxa)) return mfail(KU_EINVAL, "ku_mgpu_step_device: null buffer");
    hipStream_t s = b.stream ? (hipStream_t)b.stream : ku_ctx_stream_of(r.ctx);
    if (m->route && !m->exact)
      return rank_step_routed(m, r, KU_OK, b.d_seqs, b.d_seq_off, b.d_seq_len, b.d_calls, b.d_taxa, nullptr, n_bytes, n_reads, read_bounds,
                              pos_bounds, o, s, nullptr, nullptr, false);
    return rank_step_sharded(m, r, KU_OK, b.d_seqs, b.d_seq_off, b.d_seq_len, b.d_calls, b.d_taxa, nullptr, n_bytes, n_reads,
                             read_bounds, pos_bounds, o, s);
  });
}

extern "C" int ku_mgpu_classify_batch_rle(ku_mgpu *m, const char *seqs, uint64_t n_bytes, const uint64_t *seq_off,
                                          const uint32_t *seq_len, uint64_t n_reads, const ku_opts *opts, uint32_t *calls,
                                          uint32_t *hits, uint64_t *run_off, uint32_t *run_cnt, uint64_t *n_runs) {
  if (!m || !n_runs || (n_bytes && !seqs) || (n_reads && (!seq_off || !seq_len || !calls || !run_off || !run_cnt)))
    return mfail(KU_EINVAL, "ku_mgpu_classify_batch_rle: null argument");
  if (!single_process(m)) return mfail(KU_EUNSUP, "host batches go through a single-process group");
  if (!m->tax_set) return mfail(KU_ESTATE, "ku_mgpu_classify_batch_rle: load the database and the taxonomy first");
  *n_runs = 0;
  for (auto &r : m->ranks) r.n_runs = r.run_base = 0;
  if (n_reads == 0) return KU_OK;
  m->reduced = false;
  ku_opts o = opts ? *opts : ku_opts{0, 1, 0, 0};
  o.flags &= ~(KU_F_KEEP_SLOTS | KU_F_MERGE_CHUNK);
  uint32_t max_len = 0;
  for (uint64_t i = 0; i < n_reads; ++i) {
    if (seq_off[i] + seq_len[i] > n_bytes) return mfail(KU_EINVAL, "read " + std::to_string(i) + " exceeds the sequence buffer");
    if (i && seq_off[i] < seq_off[i - 1]) return mfail(KU_EINVAL, "ku_mgpu_classify_batch_rle: reads must be in buffer order");
    max_len = std::max(max_len, seq_len[i]);
  }
  o.max_read_len = max_len;
  const bool replicas = (m->flags & KU_MGPU_REPLICAS) != 0, quick = (o.flags & KU_F_QUICK) != 0;
  // slices of the read dimension, balanced by bytes and cut at read boundaries -- with the sparse-mode emulation on, at
  // WORK UNIT boundaries (a unit closes behind the read that fills it, classify.cpp:510-521): every rank then runs the
  // emulation on whole units; the unit that is still open when the batch ends continues on rank 0 with the next batch
  const uint32_t W = m->world;
  std::vector<uint64_t> rb(W + 1), pos(W + 1);
  if (m->sparse && ku_mgpu_sparse_state(m) == 2) {
    // a rank ran out of memory for the emulation's tables in an earlier batch and switched it off there: no rank's sets
    // are the run's any more.  The whole group goes on with the dense registers (no unit bookkeeping, no open unit to move
    // -- ku_ctx_sparse_move_open_unit would refuse a context without tables and take the run down); ku_mgpu_sparse_state
    // keeps saying 2, the report carries its estimates and says so.
    for (auto &r : m->ranks) {
      if (hipSetDevice(r.device) != hipSuccess) return mfail(KU_EHIP, "hipSetDevice failed");
      if (ku_ctx_sparse_on(r.ctx)) M_TRY(ku_ctx_disable_sparse(r.ctx));
    }
    m->sparse = false;
    m->sparse_gave_up = true;
    m->open_rank = -1;
    m->acc_nt = 0;
  }
  const bool sparse = m->sparse && !(o.flags & KU_F_NO_COUNTS);
  rb[0] = 0;
  if (sparse) {
    if (m->open_rank > 0) {  // the open unit's state moves to the rank that takes the first reads
      M_TRY(ku_ctx_sparse_move_open_unit(m->ranks[m->open_rank].ctx, m->ranks[0].ctx));
      m->open_rank = 0;
    }
    std::vector<uint64_t> unit_end;  // read indices behind which a unit closes
    uint64_t acc = m->acc_nt;
    for (uint64_t i = 0; i < n_reads; ++i) {
      acc += seq_len[i];
      if (acc >= m->unit_nt) { unit_end.push_back(i + 1); acc = 0; }
    }
    for (uint32_t q = 1; q < W; ++q) {
      const uint64_t target = n_bytes / W * q;
      // the first unit boundary at or behind the target byte (none: the rest of the batch is one rank's)
      auto it = std::lower_bound(unit_end.begin(), unit_end.end(), target, [&](uint64_t e, uint64_t t) { return (e < n_reads ? seq_off[e] : n_bytes) < t; });
      rb[q] = std::max<uint64_t>(rb[q - 1], it == unit_end.end() ? n_reads : *it);
    }
    m->acc_nt = acc;
  } else {
    for (uint32_t q = 1; q < W; ++q) {
      const uint64_t target = n_bytes / W * q;
      rb[q] = std::max<uint64_t>(rb[q - 1], (uint64_t)(std::lower_bound(seq_off, seq_off + n_reads, target) - seq_off));
    }
  }
  rb[W] = n_reads;
  for (uint32_t q = 0; q < W; ++q) pos[q] = rb[q] < n_reads ? seq_off[rb[q]] : n_bytes;
  pos[0] = 0;
  pos[W] = n_bytes;
  if (sparse) {  // where the open unit (if any) ends up: the last rank that got reads
    m->open_rank = -1;
    if (m->acc_nt > 0)
      for (uint32_t q = 0; q < W; ++q)
        if (rb[q + 1] > rb[q]) m->open_rank = (int)q;
  }
  std::vector<uint64_t> totals(W, 0);
  M_TRY(run_all(m, [&](ku_mgpu::Rank &r) -> int {
    hipStream_t s = ku_ctx_stream_of(r.ctx);
    const uint64_t r0 = rb[r.rank], nr = rb[r.rank + 1] - r0;
    // what this rank keeps resident: the whole batch (sharded: every rank scans everything) or its slice (replicas)
    const uint64_t b0 = replicas ? pos[r.rank] : 0, nb = replicas ? pos[r.rank + 1] - b0 : n_bytes;
    const uint64_t nq = replicas ? nr : n_reads;
    const uint64_t runs_cap = (replicas ? nb : pos[r.rank + 1] - pos[r.rank]) + 1;
    int st = KU_OK;
    if (!replicas && (r.seqs.reserve(nb + 16) || r.off.reserve(nq * 8 + 8) || r.len.reserve(nq * 4 + 4) || r.calls.reserve(nq * 4 + 4) ||
                      r.taxa.reserve((nb + 16) * 4) || r.hits.reserve(nq * 4 + 4) || r.runs.reserve(runs_cap * 8) ||
                      r.roff.reserve(nq * 8 + 8) || r.rcnt.reserve(nq * 4 + 4) || r.small.reserve(64)))
      st = mfail(KU_ENOMEM, "device memory for the batch");
    uint32_t *d_calls = (uint32_t *)r.calls.p, *d_hits = (uint32_t *)r.hits.p, *d_taxa = (uint32_t *)r.taxa.p;
    uint64_t *d_off = (uint64_t *)r.off.p;
    uint32_t *d_len = (uint32_t *)r.len.p;
    if (replicas) {
      // a replica is a whole single-GPU pipeline on its slice: the context's own host-buffer entry point (fused kernel with
      // run-length encoded output, the slice uploaded in segments under the kernels, the emulation's fast path)
      if (nr == 0) return KU_OK;
      std::vector<uint64_t> rel(nr);  // slice offsets are relative to the slice's first byte
      for (uint64_t i = 0; i < nr; ++i) rel[i] = seq_off[r0 + i] - b0;
      uint64_t nruns = 0;
      M_TRY(ku_classify_batch_rle(r.ctx, seqs + b0, nb, rel.data(), seq_len + r0, nr, &o, calls + r0, hits ? hits + r0 : nullptr, run_off + r0,
                                  run_cnt + r0, &nruns));
      totals[r.rank] = nruns;
      return KU_OK;
    } else if (m->route && !m->exact) {
      // owner routing: every rank takes its own slice of the host batch over its own PCIe link (no broadcast at all)
      const uint64_t p0 = pos[r.rank], pb = pos[r.rank + 1] - p0;
      if (st == KU_OK && ((pb && hipMemcpyAsync((char *)r.seqs.p + p0, seqs + p0, pb, hipMemcpyHostToDevice, s) != hipSuccess) ||
                          (nr && (hipMemcpyAsync(d_off + r0, seq_off + r0, nr * 8, hipMemcpyHostToDevice, s) != hipSuccess ||
                                  hipMemcpyAsync(d_len + r0, seq_len + r0, nr * 4, hipMemcpyHostToDevice, s) != hipSuccess))))
        st = mfail(KU_EHIP, "upload of the batch failed");
      st = rank_step_routed(m, r, st, r.seqs.p, d_off, d_len, d_calls, d_taxa, d_hits, n_bytes, n_reads, rb.data(), pos.data(), o, s,
                            seq_off, seq_len, true);
    } else {
      if (st == KU_OK && r.rank == 0 &&
          (hipMemcpyAsync(r.seqs.p, seqs, n_bytes, hipMemcpyHostToDevice, s) != hipSuccess ||
           hipMemcpyAsync(d_off, seq_off, n_reads * 8, hipMemcpyHostToDevice, s) != hipSuccess ||
           hipMemcpyAsync(d_len, seq_len, n_reads * 4, hipMemcpyHostToDevice, s) != hipSuccess))
        st = mfail(KU_EHIP, "upload of the batch failed");
      st = rank_step_sharded(m, r, st, r.seqs.p, d_off, d_len, d_calls, d_taxa, d_hits, n_bytes, n_reads, rb.data(),
                             pos.data(), o, s, seq_off, seq_len);
    }
    if (st != KU_OK) return st;
    if (nr == 0) return KU_OK;
    // run-length encoding of this rank's reads; (run_off, run_cnt) index the rank's own run array
    const uint64_t lq = replicas ? 0 : r0;  // index of the slice's first read in this rank's arrays
    unsigned long long *d_counter = (unsigned long long *)r.small.p;
    if (quick) {
      M_HIP(hipMemsetAsync(d_counter, 0, 8, s));
      M_HIP(hipMemsetAsync((uint64_t *)r.roff.p + lq, 0, nr * 8, s));
      M_HIP(hipMemsetAsync((uint32_t *)r.rcnt.p + lq, 0, nr * 4, s));
    } else {
      M_TRY(ku_launch_rle(d_taxa, ku_ctx_k_of(r.ctx), d_off + lq, d_len + lq, nr, runs_cap /* ~ bases */, r.runs.p, runs_cap, d_counter,
                          (uint64_t *)r.roff.p + lq, (uint32_t *)r.rcnt.p + lq, ku_ctx_cus_of(r.ctx), s));
    }
    unsigned long long total = 0;
    M_HIP(hipMemcpyAsync(&total, d_counter, 8, hipMemcpyDeviceToHost, s));
    M_HIP(hipMemcpyAsync(calls + r0, d_calls + lq, nr * 4, hipMemcpyDeviceToHost, s));
    if (hits) M_HIP(hipMemcpyAsync(hits + r0, d_hits + lq, nr * 4, hipMemcpyDeviceToHost, s));
    M_HIP(hipMemcpyAsync(run_off + r0, (uint64_t *)r.roff.p + lq, nr * 8, hipMemcpyDeviceToHost, s));
    M_HIP(hipMemcpyAsync(run_cnt + r0, (uint32_t *)r.rcnt.p + lq, nr * 4, hipMemcpyDeviceToHost, s));
    M_HIP(hipStreamSynchronize(s));
    if (total > runs_cap) return mfail(KU_EHIP, "run-length encoder overflowed its bound");
    totals[r.rank] = total;
    return KU_OK;
  }));
  // one run array for the caller: rank r's runs follow those of the ranks before it
  uint64_t base = 0;
  for (uint32_t q = 0; q < W; ++q) {
    m->ranks[q].run_base = base;
    m->ranks[q].n_runs = totals[q];
    if (base)
      for (uint64_t i = rb[q]; i < rb[q + 1]; ++i) run_off[i] += base;
    base += totals[q];
  }
  *n_runs = base;
  return KU_OK;
}

extern "C" int ku_mgpu_fetch_runs(ku_mgpu *m, ku_run *runs, uint64_t n_runs) {
  if (!m) return mfail(KU_EINVAL, "ku_mgpu_fetch_runs: null argument");
  uint64_t total = 0;
  for (auto &r : m->ranks) total += r.n_runs;
  if (n_runs > total) return mfail(KU_EINVAL, "ku_mgpu_fetch_runs: the last batch holds " + std::to_string(total) + " runs");
  if (n_runs == 0) return KU_OK;
  if (!runs) return mfail(KU_EINVAL, "ku_mgpu_fetch_runs: null buffer");
  return run_all(m, [&](ku_mgpu::Rank &r) -> int {
    if (r.run_base >= n_runs || r.n_runs == 0) return KU_OK;
    const uint64_t n = std::min(r.n_runs, n_runs - r.run_base);
    if (m->flags & KU_MGPU_REPLICAS) return ku_fetch_runs(r.ctx, runs + r.run_base, n);  // the runs lie in the rank's context
    hipStream_t s = ku_ctx_stream_of(r.ctx);
    M_HIP(hipMemcpyAsync(runs + r.run_base, r.runs.p, n * 8, hipMemcpyDeviceToHost, s));
    M_HIP(hipStreamSynchronize(s));
    return KU_OK;
  });
}

extern "C" int ku_mgpu_reduce_state(ku_mgpu *m, void *const *streams) {
  if (!m) return mfail(KU_EINVAL, "ku_mgpu_reduce_state: null argument");
  if (!m->tax_set) return mfail(KU_ESTATE, "ku_mgpu_reduce_state: no taxonomy set");
  if (m->reduced) return mfail(KU_ESTATE, "ku_mgpu_reduce_state: the state is reduced already (another call would add the counters again)");
  M_TRY(run_all(m, [&](ku_mgpu::Rank &r) -> int {
    hipStream_t s = (streams && streams[r.local]) ? (hipStream_t)streams[r.local] : ku_ctx_stream_of(r.ctx);
    int st = comm_allreduce_state(m, r, s);
    if (st == KU_OK && m->exact) st = comm_allreduce_exact(m, r, s);
    return st;
  }));
  bool degraded = m->sparse && ku_mgpu_sparse_state(m) != 1;
  if (m->sparse && !degraded) {
    // the emulation's state of the whole run ends up in rank 0's context: the last unit closes where it is, a taxon is
    // dense if any rank found it dense, the sparse taxa's sets are the union of the ranks' sets (hyperloglogplus.cpp:586-665)
    const size_t ns = [&] { ku_counts_dims d{}; (void)ku_counts_dims_get(m->ranks[0].ctx, &d); return (size_t)d.n_slots; }();
    std::vector<uint32_t> dense(ns, 0), one(ns);
    int fs = KU_OK;
    for (auto &r : m->ranks) {
      if (hipSetDevice(r.device) != hipSuccess) return mfail(KU_EHIP, "hipSetDevice failed");
      fs = ku_ctx_sparse_finish(r.ctx, one.data());  // (the keys of the fast path's log join the rank's set here)
      if (fs != KU_OK) break;
      for (size_t i = 0; i < ns; ++i) dense[i] |= one[i];
    }
    for (auto &r : m->ranks) {
      if (fs != KU_OK) break;
      if (hipSetDevice(r.device) != hipSuccess) return mfail(KU_EHIP, "hipSetDevice failed");
      fs = ku_ctx_sparse_set_dense(r.ctx, dense.data());
    }
    for (uint32_t q = 1; q < m->n_local && fs == KU_OK; ++q) fs = ku_ctx_sparse_absorb(m->ranks[0].ctx, m->ranks[q].ctx);
    if (fs == KU_ENOMEM) degraded = true;  // no room for a rank's set (or the union): dense estimates, as below
    else if (fs != KU_OK) return fs;
    m->acc_nt = 0;
    m->open_rank = -1;
  }
  if (m->sparse && degraded) {
    // a rank ran out of memory for its tables: no rank's sets are the run's any more -- the report falls back to the
    // dense registers everywhere (ku_mgpu_sparse_state says 2)
    for (auto &r : m->ranks) {
      if (hipSetDevice(r.device) != hipSuccess) return mfail(KU_EHIP, "hipSetDevice failed");
      M_TRY(ku_ctx_disable_sparse(r.ctx));
    }
    m->sparse = false;
    m->sparse_gave_up = true;
  }
  m->reduced = true;
  return KU_OK;
}

extern "C" int ku_mgpu_count_taxons(ku_mgpu *m, uint32_t *taxids, uint64_t *counts, uint64_t *n) {
  if (!m || !n) return mfail(KU_EINVAL, "ku_mgpu_count_taxons: null argument");
  if (!single_process(m)) return mfail(KU_EUNSUP, "ku_mgpu_count_taxons: single-process groups only");
  std::vector<std::pair<uint32_t, uint64_t>> acc;
  const uint32_t n_src = (m->flags & KU_MGPU_REPLICAS) ? 1 : m->n_local;
  for (uint32_t i = 0; i < n_src; ++i) {
    uint64_t c = 0;
    M_TRY(ku_ctx_count_taxons(m->ranks[i].ctx, nullptr, nullptr, &c));
    std::vector<uint32_t> t(c + 1);
    std::vector<uint64_t> v(c + 1);
    uint64_t cap = c;
    M_TRY(ku_ctx_count_taxons(m->ranks[i].ctx, t.data(), v.data(), &cap));
    for (uint64_t j = 0; j < cap; ++j) acc.emplace_back(t[j], v[j]);
  }
  std::sort(acc.begin(), acc.end());
  std::vector<std::pair<uint32_t, uint64_t>> out;
  for (auto &kv : acc) {
    if (!out.empty() && out.back().first == kv.first) out.back().second += kv.second;
    else out.push_back(kv);
  }
  if (taxids && counts) {
    if (*n < out.size()) return mfail(KU_EINVAL, "output arrays too small");
    for (size_t j = 0; j < out.size(); ++j) { taxids[j] = out[j].first; counts[j] = out[j].second; }
  }
  *n = out.size();
  return KU_OK;
}
