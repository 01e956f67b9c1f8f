// ku_rccl_shim.cpp -- TEST INFRASTRUCTURE, never loaded by default.
//
// The multi-GPU driver (krakenuniq_amd/csrc/ku_mgpu.cpp) binds RCCL with dlopen; KU_RCCL_LIB=<this library> makes it bind
// these stand-ins instead.  They implement the dozen entry points the driver uses BETWEEN PROCESSES through files in
// /dev/shm -- so that its peer paths (grouped ncclSend / ncclRecv all-to-alls, the scatter of the read slices, the
// all-gathers of the routing counts, the all-reduce of the per-taxon state) run with real peers on a box with a single
// GPU, where RCCL itself refuses two ranks on one device.  Nothing here is fast; what it keeps is the semantics the driver
// relies on -- program order per stream, grouped calls that may send and receive in any order without deadlock, messages
// between a pair of ranks matched in issue order.
//
// Round 5: the operations are ASYNCHRONOUS, as RCCL's are.  A call (or a group) only ENQUEUES on its stream and returns:
//   copies of the send buffers into page-locked staging memory (stream order: they see what the stream produced before),
//   a host function (hipLaunchHostFunc) that writes the messages, then waits for the peers' and folds reductions -- the
//   stream stands still meanwhile, as it does inside a collective --, the copies of the staged receive data into place,
//   and a second host function that hands the staging memory back.
// So a buffer the driver reuses too early, or a stream that does not wait for the one that produced the data, shows as wrong
// data -- the synchronous stand-in of round 4 (KU_SHIM_SYNC=1 brings it back: every operation synchronised its stream and
// blocked the caller) hid exactly that.  KU_SHIM_JITTER_US=N delays every exchange by a random time below N microseconds, a
// different one on every rank, before the sends and again behind the receives: peers that run ahead or lag behind;
// KU_SHIM_LAG_OTHER_STREAMS_US=N makes the exchanges of every stream but the communicator's first finish N us late.  Message
// numbers are taken when the call is ISSUED, so operations of two streams may complete in any order.
#include <dirent.h>
#include <fcntl.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <sys/stat.h>
#include <unistd.h>

#include <atomic>
#include <chrono>
#include <mutex>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

namespace {
struct Op {
  enum Kind { SEND, RECV, ALLGATHER, ALLREDUCE, BROADCAST, REDUCE } kind;
  const void *send;
  void *recv;
  size_t count;
  ncclDataType_t dt;
  ncclRedOp_t op;
  int peer;  // peer / root
  hipStream_t stream;
  struct ncclComm *comm;
  std::vector<long long> sq, rq;  // asynchronous form: number of the message to / from each peer (-1: none), taken at issue
};
struct Group;
}  // namespace

struct ncclComm {
  int rank = 0, n = 0;
  std::string dir;
  std::vector<unsigned long long> sseq, rseq;  // messages sent to / received from each peer so far
  std::atomic<bool> failed{false};             // an exchange went wrong inside a host function: every later call says so
  std::mutex mu;
  std::vector<Group *> groups;                 // enqueued exchanges (their staging memory goes back once they are through)
  unsigned long long jitter_state = 0;
  hipStream_t first_stream = nullptr;          // the stream of the first exchange (KU_SHIM_LAG_OTHER_STREAMS_US)
  bool have_first_stream = false;
};

namespace {
thread_local int g_depth = 0;
thread_local std::vector<Op> g_queue;

size_t dt_size(ncclDataType_t dt) {
  switch (dt) {
    case ncclInt8: case ncclUint8: return 1;
    case ncclInt32: case ncclUint32: case ncclFloat32: return 4;
    case ncclInt64: case ncclUint64: case ncclFloat64: return 8;
    default: return 0;
  }
}
double timeout_s() { const char *e = getenv("KU_SHIM_TIMEOUT"); return e ? atof(e) : 120.0; }

bool write_msg(ncclComm *c, int dst, const void *host, size_t bytes, long long seq = -1) {
  const std::string name = c->dir + "/m_" + std::to_string(c->rank) + "_" + std::to_string(dst) + "_" + std::to_string(seq >= 0 ? (unsigned long long)seq : c->sseq[dst]++);
  const std::string tmp = name + ".tmp";
  FILE *f = fopen(tmp.c_str(), "wb");
  if (!f) return false;
  const bool ok = bytes == 0 || fwrite(host, 1, bytes, f) == bytes;
  fclose(f);
  return ok && rename(tmp.c_str(), name.c_str()) == 0;
}
bool read_msg(ncclComm *c, int src, void *host, size_t bytes, long long seq = -1) {
  const std::string name = c->dir + "/m_" + std::to_string(src) + "_" + std::to_string(c->rank) + "_" + std::to_string(seq >= 0 ? (unsigned long long)seq : c->rseq[src]++);
  const auto t0 = std::chrono::steady_clock::now();
  struct stat sb;
  while (stat(name.c_str(), &sb) != 0) {
    if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > timeout_s()) {
      fprintf(stderr, "[ku_rccl_shim] rank %d: no message %s after %.0f s\n", c->rank, name.c_str(), timeout_s());
      return false;
    }
    std::this_thread::sleep_for(std::chrono::microseconds(200));
  }
  if ((size_t)sb.st_size != bytes) {
    fprintf(stderr, "[ku_rccl_shim] rank %d: message %s has %lld bytes, the receive expects %zu\n", c->rank, name.c_str(), (long long)sb.st_size, bytes);
    return false;
  }
  FILE *f = fopen(name.c_str(), "rb");
  if (!f) return false;
  const bool ok = bytes == 0 || fread(host, 1, bytes, f) == bytes;
  fclose(f);
  unlink(name.c_str());
  return ok;
}
bool d2h(std::vector<char> &h, const void *d, size_t bytes, hipStream_t s) {
  h.resize(bytes);
  if (hipStreamSynchronize(s) != hipSuccess) return false;
  return bytes == 0 || hipMemcpy(h.data(), d, bytes, hipMemcpyDeviceToHost) == hipSuccess;
}
bool h2d(void *d, const void *h, size_t bytes, hipStream_t s) {
  if (hipStreamSynchronize(s) != hipSuccess) return false;
  return bytes == 0 || hipMemcpy(d, h, bytes, hipMemcpyHostToDevice) == hipSuccess;
}
template <typename T> void fold(T *acc, const T *x, size_t n, ncclRedOp_t op) {
  for (size_t i = 0; i < n; ++i) acc[i] = op == ncclMax ? (acc[i] > x[i] ? acc[i] : x[i]) : (op == ncclMin ? (acc[i] < x[i] ? acc[i] : x[i]) : (T)(acc[i] + x[i]));
}
bool fold_any(void *acc, const void *x, size_t n, ncclDataType_t dt, ncclRedOp_t op) {
  if (op != ncclMax && op != ncclMin && op != ncclSum) return false;
  switch (dt) {
    case ncclUint8: fold((uint8_t *)acc, (const uint8_t *)x, n, op); return true;
    case ncclInt8: fold((int8_t *)acc, (const int8_t *)x, n, op); return true;
    case ncclUint32: fold((uint32_t *)acc, (const uint32_t *)x, n, op); return true;
    case ncclInt32: fold((int32_t *)acc, (const int32_t *)x, n, op); return true;
    case ncclUint64: fold((uint64_t *)acc, (const uint64_t *)x, n, op); return true;
    case ncclInt64: fold((int64_t *)acc, (const int64_t *)x, n, op); return true;
    default: return false;
  }
}

// the sending half of an operation (never blocks), then its receiving half
bool run_sends(const Op &o) {
  ncclComm *c = o.comm;
  const size_t bytes = o.count * dt_size(o.dt);
  std::vector<char> h;
  switch (o.kind) {
    case Op::SEND: return d2h(h, o.send, bytes, o.stream) && write_msg(c, o.peer, h.data(), bytes);
    case Op::RECV: return true;
    case Op::ALLGATHER:
    case Op::ALLREDUCE:
      if (!d2h(h, o.send, bytes, o.stream)) return false;
      for (int p = 0; p < c->n; ++p)
        if (p != c->rank && !write_msg(c, p, h.data(), bytes)) return false;
      return true;
    case Op::BROADCAST:
      if (c->rank != o.peer) return true;
      if (!d2h(h, o.send, bytes, o.stream)) return false;
      for (int p = 0; p < c->n; ++p)
        if (p != c->rank && !write_msg(c, p, h.data(), bytes)) return false;
      return true;
    case Op::REDUCE:
      if (c->rank == o.peer) return true;
      return d2h(h, o.send, bytes, o.stream) && write_msg(c, o.peer, h.data(), bytes);
  }
  return false;
}
bool run_recvs(const Op &o) {
  ncclComm *c = o.comm;
  const size_t bytes = o.count * dt_size(o.dt);
  std::vector<char> h(bytes), mine;
  switch (o.kind) {
    case Op::SEND: return true;
    case Op::RECV: return read_msg(c, o.peer, h.data(), bytes) && h2d(o.recv, h.data(), bytes, o.stream);
    case Op::ALLGATHER:
      if (!d2h(mine, o.send, bytes, o.stream)) return false;
      for (int p = 0; p < c->n; ++p) {
        if (p == c->rank) memcpy(h.data(), mine.data(), bytes);
        else if (!read_msg(c, p, h.data(), bytes)) return false;
        if (!h2d((char *)o.recv + (size_t)p * bytes, h.data(), bytes, o.stream)) return false;
      }
      return true;
    case Op::ALLREDUCE:
      if (!d2h(mine, o.send, bytes, o.stream)) return false;
      for (int p = 0; p < c->n; ++p) {
        if (p == c->rank) continue;
        if (!read_msg(c, p, h.data(), bytes) || !fold_any(mine.data(), h.data(), o.count, o.dt, o.op)) return false;
      }
      return h2d(o.recv, mine.data(), bytes, o.stream);
    case Op::BROADCAST:
      if (c->rank == o.peer) {
        if (o.recv == o.send) return true;
        return d2h(mine, o.send, bytes, o.stream) && h2d(o.recv, mine.data(), bytes, o.stream);
      }
      return read_msg(c, o.peer, h.data(), bytes) && h2d(o.recv, h.data(), bytes, o.stream);
    case Op::REDUCE:
      if (c->rank != o.peer) return true;
      if (!d2h(mine, o.send, bytes, o.stream)) return false;
      for (int p = 0; p < c->n; ++p) {
        if (p == c->rank) continue;
        if (!read_msg(c, p, h.data(), bytes) || !fold_any(mine.data(), h.data(), o.count, o.dt, o.op)) return false;
      }
      return h2d(o.recv, mine.data(), bytes, o.stream);
  }
  return false;
}
// ---------------------------------------------------------------------------- the asynchronous form (default)
bool sync_mode() { static const bool v = getenv("KU_SHIM_SYNC") && atoi(getenv("KU_SHIM_SYNC")); return v; }
struct Group {
  ncclComm *comm = nullptr;
  hipStream_t stream = nullptr;
  std::vector<Op> ops;
  std::vector<void *> h_send, h_recv;  // page-locked staging per operation (nullptr: none)
  std::vector<size_t> recv_bytes;
  std::atomic<int> done{0};
};
// KU_SHIM_LAG_OTHER_STREAMS_US=N: every exchange on a stream other than the communicator's first one finishes N microseconds
// late -- deterministically: a caller that does not wait for its second stream reads what is not there yet
void lag_other_streams(const Group *g) {
  static const long lag_us = getenv("KU_SHIM_LAG_OTHER_STREAMS_US") ? atol(getenv("KU_SHIM_LAG_OTHER_STREAMS_US")) : 0;
  if (lag_us > 0 && g->comm->have_first_stream && g->stream != g->comm->first_stream) std::this_thread::sleep_for(std::chrono::microseconds(lag_us));
}
void jitter(ncclComm *c) {
  static const long max_us = getenv("KU_SHIM_JITTER_US") ? atol(getenv("KU_SHIM_JITTER_US")) : 0;
  if (max_us <= 0) return;
  unsigned long long x;
  {
    std::lock_guard<std::mutex> l(c->mu);
    x = c->jitter_state = c->jitter_state * 6364136223846793005ull + 1442695040888963407ull + (unsigned long long)c->rank * 7919;
  }
  std::this_thread::sleep_for(std::chrono::microseconds((long)((x >> 33) % (unsigned long long)max_us)));
}
// numbers of the messages an operation will send / expect, in issue order (the caller's thread)
void take_numbers(Op &o) {
  ncclComm *c = o.comm;
  o.sq.assign(c->n, -1);
  o.rq.assign(c->n, -1);
  auto snd = [&](int p) { o.sq[p] = (long long)c->sseq[p]++; };
  auto rcv = [&](int p) { o.rq[p] = (long long)c->rseq[p]++; };
  switch (o.kind) {
    case Op::SEND: snd(o.peer); break;
    case Op::RECV: rcv(o.peer); break;
    case Op::ALLGATHER: case Op::ALLREDUCE:
      for (int p = 0; p < c->n; ++p) if (p != c->rank) { snd(p); rcv(p); }
      break;
    case Op::BROADCAST:
      if (c->rank == o.peer) { for (int p = 0; p < c->n; ++p) if (p != c->rank) snd(p); }
      else rcv(o.peer);
      break;
    case Op::REDUCE:
      if (c->rank == o.peer) { for (int p = 0; p < c->n; ++p) if (p != c->rank) rcv(p); }
      else snd(o.peer);
      break;
  }
}
// runs on a runtime thread while the stream stands still: no HIP calls in here
void exchange_host(void *arg) {
  Group *g = (Group *)arg;
  ncclComm *c = g->comm;
  bool ok = true;
  jitter(c);
  for (size_t i = 0; i < g->ops.size() && ok; ++i) {  // every send of the group is out ...
    const Op &o = g->ops[i];
    const size_t bytes = o.count * dt_size(o.dt);
    for (int p = 0; p < c->n && ok; ++p)
      if (o.sq[p] >= 0) ok = write_msg(c, p, g->h_send[i], bytes, o.sq[p]);
  }
  for (size_t i = 0; i < g->ops.size() && ok; ++i) {  // ... before the first receive waits
    const Op &o = g->ops[i];
    const size_t bytes = o.count * dt_size(o.dt);
    char *hr = (char *)g->h_recv[i];
    switch (o.kind) {
      case Op::SEND: break;
      case Op::RECV: ok = read_msg(c, o.peer, hr, bytes, o.rq[o.peer]); break;
      case Op::ALLGATHER:
        for (int p = 0; p < c->n && ok; ++p) {
          if (p == c->rank) memcpy(hr + (size_t)p * bytes, g->h_send[i], bytes);
          else ok = read_msg(c, p, hr + (size_t)p * bytes, bytes, o.rq[p]);
        }
        break;
      case Op::ALLREDUCE:
      case Op::REDUCE:
        if (o.kind == Op::REDUCE && c->rank != o.peer) break;
        memcpy(hr, g->h_send[i], bytes);
        {
          std::vector<char> tmp(bytes);
          for (int p = 0; p < c->n && ok; ++p) {
            if (p == c->rank) continue;
            ok = read_msg(c, p, tmp.data(), bytes, o.rq[p]) && fold_any(hr, tmp.data(), o.count, o.dt, o.op);
          }
        }
        break;
      case Op::BROADCAST:
        if (c->rank == o.peer) { if (hr) memcpy(hr, g->h_send[i], bytes); }
        else ok = read_msg(c, o.peer, hr, bytes, o.rq[o.peer]);
        break;
    }
  }
  jitter(c);
  lag_other_streams(g);
  if (!ok) {
    c->failed = true;
    fprintf(stderr, "[ku_rccl_shim] rank %d: an exchange failed (a peer did not deliver, or a message of the wrong size)\n", c->rank);
  }
}
void release_host(void *arg) { ((Group *)arg)->done = 1; }
void collect(ncclComm *c, bool all) {  // staging memory of the exchanges that are through (caller's thread: HIP calls allowed)
  std::vector<Group *> keep, gone;
  {
    std::lock_guard<std::mutex> l(c->mu);
    for (Group *g : c->groups) (g->done.load() || all ? gone : keep).push_back(g);
    c->groups.swap(keep);
  }
  for (Group *g : gone) {
    for (void *p : g->h_send) if (p) (void)hipHostFree(p);
    for (void *p : g->h_recv) if (p) (void)hipHostFree(p);
    delete g;
  }
}
ncclResult_t enqueue_group(std::vector<Op> &ops) {
  if (ops.empty()) return ncclSuccess;
  ncclComm *c = ops[0].comm;
  hipStream_t s = ops[0].stream;
  for (const Op &o : ops)
    if (o.comm != c || o.stream != s) return ncclInvalidUsage;  // (the driver's groups are one communicator on one stream)
  if (c->failed) return ncclSystemError;
  if (!c->have_first_stream) { c->first_stream = s; c->have_first_stream = true; }
  collect(c, false);
  Group *g = new Group;
  g->comm = c;
  g->stream = s;
  g->ops.swap(ops);
  const size_t n = g->ops.size();
  g->h_send.assign(n, nullptr);
  g->h_recv.assign(n, nullptr);
  g->recv_bytes.assign(n, 0);
  bool ok = true;
  for (size_t i = 0; i < n && ok; ++i) {
    Op &o = g->ops[i];
    take_numbers(o);
    const size_t bytes = o.count * dt_size(o.dt);
    const bool root = c->rank == o.peer;
    const bool has_send = o.kind == Op::SEND || o.kind == Op::ALLGATHER || o.kind == Op::ALLREDUCE || o.kind == Op::REDUCE || (o.kind == Op::BROADCAST && root);
    size_t rb = 0;
    if (o.kind == Op::RECV || o.kind == Op::ALLREDUCE || (o.kind == Op::REDUCE && root)) rb = bytes;
    if (o.kind == Op::ALLGATHER) rb = bytes * (size_t)c->n;
    if (o.kind == Op::BROADCAST && (!root || o.recv != o.send)) rb = bytes;
    if (has_send) {
      ok = hipHostMalloc(&g->h_send[i], bytes ? bytes : 1, hipHostMallocDefault) == hipSuccess;
      if (ok && bytes) ok = hipMemcpyAsync(g->h_send[i], o.send, bytes, hipMemcpyDeviceToHost, s) == hipSuccess;
    }
    if (ok && rb) ok = hipHostMalloc(&g->h_recv[i], rb, hipHostMallocDefault) == hipSuccess;
    g->recv_bytes[i] = rb;
  }
  if (ok) ok = hipLaunchHostFunc(s, exchange_host, g) == hipSuccess;
  for (size_t i = 0; i < n && ok; ++i)
    if (g->recv_bytes[i] && g->ops[i].recv) ok = hipMemcpyAsync(g->ops[i].recv, g->h_recv[i], g->recv_bytes[i], hipMemcpyHostToDevice, s) == hipSuccess;
  if (ok) ok = hipLaunchHostFunc(s, release_host, g) == hipSuccess;
  {
    std::lock_guard<std::mutex> l(c->mu);
    c->groups.push_back(g);
  }
  if (!ok) { c->failed = true; return ncclSystemError; }
  return ncclSuccess;
}

ncclResult_t submit(const Op &o) {
  if (!o.comm || dt_size(o.dt) == 0) return ncclInvalidArgument;
  if ((o.kind == Op::SEND || o.kind == Op::RECV || o.kind == Op::BROADCAST || o.kind == Op::REDUCE) && (o.peer < 0 || o.peer >= o.comm->n))
    return ncclInvalidArgument;
  if (o.comm->failed) return ncclSystemError;
  if (g_depth > 0) {
    g_queue.push_back(o);
    return ncclSuccess;
  }
  if (!sync_mode()) {
    std::vector<Op> one{o};
    return enqueue_group(one);
  }
  return run_sends(o) && run_recvs(o) ? ncclSuccess : ncclSystemError;
}
}  // namespace

extern "C" {
ncclResult_t ncclGetUniqueId(ncclUniqueId *id) {
  if (!id) return ncclInvalidArgument;
  memset(id, 0, sizeof *id);
  unsigned long long t = (unsigned long long)std::chrono::steady_clock::now().time_since_epoch().count() ^ ((unsigned long long)getpid() << 40);
  snprintf(id->internal, sizeof id->internal, "ku_shim_%llx", t);
  return ncclSuccess;
}
ncclResult_t ncclCommInitRank(ncclComm_t *comm, int nranks, ncclUniqueId id, int rank) {
  if (!comm || nranks < 1 || rank < 0 || rank >= nranks) return ncclInvalidArgument;
  id.internal[sizeof id.internal - 1] = 0;
  if (strncmp(id.internal, "ku_shim_", 8) != 0) return ncclInvalidArgument;
  ncclComm *c = new ncclComm();
  c->rank = rank;
  c->n = nranks;
  c->dir = std::string("/dev/shm/") + id.internal;
  c->sseq.assign(nranks, 0);
  c->rseq.assign(nranks, 0);
  mkdir(c->dir.c_str(), 0700);
  // everybody is there before anybody goes on
  const std::string me = c->dir + "/hello_" + std::to_string(rank);
  FILE *f = fopen(me.c_str(), "wb");
  if (!f) { delete c; return ncclSystemError; }
  fclose(f);
  const auto t0 = std::chrono::steady_clock::now();
  for (int p = 0; p < nranks; ++p) {
    struct stat sb;
    while (stat((c->dir + "/hello_" + std::to_string(p)).c_str(), &sb) != 0) {
      if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > timeout_s()) { delete c; return ncclSystemError; }
      std::this_thread::sleep_for(std::chrono::milliseconds(1));
    }
  }
  *comm = c;
  return ncclSuccess;
}
ncclResult_t ncclCommInitAll(ncclComm_t *, int, const int *) { return ncclInvalidUsage; }  // one rank per process here
ncclResult_t ncclCommDestroy(ncclComm_t comm) {
  if (!comm) return ncclInvalidArgument;
  (void)hipDeviceSynchronize();  // (exchanges still enqueued hold pointers into the communicator)
  collect(comm, true);
  delete comm;
  return ncclSuccess;
}
ncclResult_t ncclGroupStart() {
  ++g_depth;
  return ncclSuccess;
}
ncclResult_t ncclGroupEnd() {
  if (g_depth <= 0) return ncclInvalidUsage;
  if (--g_depth > 0) return ncclSuccess;
  std::vector<Op> q;
  q.swap(g_queue);
  if (!sync_mode()) return enqueue_group(q);
  bool ok = true;
  for (const Op &o : q) ok = ok && run_sends(o);   // every send of the group is out ...
  for (const Op &o : q) ok = ok && run_recvs(o);   // ... before the first receive waits
  return ok ? ncclSuccess : ncclSystemError;
}
ncclResult_t ncclSend(const void *sendbuff, size_t count, ncclDataType_t dt, int peer, ncclComm_t comm, hipStream_t stream) {
  return submit(Op{Op::SEND, sendbuff, nullptr, count, dt, ncclSum, peer, stream, comm});
}
ncclResult_t ncclRecv(void *recvbuff, size_t count, ncclDataType_t dt, int peer, ncclComm_t comm, hipStream_t stream) {
  return submit(Op{Op::RECV, nullptr, recvbuff, count, dt, ncclSum, peer, stream, comm});
}
ncclResult_t ncclAllGather(const void *sendbuff, void *recvbuff, size_t sendcount, ncclDataType_t dt, ncclComm_t comm, hipStream_t stream) {
  return submit(Op{Op::ALLGATHER, sendbuff, recvbuff, sendcount, dt, ncclSum, 0, stream, comm});
}
ncclResult_t ncclAllReduce(const void *sendbuff, void *recvbuff, size_t count, ncclDataType_t dt, ncclRedOp_t op, ncclComm_t comm, hipStream_t stream) {
  return submit(Op{Op::ALLREDUCE, sendbuff, recvbuff, count, dt, op, 0, stream, comm});
}
ncclResult_t ncclBroadcast(const void *sendbuff, void *recvbuff, size_t count, ncclDataType_t dt, int root, ncclComm_t comm, hipStream_t stream) {
  return submit(Op{Op::BROADCAST, sendbuff, recvbuff, count, dt, ncclSum, root, stream, comm});
}
ncclResult_t ncclReduce(const void *sendbuff, void *recvbuff, size_t count, ncclDataType_t dt, ncclRedOp_t op, int root, ncclComm_t comm, hipStream_t stream) {
  return submit(Op{Op::REDUCE, sendbuff, recvbuff, count, dt, op, root, stream, comm});
}
const char *ncclGetErrorString(ncclResult_t r) { return r == ncclSuccess ? "no error" : (r == ncclSystemError ? "ku_rccl_shim: a peer did not deliver (or a copy failed)" : "ku_rccl_shim: invalid call"); }
}
