// ku_rccl_shim.cpp -- TEST INFRASTRUCTURE, never loaded by default.
//
// The multi-GPU driver (krakenuniq_amd/csrc/ku_mgpu.cpp) binds RCCL with dlopen; KU_RCCL_LIB=<this library> makes it bind
// these stand-ins instead.  They implement the dozen entry points the driver uses BETWEEN PROCESSES through files in
// /dev/shm -- so that its peer paths (grouped ncclSend / ncclRecv all-to-alls, the scatter of the read slices, the
// all-gathers of the routing counts, the all-reduce of the per-taxon state) run with real peers on a box with a single
// GPU, where RCCL itself refuses two ranks on one device.  Nothing here is fast or asynchronous: an operation synchronises
// its stream, stages through host memory and blocks until its peers have delivered; what it keeps is the semantics the
// driver relies on -- program order per stream, grouped calls that may send and receive in any order without deadlock,
// messages between a pair of ranks matched in issue order.
#include <dirent.h>
#include <fcntl.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <sys/stat.h>
#include <unistd.h>

#include <chrono>
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
};
}  // namespace

struct ncclComm {
  int rank = 0, n = 0;
  std::string dir;
  std::vector<unsigned long long> sseq, rseq;  // messages sent to / received from each peer so far
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

bool write_msg(ncclComm *c, int dst, const void *host, size_t bytes) {
  const std::string name = c->dir + "/m_" + std::to_string(c->rank) + "_" + std::to_string(dst) + "_" + std::to_string(c->sseq[dst]++);
  const std::string tmp = name + ".tmp";
  FILE *f = fopen(tmp.c_str(), "wb");
  if (!f) return false;
  const bool ok = bytes == 0 || fwrite(host, 1, bytes, f) == bytes;
  fclose(f);
  return ok && rename(tmp.c_str(), name.c_str()) == 0;
}
bool read_msg(ncclComm *c, int src, void *host, size_t bytes) {
  const std::string name = c->dir + "/m_" + std::to_string(src) + "_" + std::to_string(c->rank) + "_" + std::to_string(c->rseq[src]++);
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
ncclResult_t submit(const Op &o) {
  if (!o.comm || dt_size(o.dt) == 0) return ncclInvalidArgument;
  if ((o.kind == Op::SEND || o.kind == Op::RECV || o.kind == Op::BROADCAST || o.kind == Op::REDUCE) && (o.peer < 0 || o.peer >= o.comm->n))
    return ncclInvalidArgument;
  if (g_depth > 0) {
    g_queue.push_back(o);
    return ncclSuccess;
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
