// stub_rccl.cpp -- TEST INFRASTRUCTURE, not product: a stand-in for librccl that lets SEVERAL ranks share ONE GPU, so that the
// library's own communicator (csrc/comm.cpp: file rendezvous, the initialisation watchdog thread, frcnn_allreduce_f32 / _f64,
// frcnn_broadcast_f32 and settle()'s ncclInProgress polling) can run with world size 2 on the one-GPU boxes this project is
// tested on.  Real RCCL refuses two ranks on one device.  Bound through FRCNN_RCCL_LIB (comm.cpp rccl_load); built by
// __graft_entry__.build() into tests/_stub/librccl_stub.so; used by tests/test_gpu_comm_stub.py only.
//
// The collectives move data through a POSIX shared-memory segment named by the unique id: every rank copies its operand to the
// host in chunks, the ranks meet at a barrier, every rank sums the chunks of all ranks in rank order (so every rank gets the same
// bits) and copies the result back.  Everything is synchronous inside the call; STUB_RCCL_INPROGRESS=k makes every collective
// ANSWER ncclInProgress all the same and ncclCommGetAsyncError report ncclInProgress k more times before ncclSuccess -- the
// behaviour of a communicator that was made non-blocking behind the caller's back, which settle() has to absorb.
#include <fcntl.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

namespace {

constexpr size_t kChunk = 4u << 20;   // bytes per rank and round
constexpr int kMaxRanks = 8;

struct Shared {
  std::atomic<int> joined;
  std::atomic<int> arrive;
  std::atomic<int> generation;
  char pad[64 - 3 * sizeof(std::atomic<int>)];
  unsigned char data[kMaxRanks][kChunk];
};

struct Comm {
  Shared* sh = nullptr;
  int nranks = 0, rank = 0, device = 0;
  int pending_polls = 0;     // ncclCommGetAsyncError answers ncclInProgress this many more times
  long polls_seen = 0;
  char name[64];
};

void barrier(Comm* c) {
  Shared* s = c->sh;
  const int gen = s->generation.load(std::memory_order_acquire);
  if (s->arrive.fetch_add(1, std::memory_order_acq_rel) + 1 == c->nranks) {
    s->arrive.store(0, std::memory_order_relaxed);
    s->generation.fetch_add(1, std::memory_order_acq_rel);
  } else {
    while (s->generation.load(std::memory_order_acquire) == gen) usleep(20);
  }
}

int inprogress_polls() {
  const char* e = getenv("STUB_RCCL_INPROGRESS");
  return e ? atoi(e) : 0;
}

template <class T>
void sum_into(T* out, Comm* c, size_t n) {
  for (size_t i = 0; i < n; ++i) {
    T v = reinterpret_cast<const T*>(c->sh->data[0])[i];
    for (int r = 1; r < c->nranks; ++r) v += reinterpret_cast<const T*>(c->sh->data[r])[i];
    out[i] = v;
  }
}

// every rank ends with op(all ranks' operands): sum over the ranks (root < 0) or the root's operand
ncclResult_t exchange(const void* send, void* recv, size_t count, ncclDataType_t dt, int root, Comm* c, hipStream_t stream) {
  const size_t es = dt == ncclFloat64 ? 8 : 4;
  if (dt != ncclFloat32 && dt != ncclFloat64) return ncclInvalidArgument;
  if (hipStreamSynchronize(stream) != hipSuccess) return ncclUnhandledCudaError;
  std::vector<unsigned char> host(kChunk);
  const size_t total = count * es;
  for (size_t off = 0; off < total; off += kChunk) {
    const size_t nb = total - off < kChunk ? total - off : kChunk;
    if (hipMemcpy(c->sh->data[c->rank], (const char*)send + off, nb, hipMemcpyDeviceToHost) != hipSuccess) return ncclUnhandledCudaError;
    barrier(c);
    if (root >= 0) memcpy(host.data(), c->sh->data[root], nb);
    else if (es == 4) sum_into(reinterpret_cast<float*>(host.data()), c, nb / 4);
    else sum_into(reinterpret_cast<double*>(host.data()), c, nb / 8);
    barrier(c);   // (every rank has read the slots before anyone overwrites its own)
    if (hipMemcpy((char*)recv + off, host.data(), nb, hipMemcpyHostToDevice) != hipSuccess) return ncclUnhandledCudaError;
  }
  const int k = inprogress_polls();
  if (k > 0) { c->pending_polls = k; return ncclInProgress; }
  return ncclSuccess;
}

}  // namespace

extern "C" {

ncclResult_t ncclGetUniqueId(ncclUniqueId* id) {
  memset(id, 0, sizeof(*id));
  snprintf(id->internal, sizeof(id->internal), "/frcnn_stub_rccl_%d_%ld", (int)getpid(), (long)random());
  return ncclSuccess;
}

ncclResult_t ncclCommInitRank(ncclComm_t* comm, int nranks, ncclUniqueId id, int rank) {
  if (nranks < 1 || nranks > kMaxRanks || rank < 0 || rank >= nranks) return ncclInvalidArgument;
  Comm* c = new Comm();
  c->nranks = nranks; c->rank = rank;
  strncpy(c->name, id.internal, sizeof(c->name) - 1);
  (void)hipGetDevice(&c->device);
  int fd = shm_open(c->name, O_CREAT | O_RDWR, 0600);
  if (fd < 0) { delete c; return ncclSystemError; }
  if (ftruncate(fd, sizeof(Shared)) != 0) { close(fd); delete c; return ncclSystemError; }
  void* p = mmap(nullptr, sizeof(Shared), PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
  close(fd);
  if (p == MAP_FAILED) { delete c; return ncclSystemError; }
  c->sh = static_cast<Shared*>(p);   // (a fresh segment is zero-filled: counters start at 0)
  c->sh->joined.fetch_add(1, std::memory_order_acq_rel);
  while (c->sh->joined.load(std::memory_order_acquire) < nranks) usleep(200);   // collective: a missing rank blocks here
  *comm = reinterpret_cast<ncclComm_t>(c);
  return ncclSuccess;
}

ncclResult_t ncclCommDestroy(ncclComm_t comm) {
  Comm* c = reinterpret_cast<Comm*>(comm);
  if (!c) return ncclSuccess;
  barrier(c);
  munmap(c->sh, sizeof(Shared));
  if (c->rank == 0) shm_unlink(c->name);
  if (const char* f = getenv("STUB_RCCL_POLL_LOG")) {   // (tests: how many times settle() had to ask)
    char path[512];
    snprintf(path, sizeof(path), "%s.%d", f, c->rank);
    if (FILE* fp = fopen(path, "w")) { fprintf(fp, "%ld\n", c->polls_seen); fclose(fp); }
  }
  delete c;
  return ncclSuccess;
}

ncclResult_t ncclAllReduce(const void* send, void* recv, size_t count, ncclDataType_t dt, ncclRedOp_t op, ncclComm_t comm, hipStream_t stream) {
  if (op != ncclSum) return ncclInvalidArgument;
  return exchange(send, recv, count, dt, -1, reinterpret_cast<Comm*>(comm), stream);
}

ncclResult_t ncclBroadcast(const void* send, void* recv, size_t count, ncclDataType_t dt, int root, ncclComm_t comm, hipStream_t stream) {
  Comm* c = reinterpret_cast<Comm*>(comm);
  if (root < 0 || root >= c->nranks) return ncclInvalidArgument;
  return exchange(send, recv, count, dt, root, c, stream);
}

ncclResult_t ncclCommGetAsyncError(ncclComm_t comm, ncclResult_t* state) {
  Comm* c = reinterpret_cast<Comm*>(comm);
  ++c->polls_seen;
  if (c->pending_polls > 0) { --c->pending_polls; *state = ncclInProgress; }
  else *state = ncclSuccess;
  return ncclSuccess;
}

const char* ncclGetErrorString(ncclResult_t r) {
  switch (r) {
    case ncclSuccess: return "no error (stub)";
    case ncclInProgress: return "in progress (stub)";
    case ncclInvalidArgument: return "invalid argument (stub)";
    case ncclSystemError: return "system error (stub)";
    default: return "error (stub)";
  }
}

ncclResult_t ncclCommCount(const ncclComm_t comm, int* n) { *n = reinterpret_cast<const Comm*>(comm)->nranks; return ncclSuccess; }
ncclResult_t ncclCommUserRank(const ncclComm_t comm, int* r) { *r = reinterpret_cast<const Comm*>(comm)->rank; return ncclSuccess; }
ncclResult_t ncclCommCuDevice(const ncclComm_t comm, int* d) { *d = reinterpret_cast<const Comm*>(comm)->device; return ncclSuccess; }

}  // extern "C"
