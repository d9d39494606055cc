// comm.cpp -- the exchange step of data-parallel training behind the C ABI (SURVEY 8b last row, 8e): a
// communicator over RCCL (xGMI inside a node) plus the sum all-reduces the training step needs -- the flat
// gradient and the 8 fp64 accumulators of objective.lua:52-58, just before gradient:div (objective.lua:197-200)
// -- and the one-time weight broadcast after load_model / restore (main.lua:92-98).
//
// librccl is bound at first use (dlopen + dlsym), not at link time: a single-GPU host never loads it, and inside a
// process that already holds a copy (PyTorch-ROCm ships one under the same SONAME) the loader hands back that copy.
#include <dlfcn.h>
#include <errno.h>
#include <fcntl.h>
#include <signal.h>
#include <rccl/rccl.h>
#include <sys/stat.h>
#include <time.h>
#include <unistd.h>

#include <mutex>

#include "common.h"

namespace {

struct Rccl {
  void* so = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Broadcast)(const void*, void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
  ncclResult_t (*CommCount)(const ncclComm_t, int*) = nullptr;
  ncclResult_t (*CommUserRank)(const ncclComm_t, int*) = nullptr;
  ncclResult_t (*CommCuDevice)(const ncclComm_t, int*) = nullptr;
  std::string err;
};

Rccl g_rccl;
std::mutex g_rccl_mu;

bool rccl_load() {
  std::lock_guard<std::mutex> lk(g_rccl_mu);
  if (g_rccl.so) return true;
  const char* names[] = {getenv("FRCNN_RCCL_LIB"), "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
  void* so = nullptr;
  for (const char* n : names) {
    if (!n || !*n) continue;
    so = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
    if (so) break;
    g_rccl.err = dlerror();
  }
  if (!so) return false;
#define FR_SYM(field, name)                                                \
  *(void**)(&g_rccl.field) = dlsym(so, name);                              \
  if (!g_rccl.field) { g_rccl.err = "librccl lacks " name; dlclose(so); return false; }
  FR_SYM(GetUniqueId, "ncclGetUniqueId")
  FR_SYM(CommInitRank, "ncclCommInitRank")
  FR_SYM(CommDestroy, "ncclCommDestroy")
  FR_SYM(AllReduce, "ncclAllReduce")
  FR_SYM(Broadcast, "ncclBroadcast")
  FR_SYM(GetErrorString, "ncclGetErrorString")
  FR_SYM(CommCount, "ncclCommCount")
  FR_SYM(CommUserRank, "ncclCommUserRank")
  FR_SYM(CommCuDevice, "ncclCommCuDevice")
#undef FR_SYM
  g_rccl.so = so;
  return true;
}

#define FR_RCCL(expr)                                                                              \
  do {                                                                                             \
    ncclResult_t r_ = (expr);                                                                      \
    if (r_ != ncclSuccess) {                                                                       \
      frcnn::set_error("%s failed: %s (%s:%d)", #expr, g_rccl.GetErrorString(r_), __FILE__, __LINE__); \
      return FRCNN_ERR_HIP;                                                                        \
    }                                                                                              \
  } while (0)

double now_ms() {
  timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6;
}

}  // namespace

struct frcnn_comm {
  ncclComm_t comm;
  int nranks, rank, device;
};

static_assert(sizeof(ncclUniqueId) == FRCNN_COMM_ID_BYTES, "FRCNN_COMM_ID_BYTES must match ncclUniqueId");

extern "C" {

int frcnn_comm_get_unique_id(void* id_host) {
  FR_CHECK(id_host, "frcnn_comm_get_unique_id: NULL id");
  if (!rccl_load()) { frcnn::set_error("cannot load librccl: %s", g_rccl.err.c_str()); return FRCNN_ERR_STATE; }
  ncclUniqueId id;
  FR_RCCL(g_rccl.GetUniqueId(&id));
  memcpy(id_host, &id, sizeof(id));
  return FRCNN_OK;
}

// The rendezvous file: {128-byte id, 8-byte job nonce, 8-byte pid of the writer}.  A reader accepts it only when the
// nonce is its own job's (FRCNN_COMM_NONCE) and the writer is still alive: what a crashed or killed job left behind
// under the same path is ignored (and removed by the next rank 0 before it writes).
struct IdFile {
  unsigned char id[FRCNN_COMM_ID_BYTES];
  unsigned long long nonce;
  long long pid;
};

static unsigned long long job_nonce() {
  const char* e = getenv("FRCNN_COMM_NONCE");
  unsigned long long h = 1469598103934665603ull;   // FNV-1a of the string
  for (const char* p = (e && *e) ? e : "0"; *p; ++p) { h ^= (unsigned char)*p; h *= 1099511628211ull; }
  return h;
}

int frcnn_comm_exchange_id_file(const char* path, int rank, void* id_host, int timeout_ms) {
  FR_CHECK(path && *path && id_host, "frcnn_comm_exchange_id_file: path and id are required");
  const unsigned long long nonce = job_nonce();
  if (rank == 0) {
    unlink(path);   // a previous job's id (crashed before its destroy) must not be joined
    // written under a temporary name and renamed: a reader never sees a partial record
    std::string tmp = std::string(path) + ".tmp";
    int fd = open(tmp.c_str(), O_WRONLY | O_CREAT | O_TRUNC, 0600);
    FR_CHECK(fd >= 0, "frcnn_comm_exchange_id_file: cannot create %s", tmp.c_str());
    IdFile rec;
    memcpy(rec.id, id_host, FRCNN_COMM_ID_BYTES);
    rec.nonce = nonce; rec.pid = (long long)getpid();
    ssize_t w = write(fd, &rec, sizeof(rec));
    close(fd);
    FR_CHECK(w == (ssize_t)sizeof(rec), "frcnn_comm_exchange_id_file: short write to %s", tmp.c_str());
    FR_CHECK(rename(tmp.c_str(), path) == 0, "frcnn_comm_exchange_id_file: cannot rename %s", tmp.c_str());
    return FRCNN_OK;
  }
  const double t0 = now_ms();
  const char* why = "no file";
  for (;;) {
    int fd = open(path, O_RDONLY);
    if (fd >= 0) {
      IdFile rec;
      ssize_t r = read(fd, &rec, sizeof(rec));
      close(fd);
      if (r != (ssize_t)sizeof(rec)) why = "short or foreign file";
      else if (rec.nonce != nonce) why = "another job's nonce (stale file?)";
      else if (kill((pid_t)rec.pid, 0) != 0 && errno == ESRCH) why = "its writer is gone (stale file of a dead job)";
      else { memcpy(id_host, rec.id, FRCNN_COMM_ID_BYTES); return FRCNN_OK; }
    }
    if (now_ms() - t0 > timeout_ms) {
      frcnn::set_error("frcnn_comm_exchange_id_file: rank %d waited %d ms for %s (%s)", rank, timeout_ms, path, why);
      return FRCNN_ERR_STATE;
    }
    usleep(2000);
  }
}

int frcnn_comm_init_rank(frcnn_comm** out_host, int nranks, int rank, const void* id_host) {
  FR_CHECK(out_host && id_host && nranks >= 1 && rank >= 0 && rank < nranks, "frcnn_comm_init_rank: bad arguments");
  if (!rccl_load()) { frcnn::set_error("cannot load librccl: %s", g_rccl.err.c_str()); return FRCNN_ERR_STATE; }
  ncclUniqueId id;
  memcpy(&id, id_host, sizeof(id));
  frcnn_comm* c = new frcnn_comm();
  c->nranks = nranks; c->rank = rank;
  if (hipGetDevice(&c->device) != hipSuccess) { delete c; frcnn::set_error("frcnn_comm_init_rank: no HIP device"); return FRCNN_ERR_HIP; }
  ncclResult_t r = g_rccl.CommInitRank(&c->comm, nranks, id, rank);   // collective: every rank of the job calls it
  if (r != ncclSuccess) {
    frcnn::set_error("ncclCommInitRank(rank %d of %d) failed: %s", rank, nranks, g_rccl.GetErrorString(r));
    delete c;
    return FRCNN_ERR_HIP;
  }
  *out_host = c;
  return FRCNN_OK;
}

int frcnn_comm_init_rank_file(frcnn_comm** out_host, int nranks, int rank, const char* path, int timeout_ms) {
  unsigned char id[FRCNN_COMM_ID_BYTES];
  if (rank == 0) FR_TRY(frcnn_comm_get_unique_id(id));
  FR_TRY(frcnn_comm_exchange_id_file(path, rank, id, timeout_ms));
  return frcnn_comm_init_rank(out_host, nranks, rank, id);
}

int frcnn_comm_destroy(frcnn_comm* c) {
  if (!c) return FRCNN_OK;
  ncclResult_t r = g_rccl.CommDestroy(c->comm);
  delete c;
  if (r != ncclSuccess) { frcnn::set_error("ncclCommDestroy failed: %s", g_rccl.GetErrorString(r)); return FRCNN_ERR_HIP; }
  return FRCNN_OK;
}

int frcnn_comm_info(const frcnn_comm* c, int* nranks_host, int* rank_host) {
  FR_CHECK(c, "frcnn_comm_info: NULL communicator");
  if (nranks_host) *nranks_host = c->nranks;
  if (rank_host) *rank_host = c->rank;
  return FRCNN_OK;
}

int frcnn_comm_query(const frcnn_comm* c, int* count_host, int* user_rank_host, int* device_host) {
  FR_CHECK(c, "frcnn_comm_query: NULL communicator");
  if (count_host) FR_RCCL(g_rccl.CommCount(c->comm, count_host));
  if (user_rank_host) FR_RCCL(g_rccl.CommUserRank(c->comm, user_rank_host));
  if (device_host) FR_RCCL(g_rccl.CommCuDevice(c->comm, device_host));
  return FRCNN_OK;
}

int frcnn_allreduce_f32(frcnn_comm* c, float* buf, long long n, void* stream) {
  FR_CHECK(c && (buf || n == 0) && n >= 0, "frcnn_allreduce_f32: bad arguments");
  if (n == 0) return FRCNN_OK;
  FR_RCCL(g_rccl.AllReduce(buf, buf, (size_t)n, ncclFloat32, ncclSum, c->comm, frcnn::S(stream)));
  return FRCNN_OK;
}

int frcnn_allreduce_f64(frcnn_comm* c, double* buf, long long n, void* stream) {
  FR_CHECK(c && (buf || n == 0) && n >= 0, "frcnn_allreduce_f64: bad arguments");
  if (n == 0) return FRCNN_OK;
  FR_RCCL(g_rccl.AllReduce(buf, buf, (size_t)n, ncclFloat64, ncclSum, c->comm, frcnn::S(stream)));
  return FRCNN_OK;
}

int frcnn_broadcast_f32(frcnn_comm* c, float* buf, long long n, int root, void* stream) {
  FR_CHECK(c && (buf || n == 0) && n >= 0 && root >= 0 && root < c->nranks, "frcnn_broadcast_f32: bad arguments");
  if (n == 0) return FRCNN_OK;
  FR_RCCL(g_rccl.Broadcast(buf, buf, (size_t)n, ncclFloat32, root, c->comm, frcnn::S(stream)));
  return FRCNN_OK;
}

}  // extern "C"
