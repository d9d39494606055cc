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

#include <atomic>
#include <condition_variable>
#include <memory>
#include <mutex>
#include <thread>

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
  ncclResult_t (*CommGetAsyncError)(ncclComm_t, ncclResult_t*) = nullptr;   // optional (RCCL >= 2.14): see settle()
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
  *(void**)(&g_rccl.CommGetAsyncError) = dlsym(so, "ncclCommGetAsyncError");
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

// A collective on a communicator that was made non-blocking behind our back (NCCL_COMM_BLOCKING=0 in the environment)
// may answer ncclInProgress: the operation counts as enqueued only when the communicator's state has left that value.
ncclResult_t settle(ncclComm_t comm, ncclResult_t r) {
  while (r == ncclInProgress && g_rccl.CommGetAsyncError) {
    ncclResult_t st = ncclSuccess;
    ncclResult_t q = g_rccl.CommGetAsyncError(comm, &st);
    if (q != ncclSuccess) return q;
    r = st;
    if (r == ncclInProgress) usleep(50);
  }
  return r;
}

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

// The rendezvous file: {128-byte id, 8-byte job nonce, 8-byte pid of the writer, its host name}.  A reader accepts it only
// when the nonce is its own job's (FRCNN_COMM_NONCE) and -- when the writer ran on the reader's own host, where a pid can
// be probed -- the writer is still alive: what a crashed or killed job left behind under the same path is ignored (and
// removed by the next rank 0 before it writes).  A writer on ANOTHER host (the file on a shared file system, or another
// container's pid namespace under another host name) is judged by the nonce alone.
struct IdFile {
  unsigned char id[FRCNN_COMM_ID_BYTES];
  unsigned long long nonce;
  long long pid;
  char host[64];
};

static void this_host(char out[64]) {
  memset(out, 0, 64);
  if (const char* e = getenv("FRCNN_COMM_HOSTNAME")) { strncpy(out, e, 63); return; }   // (tests: pose as another host)
  if (gethostname(out, 63) != 0) out[0] = 0;
  // two containers may share a host name but not a pid namespace: the namespace's inode is part of the identity
  struct stat st;
  if (stat("/proc/self/ns/pid", &st) == 0) {
    size_t n = strlen(out);
    snprintf(out + n, 63 - n, "#%lx", (unsigned long)st.st_ino);
  }
}

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
    this_host(rec.host);
    ssize_t w = write(fd, &rec, sizeof(rec));
    close(fd);
    FR_CHECK(w == (ssize_t)sizeof(rec), "frcnn_comm_exchange_id_file: short write to %s", tmp.c_str());
    FR_CHECK(rename(tmp.c_str(), path) == 0, "frcnn_comm_exchange_id_file: cannot rename %s", tmp.c_str());
    return FRCNN_OK;
  }
  const double t0 = now_ms();
  const char* why = "no file";
  char me[64];
  this_host(me);
  for (;;) {
    int fd = open(path, O_RDONLY);
    if (fd >= 0) {
      IdFile rec;
      ssize_t r = read(fd, &rec, sizeof(rec));
      close(fd);
      if (r != (ssize_t)sizeof(rec)) why = "short or foreign file";
      else if (rec.nonce != nonce) why = "another job's nonce (stale file?)";
      else if (memcmp(rec.host, me, 64) == 0 && kill((pid_t)rec.pid, 0) != 0 && errno == ESRCH)
        why = "its writer is gone (stale file of a dead job)";
      else { memcpy(id_host, rec.id, FRCNN_COMM_ID_BYTES); return FRCNN_OK; }
    }
    if (now_ms() - t0 > timeout_ms) {
      frcnn::set_error("frcnn_comm_exchange_id_file: rank %d waited %d ms for %s (%s)", rank, timeout_ms, path, why);
      return FRCNN_ERR_STATE;
    }
    usleep(2000);
  }
}

// ncclCommInitRank is a collective: a rank whose peer died between the rendezvous and this call would wait forever.
// timeout_ms > 0 puts a watchdog on it: the BLOCKING call runs on a helper thread the caller waits for with a deadline; on
// expiry the thread is abandoned (it owns nothing the caller touches again) and the caller gets an error.  The communicator
// itself is always a blocking one -- the ordinary kind, whose collectives have been enqueued on the caller's stream when the
// call returns.  (Round 4 built it with ncclCommInitRankConfig(blocking = 0) so that an expired initialisation could be
// aborted; but such a communicator stays non-blocking for life: its collectives may return ncclInProgress and enqueue their
// kernel later from a helper thread -- after the events the caller records behind the call.  ADVICE r4, high.)
// FRCNN_COMM_CHANNELS=n caps the channels (= CUs) RCCL's kernels take: a throughput-bound training step shares the CUs
// with the all-reduce of the previous bucket (NCCL_MAX_NCHANNELS, read by RCCL when the communicator is built).
// FRCNN_COMM_FAULT=hang_init (tests): the initialisation never returns, as with a dead peer.
static int comm_init(frcnn_comm** out_host, int nranks, int rank, const void* id_host, int timeout_ms) {
  FR_CHECK(out_host && id_host && nranks >= 1 && rank >= 0 && rank < nranks, "frcnn_comm_init_rank: bad arguments");
  const char* fault = getenv("FRCNN_COMM_FAULT");
  const bool hang = fault && strcmp(fault, "hang_init") == 0;
  if (!hang && !rccl_load()) { frcnn::set_error("cannot load librccl: %s", g_rccl.err.c_str()); return FRCNN_ERR_STATE; }
  if (const char* ch = getenv("FRCNN_COMM_CHANNELS"))
    if (atoi(ch) > 0) setenv("NCCL_MAX_NCHANNELS", ch, 1);
  ncclUniqueId id;
  memcpy(&id, id_host, sizeof(id));
  int device = 0;
  if (!hang && hipGetDevice(&device) != hipSuccess) { frcnn::set_error("frcnn_comm_init_rank: no HIP device"); return FRCNN_ERR_HIP; }
  ncclComm_t comm = nullptr;
  ncclResult_t r = ncclSuccess;
  if (timeout_ms <= 0 && !hang) {
    r = g_rccl.CommInitRank(&comm, nranks, id, rank);   // collective: every rank of the job calls it
  } else {
    struct Job { std::mutex mu; std::condition_variable cv; bool done = false; ncclComm_t comm = nullptr; ncclResult_t r = ncclSuccess; };
    std::shared_ptr<Job> job = std::make_shared<Job>();
    std::thread([job, nranks, id, rank, device, hang]() {
      if (hang) for (;;) sleep(3600);
      (void)hipSetDevice(device);
      ncclComm_t c = nullptr;
      ncclResult_t rr = g_rccl.CommInitRank(&c, nranks, id, rank);
      std::lock_guard<std::mutex> lk(job->mu);
      job->comm = c; job->r = rr; job->done = true;
      job->cv.notify_all();
    }).detach();
    std::unique_lock<std::mutex> lk(job->mu);
    const bool in_time = job->cv.wait_for(lk, std::chrono::milliseconds(timeout_ms > 0 ? timeout_ms : 1000), [&] { return job->done; });
    if (!in_time) {
      frcnn::set_error("ncclCommInitRank(rank %d of %d) did not complete within %d ms: a peer is missing or dead "
                       "(the initialising thread was abandoned)", rank, nranks, timeout_ms > 0 ? timeout_ms : 1000);
      return FRCNN_ERR_STATE;
    }
    comm = job->comm; r = job->r;
  }
  if (r != ncclSuccess) {
    frcnn::set_error("ncclCommInitRank(rank %d of %d) failed: %s", rank, nranks, g_rccl.GetErrorString(r));
    return FRCNN_ERR_HIP;
  }
  frcnn_comm* c = new frcnn_comm();
  c->nranks = nranks; c->rank = rank; c->device = device; c->comm = comm;
  *out_host = c;
  return FRCNN_OK;
}

int frcnn_comm_init_rank(frcnn_comm** out_host, int nranks, int rank, const void* id_host) {
  return comm_init(out_host, nranks, rank, id_host, 0);
}

int frcnn_comm_init_rank_timeout(frcnn_comm** out_host, int nranks, int rank, const void* id_host, int timeout_ms) {
  return comm_init(out_host, nranks, rank, id_host, timeout_ms);
}

// timeout_ms bounds the wait for the id file AND (what is left of it, at least a second) the collective initialisation
int frcnn_comm_init_rank_file(frcnn_comm** out_host, int nranks, int rank, const char* path, int timeout_ms) {
  unsigned char id[FRCNN_COMM_ID_BYTES];
  const bool hang = getenv("FRCNN_COMM_FAULT") && strcmp(getenv("FRCNN_COMM_FAULT"), "hang_init") == 0;
  if (rank == 0) {
    if (hang) memset(id, 0, sizeof(id));
    else FR_TRY(frcnn_comm_get_unique_id(id));
  }
  const double t0 = now_ms();
  FR_TRY(frcnn_comm_exchange_id_file(path, rank, id, timeout_ms));
  const int left = timeout_ms > 0 ? std::max(1000, timeout_ms - (int)(now_ms() - t0)) : 0;
  return comm_init(out_host, nranks, rank, id, left);
}

int frcnn_comm_destroy(frcnn_comm* c) {
  if (!c) return FRCNN_OK;
  ncclResult_t r = g_rccl.CommDestroy(c->comm);
  if (r == ncclInProgress) r = ncclSuccess;   // (a non-blocking communicator finishes its teardown in the background)
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
  FR_RCCL(settle(c->comm, g_rccl.AllReduce(buf, buf, (size_t)n, ncclFloat32, ncclSum, c->comm, frcnn::S(stream))));
  return FRCNN_OK;
}

int frcnn_allreduce_f64(frcnn_comm* c, double* buf, long long n, void* stream) {
  FR_CHECK(c && (buf || n == 0) && n >= 0, "frcnn_allreduce_f64: bad arguments");
  if (n == 0) return FRCNN_OK;
  FR_RCCL(settle(c->comm, g_rccl.AllReduce(buf, buf, (size_t)n, ncclFloat64, ncclSum, c->comm, frcnn::S(stream))));
  return FRCNN_OK;
}

int frcnn_broadcast_f32(frcnn_comm* c, float* buf, long long n, int root, void* stream) {
  FR_CHECK(c && (buf || n == 0) && n >= 0 && root >= 0 && root < c->nranks, "frcnn_broadcast_f32: bad arguments");
  if (n == 0) return FRCNN_OK;
  FR_RCCL(settle(c->comm, g_rccl.Broadcast(buf, buf, (size_t)n, ncclFloat32, root, c->comm, frcnn::S(stream))));
  return FRCNN_OK;
}

}  // extern "C"
