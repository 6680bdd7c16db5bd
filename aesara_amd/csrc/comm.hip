// RCCL communicator of the C-ABI (SURVEY §8b ahip_comm_init / ahip_allreduce_sum): the ONE real
// exchange step of the hot path — a CAReduce (or contraction) over a batch axis that has been
// split over the GPUs of a node — enqueued by the shim on the LAUNCH stream, so that an exchange
// round is a launch-list entry next to the kernels around it, not a Python call on another stream.
// One process per GPU; ranks agree on a 128-byte unique id (rank 0 creates it, the host side
// hands it to the others by whatever bootstrap it has: torch.distributed, a file, MPI).
// RCCL is loaded with dlopen at first use — libaesara_hip.so has no link-time dependency on it,
// and a process that already holds an RCCL (PyTorch bundles one) shares that copy.
#include <dlfcn.h>

#include <mutex>
#include <string>

#include "common.h"

namespace {
typedef struct { char internal[128]; } nccl_uid;
typedef void* nccl_comm;
typedef int (*fn_get_uid)(nccl_uid*);
typedef int (*fn_init_rank)(nccl_comm*, int, nccl_uid, int);
typedef int (*fn_allreduce)(const void*, void*, size_t, int, int, nccl_comm, hipStream_t);
typedef int (*fn_destroy)(nccl_comm);
typedef int (*fn_abort)(nccl_comm);
typedef const char* (*fn_errstr)(int);

struct Rccl {
  void* h = nullptr;
  fn_get_uid get_uid = nullptr;
  fn_init_rank init_rank = nullptr;
  fn_allreduce allreduce = nullptr;
  fn_destroy destroy = nullptr;
  fn_abort abort = nullptr;
  fn_errstr errstr = nullptr;
};
Rccl g_rccl;
std::mutex g_mu;
std::string g_path;

int load_rccl() {
  std::lock_guard<std::mutex> lk(g_mu);
  if (g_rccl.h) return AHIP_OK;
  const char* env = getenv("AESARA_HIP_RCCL");
  const char* cands[] = {g_path.empty() ? nullptr : g_path.c_str(), env, "librccl.so.1", "librccl.so",
                         "/opt/rocm/lib/librccl.so.1"};
  void* h = nullptr;
  // an RCCL that is already in the process first (never two copies with two sets of state)
  for (const char* c : {"librccl.so.1", "librccl.so"}) {
    h = dlopen(c, RTLD_NOW | RTLD_NOLOAD);
    if (h) break;
  }
  for (const char* c : cands) {
    if (h) break;
    if (c && *c) h = dlopen(c, RTLD_NOW | RTLD_LOCAL);
  }
  if (!h) {
    ahip_set_error("RCCL not found (librccl.so.1; set AESARA_HIP_RCCL): %s", dlerror());
    return AHIP_EINVAL;
  }
  Rccl r;
  r.h = h;
  r.get_uid = (fn_get_uid)dlsym(h, "ncclGetUniqueId");
  r.init_rank = (fn_init_rank)dlsym(h, "ncclCommInitRank");
  r.allreduce = (fn_allreduce)dlsym(h, "ncclAllReduce");
  r.destroy = (fn_destroy)dlsym(h, "ncclCommDestroy");
  r.abort = (fn_abort)dlsym(h, "ncclCommAbort");      // optional
  r.errstr = (fn_errstr)dlsym(h, "ncclGetErrorString");
  if (!r.get_uid || !r.init_rank || !r.allreduce || !r.destroy) {
    ahip_set_error("RCCL library lacks ncclGetUniqueId / ncclCommInitRank / ncclAllReduce");
    return AHIP_EINVAL;
  }
  g_rccl = r;
  return AHIP_OK;
}

int nccl_dtype(int dt) {   // ncclDataType_t values (rccl.h:459-472)
  switch (dt) {
    case AHIP_I8: return 0;
    case AHIP_U8: case AHIP_BOOL: return 1;
    case AHIP_I32: return 2;
    case AHIP_U32: return 3;
    case AHIP_I64: return 4;
    case AHIP_U64: return 5;
    case AHIP_F32: return 7;
    case AHIP_F64: return 8;
    default: return -1;
  }
}
}  // namespace

struct ahip_comm_s { nccl_comm comm; int nranks, rank; };

bool ahip_list_recording();
int ahip_list_record_allreduce(ahip_comm_s* c, int dtype, int op, const void* send, void* recv,
                               int64_t count);

#define AHIP_CHECK_NCCL(expr)                                                               \
  do {                                                                                      \
    int _r = (expr);                                                                        \
    if (_r != 0) {                                                                          \
      ahip_set_error("%s failed: %s", #expr, g_rccl.errstr ? g_rccl.errstr(_r) : "rccl error"); \
      return AHIP_EHIP;                                                                     \
    }                                                                                       \
  } while (0)

int ahip_comm_issue(ahip_comm_s* c, int dtype, int op, const void* send, void* recv, int64_t count,
                    hipStream_t s) {
  AHIP_CHECK_NCCL(g_rccl.allreduce(send, recv, (size_t)count, nccl_dtype(dtype), op, c->comm, s));
  return AHIP_OK;
}

extern "C" {

int ahip_comm_set_library(const char* path) {
  std::lock_guard<std::mutex> lk(g_mu);
  g_path = path ? path : "";
  return AHIP_OK;
}

int ahip_comm_unique_id(void* id_out, size_t id_bytes) {
  AHIP_REQUIRE(id_out && id_bytes >= AHIP_COMM_ID_BYTES, "id buffer must hold %d bytes", AHIP_COMM_ID_BYTES);
  int rc = load_rccl();
  if (rc) return rc;
  nccl_uid u;
  AHIP_CHECK_NCCL(g_rccl.get_uid(&u));
  memcpy(id_out, &u, sizeof(u));
  return AHIP_OK;
}

int ahip_comm_init_rank(const void* id, int nranks, int rank, ahip_comm_t* out) {
  AHIP_REQUIRE(id && out, "null argument");
  AHIP_REQUIRE(nranks >= 1 && rank >= 0 && rank < nranks, "rank %d outside [0, %d)", rank, nranks);
  int rc = load_rccl();
  if (rc) return rc;
  nccl_uid u;
  memcpy(&u, id, sizeof(u));
  nccl_comm c = nullptr;
  AHIP_CHECK_NCCL(g_rccl.init_rank(&c, nranks, u, rank));
  *out = new ahip_comm_s{c, nranks, rank};
  return AHIP_OK;
}

int ahip_comm_size(ahip_comm_t c) { return c ? c->nranks : -1; }
int ahip_comm_rank(ahip_comm_t c) { return c ? c->rank : -1; }

int ahip_allreduce(ahip_comm_t c, int dtype, int op, const void* sendbuf, void* recvbuf,
                   int64_t count, void* stream) {
  AHIP_REQUIRE(c != nullptr, "null communicator");
  AHIP_REQUIRE(nccl_dtype(dtype) >= 0, "dtype %d has no RCCL type", dtype);
  AHIP_REQUIRE(op == AHIP_RED_SUM || op == AHIP_RED_PROD || op == AHIP_RED_MAX || op == AHIP_RED_MIN,
               "bad reduction op %d", op);
  AHIP_REQUIRE(count >= 0, "negative count");
  if (count == 0) return AHIP_OK;
  AHIP_REQUIRE(sendbuf && recvbuf, "null buffer");
  if (ahip_list_recording()) return ahip_list_record_allreduce(c, dtype, op, sendbuf, recvbuf, count);
  return ahip_comm_issue(c, dtype, op, sendbuf, recvbuf, count, as_stream(stream));
}

int ahip_comm_abort(ahip_comm_t c) {
  // ncclCommAbort: frees the communicator AND terminates collectives that are stuck on the device
  // (a peer that never arrived); ncclCommDestroy would wait for them
  if (!c) return AHIP_OK;
  if (g_rccl.abort) (void)g_rccl.abort(c->comm);
  else if (g_rccl.destroy) (void)g_rccl.destroy(c->comm);
  delete c;
  return AHIP_OK;
}

int ahip_comm_destroy(ahip_comm_t c) {
  if (!c) return AHIP_OK;
  if (g_rccl.destroy) (void)g_rccl.destroy(c->comm);
  delete c;
  return AHIP_OK;
}

}  // extern "C"
