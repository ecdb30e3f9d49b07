// Data-parallel gradient exchange on RCCL behind the C ABI (SURVEY.md section 8b's designed exports: the multi-GPU
// surface of the library).  The reference trains under accelerate / torch DDP (Examples/vyom-ai-decoder_clm.ipynb cell 31,
// Examples/vyom-ai-accelerate-multimodel-2t4.ipynb cells 1-2): one process per GPU, gradients averaged by an all-reduce
// while backward is still running.  Here: one communicator per process, the bucket all-reduce enqueued on a caller-given
// HIP stream (the caller orders it against its compute and optimizer streams with events), in place, sum.
//
// RCCL is bound at RUN time (dlsym): a process that already holds an RCCL (torch.distributed's, loaded with the torch
// package) uses that very library -- two RCCL copies in one address space are trouble -- otherwise librccl.so is opened
// from the loader path / /opt/rocm/lib.  No link-time dependency: the library loads where RCCL is absent, and the
// entry points then fail with VY_ERR_UNSUPPORTED.
#include "vy_common.h"
#include <dlfcn.h>
#include <string.h>
#include <mutex>

namespace {
// the few RCCL declarations used (rccl.h: ncclUniqueId is 128 opaque bytes; enums as of NCCL 2.x)
struct NcclUniqueId { char internal[128]; };
typedef void* NcclComm;
enum { kNcclSuccess = 0 };
enum { kNcclFloat32 = 7, kNcclBfloat16 = 9 };
enum { kNcclSum = 0 };
typedef int (*GetUniqueIdFn)(NcclUniqueId*);
typedef int (*CommInitRankFn)(NcclComm*, int, NcclUniqueId, int);
typedef int (*AllReduceFn)(const void*, void*, size_t, int, int, NcclComm, hipStream_t);
typedef int (*CommDestroyFn)(NcclComm);
typedef const char* (*GetErrorStringFn)(int);

struct Rccl {
  GetUniqueIdFn get_unique_id = nullptr;
  CommInitRankFn comm_init_rank = nullptr;
  AllReduceFn all_reduce = nullptr;
  CommDestroyFn comm_destroy = nullptr;
  GetErrorStringFn error_string = nullptr;
  bool ok = false;
};
Rccl g_rccl;
std::once_flag g_rccl_once;
NcclComm g_comm = nullptr;
int g_world = 0, g_rank = -1;

void bind_rccl() {
  void* h = RTLD_DEFAULT;
  if (!dlsym(h, "ncclAllReduce")) {
    h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
    if (!h) h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!h) h = dlopen("/opt/rocm/lib/librccl.so", RTLD_NOW | RTLD_GLOBAL);
    if (!h) return;
  }
  g_rccl.get_unique_id = (GetUniqueIdFn)dlsym(h, "ncclGetUniqueId");
  g_rccl.comm_init_rank = (CommInitRankFn)dlsym(h, "ncclCommInitRank");
  g_rccl.all_reduce = (AllReduceFn)dlsym(h, "ncclAllReduce");
  g_rccl.comm_destroy = (CommDestroyFn)dlsym(h, "ncclCommDestroy");
  g_rccl.error_string = (GetErrorStringFn)dlsym(h, "ncclGetErrorString");
  g_rccl.ok = g_rccl.get_unique_id && g_rccl.comm_init_rank && g_rccl.all_reduce && g_rccl.comm_destroy;
}
const Rccl* rccl() {
  std::call_once(g_rccl_once, bind_rccl);
  return g_rccl.ok ? &g_rccl : nullptr;
}
const char* rccl_err(const Rccl* r, int rc) { return r->error_string ? r->error_string(rc) : "RCCL error"; }
}  // namespace

extern "C" int vy_ddp_unique_id(void* id128) {
  const Rccl* r = rccl();
  if (!r) VY_FAIL(VY_ERR_UNSUPPORTED, "vy_ddp_unique_id: RCCL is not available in this process");
  if (!id128) VY_FAIL(VY_ERR_ARG, "vy_ddp_unique_id: NULL");
  const int rc = r->get_unique_id(reinterpret_cast<NcclUniqueId*>(id128));
  if (rc != kNcclSuccess) VY_FAIL(VY_ERR_LAUNCH, "vy_ddp_unique_id: %s", rccl_err(r, rc));
  return VY_OK;
}

extern "C" int vy_ddp_init(const void* id128, int rank, int world) {
  const Rccl* r = rccl();
  if (!r) VY_FAIL(VY_ERR_UNSUPPORTED, "vy_ddp_init: RCCL is not available in this process");
  if (!id128 || world < 1 || rank < 0 || rank >= world) VY_FAIL(VY_ERR_ARG, "vy_ddp_init: bad arguments");
  if (g_comm) VY_FAIL(VY_ERR_ARG, "vy_ddp_init: a communicator exists already (vy_ddp_destroy first)");
  NcclUniqueId id;
  memcpy(&id, id128, sizeof(id));
  const int rc = r->comm_init_rank(&g_comm, world, id, rank);   // on the calling thread's current device; collective
  if (rc != kNcclSuccess) { g_comm = nullptr; VY_FAIL(VY_ERR_LAUNCH, "vy_ddp_init: %s", rccl_err(r, rc)); }
  g_world = world; g_rank = rank;
  return VY_OK;
}

extern "C" int vy_ddp_world(void) { return g_comm ? g_world : 0; }
extern "C" int vy_ddp_rank(void) { return g_comm ? g_rank : -1; }

extern "C" int vy_ddp_all_reduce_async(void* buf, int64_t count, int dtype, void* stream) {
  const Rccl* r = rccl();
  if (!r || !g_comm) VY_FAIL(VY_ERR_ARG, "vy_ddp_all_reduce_async: no communicator (vy_ddp_init)");
  if (!buf || count <= 0) VY_FAIL(VY_ERR_ARG, "vy_ddp_all_reduce_async: bad arguments");
  const int dt = dtype == VY_F32 ? kNcclFloat32 : (dtype == VY_BF16 ? kNcclBfloat16 : -1);
  if (dt < 0) VY_FAIL(VY_ERR_ARG, "vy_ddp_all_reduce_async: bad dtype %d", dtype);
  const int rc = r->all_reduce(buf, buf, (size_t)count, dt, kNcclSum, g_comm, (hipStream_t)stream);
  if (rc != kNcclSuccess) VY_FAIL(VY_ERR_LAUNCH, "vy_ddp_all_reduce_async: %s", rccl_err(r, rc));
  return VY_OK;
}

extern "C" int vy_ddp_destroy(void) {
  const Rccl* r = rccl();
  if (r && g_comm) r->comm_destroy(g_comm);
  g_comm = nullptr; g_world = 0; g_rank = -1;
  return VY_OK;
}
