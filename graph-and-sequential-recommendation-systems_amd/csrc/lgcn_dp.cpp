// lgcn_dp.cpp -- RCCL communicator of the data-parallel training path (include/lgcn_hip.h,
// "Data parallel").  The reference has no distributed code; the contract is
// BASELINE.json's north_star: one process per GPU, replicated tables, sharded BPR batches, gradient
// exchange over RCCL/xGMI.  RCCL is resolved at run time (dlopen), so the library loads -- and the
// single-GPU path runs -- on a machine without it.
#include <dlfcn.h>
#include <link.h>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>
#include <string>

#include "lgcn_hip.h"
#include "lgcn_internal.h"

namespace {

int find_loaded_rccl(struct dl_phdr_info *info, size_t, void *data) {
    const char *name = info->dlpi_name;
    if (name && std::strstr(name, "librccl.so")) { *static_cast<std::string *>(data) = name; return 1; }
    return 0;
}

RcclApi g_api;
bool g_ok = false;
std::once_flag g_once;

void resolve() {
    std::string path;
    dl_iterate_phdr(find_loaded_rccl, &path);
    void *h = nullptr;
    if (!path.empty()) h = dlopen(path.c_str(), RTLD_NOW | RTLD_NOLOAD | RTLD_LOCAL);
    if (!h && std::getenv("LGCN_RCCL_PATH")) h = dlopen(std::getenv("LGCN_RCCL_PATH"), RTLD_NOW | RTLD_LOCAL);
    if (!h) h = dlopen("librccl.so.1", RTLD_NOW | RTLD_LOCAL);
    if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_LOCAL);
    if (!h) h = dlopen("/opt/rocm/lib/librccl.so.1", RTLD_NOW | RTLD_LOCAL);
    if (!h) return;
#define SYM(field, name) \
    g_api.field = reinterpret_cast<decltype(g_api.field)>(dlsym(h, name)); \
    if (!g_api.field) return;
    SYM(GetUniqueId, "ncclGetUniqueId") SYM(CommInitRank, "ncclCommInitRank") SYM(CommDestroy, "ncclCommDestroy")
    SYM(AllGather, "ncclAllGather") SYM(AllReduce, "ncclAllReduce") SYM(Broadcast, "ncclBroadcast")
    SYM(GroupStart, "ncclGroupStart") SYM(GroupEnd, "ncclGroupEnd") SYM(GetErrorString, "ncclGetErrorString")
#undef SYM
    g_ok = true;
}

}  // namespace

const RcclApi *lgcn_rccl() {
    std::call_once(g_once, resolve);
    if (!g_ok) { lgcn_set_error("RCCL (librccl.so) not found: set LGCN_RCCL_PATH or install ROCm's rccl"); return nullptr; }
    return &g_api;
}

static int rccl_fail(const RcclApi *api, const char *what, ncclResult_t r) {
    std::string m = std::string(what) + " failed: " + (api->GetErrorString ? api->GetErrorString(r) : "?");
    lgcn_set_error(m.c_str());
    return 11;
}

extern "C" int lgcn_dp_available(void) { return lgcn_rccl() != nullptr; }

extern "C" int lgcn_dp_unique_id(void *id128) {
    if (!id128) { lgcn_set_error("lgcn_dp_unique_id: null argument"); return 3; }
    const RcclApi *api = lgcn_rccl();
    if (!api) return 12;
    ncclUniqueId id;
    static_assert(sizeof(id) == LGCN_DP_ID_BYTES, "ncclUniqueId size");
    ncclResult_t r = api->GetUniqueId(&id);
    if (r != ncclSuccess) return rccl_fail(api, "ncclGetUniqueId", r);
    std::memcpy(id128, &id, sizeof id);
    return 0;
}

extern "C" int lgcn_dp_init(const void *id128, int world, int rank, lgcn_dp **out) {
    if (!id128 || !out || world < 1 || rank < 0 || rank >= world) { lgcn_set_error("lgcn_dp_init: invalid argument"); return 3; }
    const RcclApi *api = lgcn_rccl();
    if (!api) return 12;
    lgcn_dp *dp = new (std::nothrow) lgcn_dp;
    if (!dp) { lgcn_set_error("out of memory"); return 4; }
    ncclUniqueId id;
    std::memcpy(&id, id128, sizeof id);
    ncclResult_t r = api->CommInitRank(&dp->comm, world, id, rank);     // on the calling thread's current HIP device
    if (r != ncclSuccess) { delete dp; return rccl_fail(api, "ncclCommInitRank", r); }
    dp->world = world; dp->rank = rank; dp->api = api; dp->loopback = false;
    *out = dp;
    return 0;
}

extern "C" void lgcn_dp_destroy(lgcn_dp *dp) {
    if (!dp) return;
    if (dp->loopback) lgcn_dp_loopback_release(dp);
    else if (dp->api) (void)dp->api->CommDestroy(dp->comm);
    delete dp;
}

extern "C" int lgcn_dp_allreduce_sum_f32(lgcn_dp *dp, float *buf, int64_t n, void *stream) {
    if (!dp || !dp->api || !buf || n <= 0) { lgcn_set_error("lgcn_dp_allreduce_sum_f32: invalid argument"); return 3; }
    ncclResult_t r = dp->api->AllReduce(buf, buf, (size_t)n, ncclFloat32, ncclSum, dp->comm, (hipStream_t)stream);
    if (r != ncclSuccess) { lgcn_set_error("ncclAllReduce failed"); return 11; }
    return 0;
}

extern "C" int lgcn_dp_world(const lgcn_dp *dp) { return dp ? dp->world : 0; }
extern "C" int lgcn_dp_rank(const lgcn_dp *dp) { return dp ? dp->rank : -1; }
