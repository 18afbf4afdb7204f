// lgcn_internal.h -- declarations shared by the translation units of liblgcn_hip.so (not installed).
#ifndef LGCN_INTERNAL_H
#define LGCN_INTERNAL_H
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>       // types only: RCCL is resolved with dlopen at run time, never linked

extern "C" void lgcn_set_error(const char *msg);   // lgcn_host.cpp

// The RCCL entry points the data-parallel path uses.  Resolved once, preferring the librccl that is
// already mapped into the process (PyTorch-ROCm ships its own copy; two RCCL instances in one process
// would each build their own topology/IPC state), else LGCN_RCCL_PATH, else the system librccl.so.1.
struct RcclApi {
    ncclResult_t (*GetUniqueId)(ncclUniqueId *);
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int);
    ncclResult_t (*CommDestroy)(ncclComm_t);
    ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t);
    ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t);
    ncclResult_t (*Broadcast)(const void *, void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t);
    ncclResult_t (*GroupStart)();
    ncclResult_t (*GroupEnd)();
    const char *(*GetErrorString)(ncclResult_t);
};
const RcclApi *lgcn_rccl();      // nullptr (+ error text) if RCCL cannot be found

struct lgcn_dp {
    ncclComm_t comm;
    int world, rank;
    const RcclApi *api;      // the collectives this communicator runs on: RCCL (lgcn_dp.cpp) or the in-process loopback
    bool loopback;
};
void lgcn_dp_loopback_release(lgcn_dp *dp);      // lgcn_dp_loopback.hip
void lgcn_dp_loopback_abort(lgcn_dp *dp);        // a rank failed outside a collective: wake every waiter, all later collectives fail
#endif
