// lgcn_dp_loopback.hip -- a communicator for W ranks that are THREADS OF ONE PROCESS on one GPU (test hook; the
// product communicator is RCCL, lgcn_dp.cpp).  It implements the five collectives lgcn_train_epoch_dp issues
// (AllGather, AllReduce SUM, grouped in-place Broadcast) with a host rendezvous and hipMemcpyAsync / one small
// reduction kernel, so that the C loop of a data-parallel epoch -- its world > 1 control flow, ragged and empty
// shards, rs_exchange's op order, LocalScope -- runs for real on a one-GPU box, every rank with its own context,
// tables and stream.  It is NOT asynchronous (every collective synchronises the rank's stream and meets the other
// ranks on the host): it exists to be bitwise comparable with the single-GPU epoch, not to be fast.
#include <hip/hip_runtime.h>

#include <condition_variable>
#include <cstring>
#include <mutex>
#include <new>
#include <vector>

#include "lgcn_hip.h"
#include "lgcn_internal.h"

namespace {

struct Op { const void *send; void *recv; size_t bytes; int root; };

struct Group {
    int world;
    std::mutex mu; std::condition_variable cv;
    int arrived = 0; long generation = 0; bool failed = false, aborted = false;
    std::vector<std::vector<Op>> ops;          // per rank: the ops of the open group (or the single op of a collective)
    std::vector<void *> scratch; std::vector<size_t> scratch_bytes;     // per rank: all-reduce staging
    int refs;
    explicit Group(int w) : world(w), ops((size_t)w), scratch((size_t)w, nullptr), scratch_bytes((size_t)w, 0), refs(w) {}
    // host barrier of the world's threads; any rank may report a failure, all see it
    bool barrier(bool ok = true) {
        std::unique_lock<std::mutex> lk(mu);
        if (!ok) failed = true;
        if (aborted) return false;                 // a rank has left the epoch: nobody waits for it again
        const long gen = generation;
        if (++arrived == world) { arrived = 0; generation++; cv.notify_all(); }
        else cv.wait(lk, [&] { return generation != gen || aborted; });
        return !failed;
    }
    void abort() {
        std::lock_guard<std::mutex> lk(mu);
        failed = aborted = true;
        cv.notify_all();
    }
};

struct Loop { Group *g; int rank; bool in_group; };

inline Loop *loop_of(ncclComm_t c) { return reinterpret_cast<Loop *>(c); }

size_t type_bytes(ncclDataType_t t) {
    switch (t) {
    case ncclInt64: case ncclUint64: case ncclFloat64: return 8;
    case ncclFloat32: case ncclInt32: case ncclUint32: return 4;
    case ncclBfloat16: case ncclFloat16: return 2;
    default: return 1;
    }
}

template <typename T>
__global__ void __launch_bounds__(256) k_sum_ranks(T *dst, const T *const *src, int world, size_t n) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        T s = src[0][i];
        for (int q = 1; q < world; q++) s += src[q][i];          // fixed rank order
        dst[i] = s;
    }
}

// run the recorded ops of every rank: rank r copies, for each op i, the root's send range into its own recv range
ncclResult_t run_ops(Loop *l, hipStream_t st) {
    Group *g = l->g;
    bool ok = hipStreamSynchronize(st) == hipSuccess;             // my buffers are final
    if (!g->barrier(ok)) return ncclUnhandledCudaError;           // ... and so are everybody's; op lists are published
    const std::vector<Op> &mine = g->ops[(size_t)l->rank];
    for (size_t i = 0; ok && i < mine.size(); i++) {
        const int root = mine[i].root;
        if (root == l->rank) continue;
        const std::vector<Op> &theirs = g->ops[(size_t)root];
        if (i >= theirs.size() || theirs[i].bytes != mine[i].bytes || theirs[i].root != root) { ok = false; break; }   // ranks disagree on the sequence
        ok = hipMemcpyAsync(mine[i].recv, theirs[i].send, mine[i].bytes, hipMemcpyDeviceToDevice, st) == hipSuccess;
    }
    ok = ok && hipStreamSynchronize(st) == hipSuccess;
    const bool all_ok = g->barrier(ok);                           // nobody's send buffer is reused before everyone has read it
    g->ops[(size_t)l->rank].clear();
    return all_ok ? ncclSuccess : ncclUnhandledCudaError;
}

ncclResult_t lb_group_start() { return ncclSuccess; }             // (state lives in the communicator: see lb_broadcast)

ncclResult_t lb_all_gather(const void *send, void *recv, size_t count, ncclDataType_t t, ncclComm_t c, hipStream_t st) {
    Loop *l = loop_of(c); Group *g = l->g;
    const size_t bytes = count * type_bytes(t);
    std::vector<Op> &mine = g->ops[(size_t)l->rank];
    mine.clear();
    for (int q = 0; q < g->world; q++) mine.push_back(Op{send, (char *)recv + (size_t)q * bytes, bytes, q});
    // op q of rank r: "copy rank q's send block into my slot q"; my own block too
    bool ok = hipMemcpyAsync((char *)recv + (size_t)l->rank * bytes, send, bytes, hipMemcpyDeviceToDevice, st) == hipSuccess;
    if (!ok) { (void)g->barrier(false); (void)g->barrier(false); mine.clear(); return ncclUnhandledCudaError; }
    return run_ops(l, st);
}

ncclResult_t lb_broadcast(const void *send, void *recv, size_t count, ncclDataType_t t, int root, ncclComm_t c, hipStream_t) {
    Loop *l = loop_of(c);
    l->g->ops[(size_t)l->rank].push_back(Op{send, recv, count * type_bytes(t), root});     // executed by the group's end
    return ncclSuccess;
}

ncclResult_t lb_all_reduce(const void *send, void *recv, size_t count, ncclDataType_t t, ncclRedOp_t op, ncclComm_t c, hipStream_t st) {
    Loop *l = loop_of(c); Group *g = l->g;
    if (op != ncclSum || (t != ncclInt64 && t != ncclFloat32)) return ncclInvalidArgument;
    const size_t bytes = count * type_bytes(t);
    const size_t r = (size_t)l->rank;
    bool ok = true;
    if (g->scratch_bytes[r] < bytes + 64 * sizeof(void *)) {
        if (g->scratch[r]) (void)hipFree(g->scratch[r]);
        g->scratch[r] = nullptr; g->scratch_bytes[r] = 0;
        ok = hipMalloc(&g->scratch[r], bytes + 64 * sizeof(void *)) == hipSuccess;
        if (ok) g->scratch_bytes[r] = bytes + 64 * sizeof(void *);
    }
    // stage my contribution (the reduction may be in place), meet, sum every rank's staging buffer in rank order
    ok = ok && hipMemcpyAsync(g->scratch[r], send, bytes, hipMemcpyDeviceToDevice, st) == hipSuccess;
    ok = ok && hipStreamSynchronize(st) == hipSuccess;
    if (!g->barrier(ok)) { (void)g->barrier(false); return ncclUnhandledCudaError; }
    const void *ptrs[64];
    if (g->world > 64) ok = false;
    for (int q = 0; ok && q < g->world; q++) ptrs[q] = g->scratch[(size_t)q];
    void **dptrs = (void **)((char *)g->scratch[r] + ((bytes + 7) & ~(size_t)7));
    ok = ok && hipMemcpyAsync(dptrs, ptrs, sizeof(void *) * (size_t)g->world, hipMemcpyHostToDevice, st) == hipSuccess;
    if (ok) {
        const size_t blocks = (count + 255) / 256;
        const unsigned grid = (unsigned)(blocks < 4096 ? blocks : 4096);
        if (t == ncclInt64) hipLaunchKernelGGL(k_sum_ranks<long long>, dim3(grid), dim3(256), 0, st, (long long *)recv, (const long long *const *)dptrs, g->world, count);
        else hipLaunchKernelGGL(k_sum_ranks<float>, dim3(grid), dim3(256), 0, st, (float *)recv, (const float *const *)dptrs, g->world, count);
        ok = hipGetLastError() == hipSuccess && hipStreamSynchronize(st) == hipSuccess;
    }
    return g->barrier(ok) ? ncclSuccess : ncclUnhandledCudaError;
}

// the grouped broadcasts of rs_exchange run when the group closes; GroupEnd has no communicator argument, so the
// loop keeps the communicator of the calling thread here between GroupStart and GroupEnd
thread_local Loop *t_open = nullptr;
thread_local hipStream_t t_stream = nullptr;
ncclResult_t lb_broadcast_grouped(const void *send, void *recv, size_t count, ncclDataType_t t, int root, ncclComm_t c, hipStream_t st) {
    t_open = loop_of(c); t_stream = st;
    return lb_broadcast(send, recv, count, t, root, c, st);
}
ncclResult_t lb_group_end_run() {
    Loop *l = t_open;
    t_open = nullptr;
    if (!l) return ncclSuccess;                    // a group without ops on this rank cannot happen: the ranges table is global
    return run_ops(l, t_stream);
}

const char *lb_error_string(ncclResult_t) { return "loopback collective failed"; }

const RcclApi g_loop_api = {
    nullptr, nullptr, nullptr,
    lb_all_gather, lb_all_reduce, lb_broadcast_grouped, lb_group_start, lb_group_end_run, lb_error_string,
};

}  // namespace

extern "C" int lgcn_dp_init_loopback(int world, lgcn_dp **out) {
    if (!out || world < 1 || world > 64) { lgcn_set_error("lgcn_dp_init_loopback: invalid argument"); return 3; }
    Group *g = new (std::nothrow) Group(world);
    if (!g) { lgcn_set_error("out of memory"); return 4; }
    for (int r = 0; r < world; r++) {
        lgcn_dp *dp = new (std::nothrow) lgcn_dp;
        Loop *l = new (std::nothrow) Loop{g, r, false};
        if (!dp || !l) { lgcn_set_error("out of memory"); return 4; }
        dp->comm = reinterpret_cast<ncclComm_t>(l); dp->world = world; dp->rank = r; dp->api = &g_loop_api; dp->loopback = true;
        out[r] = dp;
    }
    return 0;
}

void lgcn_dp_loopback_abort(lgcn_dp *dp) { reinterpret_cast<Loop *>(dp->comm)->g->abort(); }

// called by lgcn_dp_destroy for a loopback communicator
void lgcn_dp_loopback_release(lgcn_dp *dp) {
    Loop *l = reinterpret_cast<Loop *>(dp->comm);
    Group *g = l->g;
    bool last;
    { std::lock_guard<std::mutex> lk(g->mu); last = --g->refs == 0; }
    if (last) {
        for (void *p : g->scratch) if (p) (void)hipFree(p);
        delete g;
    }
    delete l;
}
