// lgcn_sampler.hip -- bit-exact BPR triplet sampler on the GPU.
//
// Replaces sampling.sample_negative (sources/sampling.cpp:27-56, neg_num = 1) with the SAME output,
// int32 [user_num * (train_num / user_num), 3] rows (user, positive, negative) grouped by user, drawn
// from the SAME glibc rand() stream -- but produced on the device, so neither the serial host loop
// (20 ms per Gowalla epoch, tens of seconds at 200 M triplets) nor the upload of its output remain.
//
// The reference loop is serial because the number of draws a triplet consumes is data dependent:
//      pos = allPos[u][rand() % deg] ;  do neg = rand() % m while neg in allPos[u]
// Two facts make it parallel:
//   * glibc's TYPE_3 generator is the linear recurrence x[n] = x[n-3] + x[n-31] over Z/2^32, so the raw
//     stream can be entered anywhere: the host computes the start history of every block of 4 092 draws
//     with the 31 x 31 step matrix raised to the block length (lgcn_host.cpp), and k_glibc_expand fills
//     all blocks at once (one thread per block, the 31-word history in registers);
//   * rejections are rare (deg / m: 0.07 % of the triplets on Gowalla), and a triplet that starts at draw
//     2t + delta (delta = rejections so far) is a pure function of (t, delta).  k_sample evaluates a
//     window of triplets with the current delta in parallel, finds the FIRST one that rejects, commits
//     everything before it, resolves that triplet serially (it is the only one whose draw count is
//     unknown), bumps delta and continues behind it.  Iterations = rejections + T / window.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <vector>

#include "lgcn_hip.h"
#include "lgcn_internal.h"

extern "C" void lgcn_glibc_block_histories(int64_t nblocks, int64_t block_len, uint32_t *out);   // lgcn_host.cpp
extern "C" void lgcn_glibc_advance(uint64_t n);

#define GLIBC_BLOCK (31 * 132)        /* draws per block: a whole number of 31-step rounds */

// raw stream: R[b * GLIBC_BLOCK + i] = rand() value number i of block b
__global__ void __launch_bounds__(64) k_glibc_expand(const uint32_t *hist, int64_t nblocks, uint32_t *R) {
    const int64_t b = (int64_t)blockIdx.x * 64 + threadIdx.x;
    if (b >= nblocks) return;
    uint32_t h[31];
#pragma unroll
    for (int j = 0; j < 31; j++) h[j] = hist[b * 31 + j];
    uint32_t *out = R + b * GLIBC_BLOCK;
    for (int r = 0; r < GLIBC_BLOCK / 31; r++) {
#pragma unroll
        for (int j = 0; j < 31; j++) {                 // slot j holds the oldest value at step j of a round
            h[j] += h[(j + 28) % 31];
            out[r * 31 + j] = h[j] >> 1;
        }
    }
}

struct SampleArgs {
    const uint32_t *R; int64_t n_draws;
    const int64_t *indptr; const int32_t *indices;
    int32_t user_num, item_num, per_user;
    int64_t T;
    int32_t *S;
    int64_t *result;      // [0] = draws consumed, [1] = 0 ok / 1 stream exhausted
};

__device__ __forceinline__ bool is_positive(const int32_t *row, int deg, int item) {
    int lo = 0, hi = deg;
    while (lo < hi) { const int mid = (lo + hi) >> 1; if (row[mid] < item) lo = mid + 1; else hi = mid; }
    return lo < deg && row[lo] == item;
}

#define SAMPLE_THREADS 1024
__global__ void __launch_bounds__(SAMPLE_THREADS) k_sample(SampleArgs a) {
    __shared__ long long first_rej;      // smallest rejecting triplet of the window
    __shared__ long long s_t0, s_delta;
    __shared__ int s_fail;
    const int tid = threadIdx.x;
    if (tid == 0) { s_t0 = 0; s_delta = 0; s_fail = 0; }
    __syncthreads();
    long long window = 2 * SAMPLE_THREADS;
    while (true) {
        const long long t0 = s_t0, delta = s_delta;
        if (t0 >= a.T || s_fail) break;
        if (tid == 0) first_rej = a.T;
        __syncthreads();
        const long long t1 = min((long long)a.T, t0 + window);
        if (2 * t1 + delta + 2 > a.n_draws) { if (tid == 0) s_fail = 1; __syncthreads(); break; }
        for (long long t = t0 + tid; t < t1; t += SAMPLE_THREADS) {
            const int u = (int)(t / a.per_user);
            const int64_t rs = a.indptr[u];
            const int deg = (int)(a.indptr[u + 1] - rs);
            const int32_t *row = a.indices + rs;
            const long long base = 2 * t + delta;
            const int pos = row[(int)(a.R[base] % (uint32_t)deg)];
            const int cand = (int)(a.R[base + 1] % (uint32_t)a.item_num);
            if (is_positive(row, deg, cand)) atomicMin(&first_rej, t);
            else { a.S[3 * t] = u; a.S[3 * t + 1] = pos; a.S[3 * t + 2] = cand; }
        }
        __syncthreads();
        const long long fr = first_rej;
        if (tid == 0) {
            if (fr >= t1) { s_t0 = t1; }
            else {                              // triplet fr: walk its rejection loop, the only serial part
                const int u = (int)(fr / a.per_user);
                const int64_t rs = a.indptr[u];
                const int deg = (int)(a.indptr[u + 1] - rs);
                const int32_t *row = a.indices + rs;
                long long i = 2 * fr + delta;
                const int pos = row[(int)(a.R[i++] % (uint32_t)deg)];
                int cand;
                bool ok = true;
                do {
                    if (i >= a.n_draws) { ok = false; break; }
                    cand = (int)(a.R[i++] % (uint32_t)a.item_num);
                } while (is_positive(row, deg, cand));
                if (!ok) s_fail = 1;
                else {
                    a.S[3 * fr] = u; a.S[3 * fr + 1] = pos; a.S[3 * fr + 2] = cand;
                    s_delta = i - 2 * (fr + 1);
                    s_t0 = fr + 1;
                }
            }
        }
        // window follows the distance between rejections (2x the last gap, within [2, 64] x threads)
        const long long gap = (fr < t1 ? fr : t1) - t0 + 1;
        window = min(max(2 * gap, 2LL * SAMPLE_THREADS), 64LL * SAMPLE_THREADS);
        __syncthreads();
    }
    if (tid == 0) { a.result[0] = 2 * a.T + s_delta; a.result[1] = s_fail; }
}

extern "C" int64_t lgcn_sample_negative_device_workspace(int user_num, int64_t train_num) {
    if (user_num <= 0 || train_num < 0) return 0;
    const int64_t T = (int64_t)user_num * (train_num / user_num);
    const int64_t draws = 2 * T + T / 50 + 65536;                    // 2 % + slack for rejections
    const int64_t nblocks = (draws + GLIBC_BLOCK - 1) / GLIBC_BLOCK;
    return nblocks * GLIBC_BLOCK * 4 + nblocks * 31 * 4 + 256;
}

extern "C" int lgcn_sample_negative_device(int user_num, int item_num, int64_t train_num,
                                           const int64_t *h_indptr, const int64_t *d_indptr, const int32_t *d_indices,
                                           int32_t *d_S, void *workspace, int64_t workspace_bytes, void *stream) {
    if (user_num <= 0 || item_num <= 0 || train_num < 0 || !h_indptr || !d_indptr || !d_indices || !d_S || !workspace) {
        lgcn_set_error("sample_negative_device: invalid argument"); return 3;
    }
    const int per_user = (int)(train_num / user_num);
    const int64_t T = (int64_t)user_num * per_user;
    if (T == 0) return 0;
    for (int u = 0; u < user_num; u++) {
        const int64_t deg = h_indptr[u + 1] - h_indptr[u];
        if (deg <= 0) { lgcn_set_error("sample_negative: a user has no training positives (the reference divides by zero here); use the python-mode sampler for such datasets"); return 2; }
        if (deg >= item_num) { lgcn_set_error("sample_negative: a user is positive on every item (rejection loop would not end)"); return 2; }
    }
    if (workspace_bytes < lgcn_sample_negative_device_workspace(user_num, train_num)) { lgcn_set_error("sample_negative_device: workspace too small"); return 3; }
    const int64_t draws = 2 * T + T / 50 + 65536;
    const int64_t nblocks = (draws + GLIBC_BLOCK - 1) / GLIBC_BLOCK;
    hipStream_t st = (hipStream_t)stream;
    uint32_t *R = (uint32_t *)workspace;
    uint32_t *hist = R + nblocks * GLIBC_BLOCK;
    int64_t *result = (int64_t *)(((uintptr_t)(hist + nblocks * 31) + 15) & ~(uintptr_t)15);
    std::vector<uint32_t> h((size_t)nblocks * 31);
    lgcn_glibc_block_histories(nblocks, GLIBC_BLOCK, h.data());
    if (hipMemcpyAsync(hist, h.data(), h.size() * 4, hipMemcpyHostToDevice, st) != hipSuccess ||
        hipStreamSynchronize(st) != hipSuccess) { lgcn_set_error("sample_negative_device: history upload failed"); return 10; }
    hipLaunchKernelGGL(k_glibc_expand, dim3((unsigned)((nblocks + 63) / 64)), dim3(64), 0, st, hist, nblocks, R);
    SampleArgs a{R, nblocks * GLIBC_BLOCK, d_indptr, d_indices, user_num, item_num, per_user, T, d_S, result};
    hipLaunchKernelGGL(k_sample, dim3(1), dim3(SAMPLE_THREADS), 0, st, a);
    int64_t res[2] = {0, 1};
    if (hipMemcpyAsync(res, result, sizeof res, hipMemcpyDeviceToHost, st) != hipSuccess ||
        hipStreamSynchronize(st) != hipSuccess) { lgcn_set_error("sample_negative_device: kernel failed"); return 10; }
    if (res[1]) { lgcn_set_error("sample_negative_device: more rejections than the stream margin allows"); return 5; }
    lgcn_glibc_advance((uint64_t)res[0]);          // the host generator continues where the device stopped
    return 0;
}
