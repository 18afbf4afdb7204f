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
//
// Round 3 puts the bulk of that work on the whole GPU.  Whether a first candidate is rejected depends only on WHERE in
// the stream it sits and on the user: position p holds the candidate R[p] % m whatever triplet reads it, and while delta
// stays within [delta_in, delta_in + SAMP_DMAX] the triplets that can read p as their first candidate span SAMP_DMAX / 2
// consecutive triplets = a handful of users.  So, per segment of triplets:
//   k_samp_pairs  (all CUs)   every position of the segment x those few users -> the sparse list of rejecting
//                             (position, user) pairs, in position order (0.07 % of the tests on Gowalla);
//   k_samp_events (one workgroup, the pairs in LDS)   walks the chain: with the current delta a pair is LIVE when its
//                             position is the first-candidate position of a triplet of that user; all pairs are tested in
//                             parallel, the first live one is a rejection event, one thread walks its rejection loop, delta
//                             changes, the walk continues behind it.  Iterations = events, each a few LDS round trips;
//   k_samp_emit   (all CUs)   every triplet finds its delta among the segment's events and writes its row.
// The segments follow each other through a state record in device memory (first triplet, delta): no host round trip.  A
// segment ends early when delta has grown by more than SAMP_DMAX; any capacity that does not hold (pairs per block, pairs
// per segment) raises a flag and k_sample -- always launched last -- finishes from the state the segments reached.
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

// ---- segments on the whole GPU ---------------------------------------------------------------------------------------
#ifndef SAMP_SEGMENTS
#define SAMP_SEGMENTS 1                /* 0: the one-workgroup kernel alone (the round-2 form; A/B builds) */
#endif
#define SAMP_DMAX 128                 /* extra draws a segment may accumulate before it ends */
#define SAMP_SEG_MAX 131072           /* triplets per segment at most (a multiple of 256) */
#define SAMP_PCAP 256                 /* rejecting pairs a block of 256 positions can hold (a user positive on a quarter of the items fills 50-60) */
#define SAMP_LDS_PAIRS 8192           /* pairs of a segment the event walk holds in LDS */
#define SAMP_EV_MAX (SAMP_DMAX + 8)

struct SampState {
    long long t_begin;                // first triplet not yet produced
    long long delta;                  // extra draws consumed before it
    long long seg_t0, seg_t1, seg_delta;      // the segment just walked: triplets [seg_t0, seg_t1) start with seg_delta ...
    int n_events;                     // ... and these many rejection events
    int fail;                         // 1: a capacity did not hold (k_sample finishes from t_begin / delta); 2: stream exhausted
};
struct SampEvent { long long t, delta_after; };          // triplet, extra draws consumed once it is done
struct SegArgs {
    SampleArgs a;
    long long seg;                    // triplets per segment
    SampState *st;
    uint2 *pairs; int *pair_cnt;      // [blocks][SAMP_PCAP] (position - first position of the segment, user), [blocks]
    SampEvent *events;                // [SAMP_EV_MAX]
};

__global__ void __launch_bounds__(256) k_samp_pairs(SegArgs g) {
    const SampleArgs &a = g.a;
    const int tid = threadIdx.x;
    __shared__ int cnt[256];
    __shared__ int total;
    __shared__ long long s_t0, s_din;
    __shared__ int s_skip;
    // the state is read ONCE per workgroup: another workgroup of this launch may raise `fail` at any time, and a
    // workgroup whose threads saw different values would leave cnt[] half written
    if (tid == 0) { s_t0 = g.st->t_begin; s_din = g.st->delta; s_skip = g.st->fail != 0; }
    __syncthreads();
    const long long t0 = s_t0;
    if (t0 >= a.T || s_skip) return;
    const long long t1 = min((long long)a.T, t0 + g.seg), din = s_din;
    const long long pbase = 2 * t0 + din + 1, plast = 2 * (t1 - 1) + din + SAMP_DMAX + 1;
    const long long p = pbase + (long long)blockIdx.x * 256 + tid;
    int c = 0;
    int ulo = 0, uhi = -1, cand = 0;
    if (p <= plast && p < a.n_draws) {
        cand = (int)(a.R[p] % (uint32_t)a.item_num);
        // triplets that can have their first candidate at p: t = (p - 1 - d) / 2 for a d in [din, din + SAMP_DMAX]
        long long thi = (p - 1 - din) >> 1, tlo = (p - 1 - din - SAMP_DMAX + 1) >> 1;
        if (thi > t1 - 1) thi = t1 - 1;
        if (tlo < t0) tlo = t0;
        if (tlo <= thi) { ulo = (int)(tlo / a.per_user); uhi = (int)(thi / a.per_user); }
        for (int u = ulo; u <= uhi; u++) {
            const int64_t rs = a.indptr[u];
            if (is_positive(a.indices + rs, (int)(a.indptr[u + 1] - rs), cand)) c++;
        }
    }
    cnt[tid] = c;
    if (tid == 0) total = 0;
    __syncthreads();
    if (c > 0) {                                              // rare: position order is kept by an explicit prefix
        int off = 0;
        for (int i = 0; i < tid; i++) off += cnt[i];
        atomicAdd(&total, c);
        uint2 *dst = g.pairs + (long long)blockIdx.x * SAMP_PCAP;
        for (int u = ulo; u <= uhi; u++) {
            const int64_t rs = a.indptr[u];
            if (is_positive(a.indices + rs, (int)(a.indptr[u + 1] - rs), cand)) {
                if (off < SAMP_PCAP) dst[off] = make_uint2((uint32_t)(p - pbase), (uint32_t)u);
                off++;
            }
        }
    }
    __syncthreads();
    if (tid == 0) {
        g.pair_cnt[blockIdx.x] = total;
        if (total > SAMP_PCAP) atomicExch(&g.st->fail, 1);
    }
}

__global__ void __launch_bounds__(256) k_samp_events(SegArgs g, int nblocks) {
    const SampleArgs &a = g.a;
    __shared__ uint2 pr[SAMP_LDS_PAIRS];
    __shared__ int part[256];
    __shared__ int s_first, s_i0, s_end, s_np;
    __shared__ long long s_tcur, s_d;
    const int tid = threadIdx.x;
    SampState *st = g.st;
    const long long t0 = st->t_begin;
    if (t0 >= a.T || st->fail) { if (tid == 0) { st->seg_t0 = st->seg_t1 = 0; st->n_events = 0; } return; }
    const long long t1 = min((long long)a.T, t0 + g.seg), din = st->delta;
    const long long pbase = 2 * t0 + din + 1;
    // ---- the blocks' pairs, in block (= position) order, into LDS
    const int per = (nblocks + 255) / 256, b0 = tid * per, b1 = min(nblocks, b0 + per);
    int mine = 0;
    for (int b = b0; b < b1; b++) mine += min(g.pair_cnt[b], SAMP_PCAP);
    part[tid] = mine;
    __syncthreads();
    if (tid == 0) { int run = 0; for (int i = 0; i < 256; i++) { const int v = part[i]; part[i] = run; run += v; } s_np = run; }
    __syncthreads();
    const int np = s_np;
    if (np > SAMP_LDS_PAIRS) {                                // this segment keeps its state: k_sample takes over from it
        if (tid == 0) { atomicExch(&st->fail, 1); st->seg_t0 = st->seg_t1 = 0; st->n_events = 0; }
        return;
    }
    {
        int o = part[tid];
        for (int b = b0; b < b1; b++) {
            const int n = min(g.pair_cnt[b], SAMP_PCAP);
            for (int i = 0; i < n; i++) pr[o++] = g.pairs[(long long)b * SAMP_PCAP + i];
        }
    }
    if (tid == 0) { s_tcur = t0; s_d = din; s_i0 = 0; s_end = 0; }
    __syncthreads();
    int nev = 0;                                               // (thread 0's count)
    long long seg_end = t1;
    while (true) {
        const long long tcur = s_tcur, d = s_d;
        const int i0 = s_i0;
        if (s_end || i0 >= np) break;
        __syncthreads();
        if (tid == 0) s_first = 0x7fffffff;
        __syncthreads();
        const int i1 = min(np, i0 + 1024);
        for (int i = i0 + tid; i < i1; i += 256) {
            const long long x = pbase + pr[i].x - 1 - d;
            if (x >= 0 && !(x & 1)) {
                const long long tt = x >> 1;
                if (tt >= tcur && tt < t1 && (uint32_t)(tt / a.per_user) == pr[i].y) atomicMin(&s_first, i);
            }
        }
        __syncthreads();
        const int f = s_first;
        if (tid == 0) {
            if (f == 0x7fffffff) s_i0 = i1;
            else {                                              // a rejection event: walk its loop, the only serial part
                const long long pf = pbase + pr[f].x;           // position of the rejected first candidate
                const long long tt = (pf - 1 - d) >> 1;
                const uint32_t u = pr[f].y;
                // the following candidates sit at pf + 1, pf + 2, ...: whether this user rejects them is in the pair list too
                // (the user stays inside those positions' windows while delta stays inside the segment's range) -- LDS only
                long long q = pf + 1;
                int j = f + 1;
                bool ok = true;
                while (true) {
                    if (q >= a.n_draws) { ok = false; break; }
                    bool rej = false;
                    if (d + (q - pf) > din + SAMP_DMAX) {       // beyond the range the pairs cover: ask the row itself
                        const int64_t rs = a.indptr[u];
                        rej = is_positive(a.indices + rs, (int)(a.indptr[u + 1] - rs), (int)(a.R[q] % (uint32_t)a.item_num));
                    } else {
                        while (j < np && pbase + pr[j].x < q) j++;
                        for (int jj = j; jj < np && pbase + pr[jj].x == q; jj++) rej |= pr[jj].y == u;
                    }
                    if (!rej) break;
                    q++;
                }
                if (!ok) { atomicExch(&st->fail, 2); s_end = 1; seg_end = tt; }      // (k_sample reports the exhausted stream)
                else {
                    const long long d_after = d + (q - pf);
                    g.events[nev++] = SampEvent{tt, d_after};
                    s_tcur = tt + 1; s_d = d_after;
                    while (j < np && pbase + pr[j].x < q + 2) j++;   // triplet tt + 1 starts at q + 1: its first candidate sits at q + 2
                    s_i0 = j;
                    if (d_after - din > SAMP_DMAX || nev >= SAMP_EV_MAX - 1) { s_end = 1; seg_end = tt + 1; }
                }
            }
        }
        __syncthreads();
    }
    if (tid == 0) {
        // Every draw the triplets [t0, seg_end) read lies below 2 * seg_end + delta_out.  k_samp_pairs skipped the positions
        // past the end of the expanded stream (they count as accepted) and k_samp_emit reads R unchecked, so a segment that
        // reaches past the stream is DISCARDED: its state stays where it was, `fail` is raised and k_sample -- which checks
        // every window against n_draws -- finishes from there and reports the exhausted stream (rc 5) if it really is.
        if (2 * seg_end + s_d > a.n_draws) {
            atomicExch(&st->fail, 1);
            st->seg_t0 = st->seg_t1 = 0; st->n_events = 0;
        } else {
            st->seg_t0 = t0; st->seg_t1 = seg_end; st->seg_delta = din; st->n_events = nev;
            st->t_begin = seg_end; st->delta = s_d;
        }
    }
}

__global__ void __launch_bounds__(256) k_samp_emit(SegArgs g) {
    const SampleArgs &a = g.a;
    __shared__ SampEvent ev[SAMP_EV_MAX];
    const SampState *st = g.st;
    const long long t0 = st->seg_t0, t1 = st->seg_t1;
    const long long t = t0 + (long long)blockIdx.x * 256 + threadIdx.x;
    if (t0 + (long long)blockIdx.x * 256 >= t1) return;
    const int nev = st->n_events;
    for (int i = threadIdx.x; i < nev; i += 256) ev[i] = g.events[i];
    __syncthreads();
    if (t >= t1) return;
    int lo = 0, hi = nev;                                      // events with ev.t < t
    while (lo < hi) { const int mid = (lo + hi) >> 1; if (ev[mid].t < t) lo = mid + 1; else hi = mid; }
    const int u = (int)(t / a.per_user);
    const long long d = lo > 0 ? ev[lo - 1].delta_after : st->seg_delta;
    // an event triplet's accepted candidate follows its rejected ones: their number is what the event added to delta
    const long long nrej = (lo < nev && ev[lo].t == t) ? ev[lo].delta_after - d : 0;
    const int64_t rs = a.indptr[u];
    const int deg = (int)(a.indptr[u + 1] - rs);
    const long long base = 2 * t + d;
    a.S[3 * t] = u;
    a.S[3 * t + 1] = a.indices[rs + (int)(a.R[base] % (uint32_t)deg)];
    a.S[3 * t + 2] = (int)(a.R[base + 1 + nrej] % (uint32_t)a.item_num);
}

// ---- one workgroup: the finisher (and, alone, the whole job: `init` NULL) -------------------------------------------
#define SAMPLE_THREADS 1024
__global__ void __launch_bounds__(SAMPLE_THREADS) k_sample(SampleArgs a, const SampState *init) {
    __shared__ long long first_rej;      // smallest rejecting triplet of the window
    __shared__ long long s_t0, s_delta;
    __shared__ int s_fail;
    const int tid = threadIdx.x;
    if (tid == 0) { s_t0 = init ? init->t_begin : 0; s_delta = init ? init->delta : 0; s_fail = init && init->fail == 2 ? 1 : 0; }
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

// Stream margin: draws expanded beyond the 2 per triplet every epoch needs = T / g_margin_div + g_margin_fixed (2 % + slack).
// lgcn_sampler_test_margin is a TEST HOOK (tests/test_gpu_parity.py drives the segment path into the end of the stream
// with it); values <= 0 restore the defaults.
static int64_t g_margin_div = 50, g_margin_fixed = 65536;
extern "C" void lgcn_sampler_test_margin(int64_t divisor, int64_t fixed) {
    g_margin_div = divisor > 0 ? divisor : 50;
    g_margin_fixed = fixed > 0 ? fixed : 65536;
}
static inline int64_t stream_draws(int64_t T) { return 2 * T + T / g_margin_div + g_margin_fixed; }

static inline int64_t seg_pair_blocks() { return (2 * (int64_t)SAMP_SEG_MAX + SAMP_DMAX) / 256 + 2; }
static inline int64_t seg_workspace_bytes() {
    return 256 /* state */ + seg_pair_blocks() * (SAMP_PCAP * 8 + 4) + SAMP_EV_MAX * (int64_t)sizeof(SampEvent) + 256;
}

extern "C" int64_t lgcn_sample_negative_device_workspace(int user_num, int64_t train_num) {
    if (user_num <= 0 || train_num < 0) return 0;
    const int64_t T = (int64_t)user_num * (train_num / user_num);
    const int64_t draws = stream_draws(T);
    const int64_t nblocks = (draws + GLIBC_BLOCK - 1) / GLIBC_BLOCK;
    return nblocks * GLIBC_BLOCK * 4 + nblocks * 31 * 4 + 256 + seg_workspace_bytes();
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
    const int64_t draws = stream_draws(T);
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
    // ---- segments on the whole GPU, then the one-workgroup kernel from wherever they got to (normally: the end)
    char *seg_ws = (char *)(((uintptr_t)(result + 2) + 255) & ~(uintptr_t)255);
    SegArgs g{a, 0, (SampState *)seg_ws, nullptr, nullptr, nullptr};
    g.pairs = (uint2 *)(seg_ws + 256);
    g.pair_cnt = (int *)(g.pairs + seg_pair_blocks() * SAMP_PCAP);
    g.events = (SampEvent *)(((uintptr_t)(g.pair_cnt + seg_pair_blocks()) + 15) & ~(uintptr_t)15);
    // k_samp_pairs tests every position against the SAMP_DMAX / 2 / per_user + 1 users whose triplets can read it: with very few
    // interactions per user that is dozens of binary searches per position -- the one-workgroup kernel does those datasets
    const bool segments = SAMP_SEGMENTS != 0 && per_user >= 8;
    if (hipMemsetAsync(g.st, 0, sizeof(SampState), st) != hipSuccess) { lgcn_set_error("sample_negative_device: memset failed"); return 10; }
    if (segments) {
        // a segment should end by itself, not on SAMP_DMAX: size it for half that many expected rejections (rate = deg / items)
        const double rate = (double)h_indptr[user_num] / (double)user_num / (double)item_num;
        long long seg = rate > 0 ? (long long)(SAMP_DMAX / (2.0 * rate)) : SAMP_SEG_MAX;
        seg = seg > SAMP_SEG_MAX ? SAMP_SEG_MAX : seg < 4096 ? 4096 : seg / 256 * 256;
        g.seg = seg;
        const long long nseg = (T + seg - 1) / seg * 5 / 4 + 4;
        const unsigned pair_blocks = (unsigned)((2 * seg + SAMP_DMAX) / 256 + 1);
        for (long long s = 0; s < nseg; s++) {
            hipLaunchKernelGGL(k_samp_pairs, dim3(pair_blocks), dim3(256), 0, st, g);
            hipLaunchKernelGGL(k_samp_events, dim3(1), dim3(256), 0, st, g, (int)pair_blocks);
            hipLaunchKernelGGL(k_samp_emit, dim3((unsigned)(seg / 256)), dim3(256), 0, st, g);
        }
    }
    hipLaunchKernelGGL(k_sample, dim3(1), dim3(SAMPLE_THREADS), 0, st, a, (const SampState *)g.st);
    int64_t res[2] = {0, 1};
    if (hipMemcpyAsync(res, result, sizeof res, hipMemcpyDeviceToHost, st) != hipSuccess ||
        hipStreamSynchronize(st) != hipSuccess) { lgcn_set_error("sample_negative_device: kernel failed"); return 10; }
    if (res[1]) { lgcn_set_error("sample_negative_device: more rejections than the stream margin allows"); return 5; }
    lgcn_glibc_advance((uint64_t)res[0]);          // the host generator continues where the device stopped
    return 0;
}
