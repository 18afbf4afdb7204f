// lgcn_shuffle.hip -- the epoch shuffle on the GPU, bit for bit.
//
// Replaces  idx = np.arange(n); np.random.shuffle(idx)  (utils.py:148-149 of the reference; numpy's LEGACY global RandomState:
// MT19937 + Fisher-Yates from the top with masked-rejection bounded integers) with the SAME permutation from the SAME stream,
// produced on the device: the host loop costs ~5 ms per Gowalla epoch (hidden only when the epoch prefetch is on) and
// seconds at 200 M triplets.  The reference loop is serial three times over; each part has a parallel form:
//
//   1. MT19937.  A twist of the 624-word state is serial as written (mt[i] depends on mt[i+1], mt[i+397 mod 624]) but falls
//      into passes whose inputs are all final or all old: 64 consecutive words at a time, every lane reading its three words
//      before any lane writes (mt[i+1] still old; the far word old for i < 227, new -- written by an earlier pass -- after).
//      ONE wave twists in LDS with no workgroup barrier (four LDS round trips per twist) and keeps the 624 tempered outputs
//      in registers; its final state goes back to the host generator, which continues where the device stopped.
//   2. Which draw belongs to which step.  Step i takes draws until (draw & mask_i) <= i, so the alignment is data dependent,
//      but only weakly: within 64 consecutive draws under one mask, a draw v <= i0 - 64 is accepted whatever came before
//      and v > i0 is rejected whatever came before; only i0 - 64 < v <= i0 depends on the count so far (a few draws in 10^5).
//      The same wave aligns the stream as it generates it, 64 draws per pass (two ballots + popcounts; the rare ambiguous
//      draws are resolved in order from the ballot masks), serially only across a mask boundary and below step 1024
//      (k_fy_stream) -- no stream buffer in memory, no draw budget that could be exceeded.
//      (First form, measured: a 256-thread twist with a barrier pair per phase + a separate alignment kernel over a stream
//      buffer: 2.6 + 3.4 ms at Gowalla's 806 166 -- no faster than the host loop.)
//   3. The swaps.  x[i] <-> x[j_i] for i = n-1 .. 1 is a chain of dependent swaps, but where the value of position p ends up is
//      a walk through the steps that touch it: sort the steps by (j_i, i) (rocPRIM radix sort), the parent of node i is the
//      next step that targets i, pointer doubling finds the chains' roots (tools/fy_parallel_prototype.py checks this form
//      against np.random.shuffle on the CPU).
// tests/test_gpu_parity.py compares the device permutation and the generator position afterwards with the host restatement
// (lgcn_np_shuffle_perm), which tests/test_oracle.py pins to numpy itself.
#include <hip/hip_runtime.h>
#include <cstring>
#include <rocprim/device/device_radix_sort.hpp>
#include <stdint.h>
#include <math.h>

#include "lgcn_hip.h"
#include "lgcn_internal.h"

extern "C" void lgcn_np_get_state(uint32_t *key624, uint32_t *pos);          // lgcn_host.cpp
extern "C" void lgcn_np_set_state(const uint32_t *key624, uint32_t pos);

#define MT_N 624
#define MT_M 397
#define FY_TAIL 1024          /* steps below this index are aligned serially (the ambiguity window is no longer small) */

__device__ __forceinline__ uint32_t mt_temper(uint32_t y) {
    y ^= y >> 11;
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= y >> 18;
    return y;
}
__device__ __forceinline__ uint32_t mt_mix(uint32_t hi, uint32_t lo, uint32_t far) {
    const uint32_t y = (hi & 0x80000000u) | (lo & 0x7fffffffu);
    return far ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
}
// One twist of the state in LDS by ONE wave, no workgroup barrier.  The sequential loop's data flow (mt[i] from the OLD mt[i],
// OLD mt[i+1] and mt[i+397 mod 624], which is old for i < 227 and NEW -- already rewritten -- after) is kept by doing 64 words
// per pass in increasing i, every lane reading before any lane writes; passes whose inputs are all available are issued
// together (three at a time: i in [0,192) reads old words only, [192,384) the new words of the first group, [384,576) of the
// first two, [576,624) -- with mt[624] = the NEW mt[0] -- of the rest), so a twist costs four LDS round trips, not ten.
// t[k] = the tempered output of word 64 k + lane (the generator's next 624 outputs, in registers).
__device__ __forceinline__ void mt_twist(uint32_t *mt, int lane, uint32_t t[10]) {
    uint32_t v[10];
#pragma unroll
    for (int grp = 0; grp < 4; grp++) {
        const int k0 = 3 * grp, k1 = grp == 3 ? 10 : k0 + 3;
#pragma unroll
        for (int k = k0; k < k1; k++) {
            const int i = 64 * k + lane;
            v[k] = 0;
            if (i < MT_N) v[k] = mt_mix(mt[i], mt[i + 1 == MT_N ? 0 : i + 1], mt[i + MT_M >= MT_N ? i + MT_M - MT_N : i + MT_M]);
        }
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int k = k0; k < k1; k++) { const int i = 64 * k + lane; if (i < MT_N) mt[i] = v[k]; }
        __builtin_amdgcn_wave_barrier();
    }
#pragma unroll
    for (int k = 0; k < 10; k++) t[k] = mt_temper(v[k]);
}

__device__ __forceinline__ uint32_t fy_mask(uint32_t top) {
    uint32_t m = top;
    m |= m >> 1; m |= m >> 2; m |= m >> 4; m |= m >> 8; m |= m >> 16;
    return m;
}

// One pass of the alignment: the next c <= 64 draws of the stream (lane l holds draw l, already tempered) against the current
// step i under `mask`: J[i - (accepted before)] = draw for every accepted draw; returns the number accepted.  A draw v <= i - 64
// is accepted whatever came before it in the pass, v > i rejected whatever came before; the few in between are resolved in
// stream order from the ballot masks.  Everything but the compare and the store is wave-uniform (scalar registers).
__device__ __forceinline__ int fy_pass(uint32_t draw, int c, int i, uint32_t mask, uint32_t *J, int lane) {
    const uint32_t v = draw & mask;
    const bool in = lane < c;
    const bool def = in && (int)v <= i - 64;           // (i >= 1024 here and v <= mask < 2^31: int compares are safe)
    const bool amb = in && !def && v <= (uint32_t)i;
    unsigned long long acc = __ballot(def), a = __ballot(amb);
    while (a) {
        const int l = __builtin_amdgcn_readfirstlane(__builtin_ctzll(a));
        a &= a - 1ull;
        const int cnt = __popcll(acc & ((1ull << l) - 1ull));
        const uint32_t val = (uint32_t)__builtin_amdgcn_readlane((int)v, l);
        if (val <= (uint32_t)(i - cnt)) acc |= 1ull << l;
    }
    if ((acc >> lane) & 1ull) J[i - __popcll(acc & ((1ull << lane) - 1ull))] = v;
    return __builtin_amdgcn_readfirstlane(__popcll(acc));
}

// J[i] = the accepted draw of step i (i = n-1 .. 1) of np.random.shuffle on the generator state (key0, pos0); the state the host
// loop would be left in comes back in key_out / result[0] (read position).  ONE wave generates the stream block by block (no
// stream buffer in memory, no budget to exceed) and aligns it as it comes: a freshly twisted block is consumed straight from the
// registers the twist left its outputs in, 64 draws per pass; after a serial stretch (across a mask boundary, or below step
// FY_TAIL where the ambiguity window is no longer small) the rest of the block is read back from LDS.  n < 2^31: the step
// index, the block position and the masks live in scalar registers.
__global__ void __launch_bounds__(64) k_fy_stream(const uint32_t *key0, uint32_t pos0, int64_t n, uint32_t *J, uint32_t *key_out, long long *result) {
    __shared__ uint32_t mt[MT_N];
    const int lane = threadIdx.x;
    for (int k = lane; k < MT_N; k += 64) mt[k] = key0[k];
    __builtin_amdgcn_wave_barrier();
    int idx = __builtin_amdgcn_readfirstlane(pos0 > MT_N ? MT_N : (int)pos0);
    int i = __builtin_amdgcn_readfirstlane((int)(n - 1));
    if (lane == 0) J[0] = 0u;
    while (i >= 1) {
        if (idx >= MT_N) {
            uint32_t t[10];
            mt_twist(mt, lane, t);
            idx = 0;
            // fast path: the whole new block from registers when its 624 draws cannot reach a mask boundary or the tail.  The
            // ten compares are independent of each other (thresholds of the block's FIRST step: a draw <= i0 - 624 is accepted
            // wherever the block stands, a draw > i0 never is), so only scalar popcounts chain from pass to pass.
            const uint32_t mask = fy_mask((uint32_t)i);
            if (i - MT_N >= FY_TAIL && i - MT_N >= (int)(mask >> 1) + 1) {
                unsigned long long acc[10], am[10];
                uint32_t v[10];
#pragma unroll
                for (int k = 0; k < 10; k++) {
                    v[k] = t[k] & mask;
                    const bool in = k < 9 || lane < MT_N - 576;
                    const bool def = in && (int)v[k] <= i - MT_N;
                    acc[k] = __ballot(def);
                    am[k] = __ballot(in && !def && v[k] <= (uint32_t)i);
                }
#pragma unroll
                for (int k = 0; k < 10; k++) {
                    unsigned long long a = am[k];
                    while (a) {                                         // (a draw in 1 500: exact test against the running step)
                        const int l = __builtin_amdgcn_readfirstlane(__builtin_ctzll(a));
                        a &= a - 1ull;
                        const int cnt = __popcll(acc[k] & ((1ull << l) - 1ull));
                        if ((uint32_t)__builtin_amdgcn_readlane((int)v[k], l) <= (uint32_t)(i - cnt)) acc[k] |= 1ull << l;
                    }
                    if ((acc[k] >> lane) & 1ull) J[i - __popcll(acc[k] & ((1ull << lane) - 1ull))] = v[k];
                    i -= __builtin_amdgcn_readfirstlane(__popcll(acc[k]));
                }
                idx = MT_N;
                continue;
            }
            // otherwise pass by pass, still from registers, while the steps stay clear of a mask boundary and the tail
            bool go = true;
#pragma unroll
            for (int k = 0; k < 10; k++) {
                const uint32_t mk = fy_mask((uint32_t)i);
                go = go && i >= FY_TAIL && i - 64 >= (int)(mk >> 1) + 1;
                if (go) {
                    const int c = k == 9 ? MT_N - 576 : 64;
                    i -= fy_pass(t[k], c, i, mk, J, lane);
                    idx += c;
                }
            }
            continue;
        }
        const uint32_t mask = fy_mask((uint32_t)i);
        const int low = (int)(mask >> 1) + 1;                           // the smallest step that draws under this mask
        if (i < FY_TAIL || i - 64 < low) {
            // serial stretch by one lane: down to the mask boundary or to the end, within this block
            const int stop = i < FY_TAIL ? 1 : low;
            int ii = i, id = idx;
            if (lane == 0) {
                while (ii >= stop && id < MT_N) {
                    const uint32_t v = mt_temper(mt[id++]) & fy_mask((uint32_t)ii);
                    if (v <= (uint32_t)ii) { J[ii] = v; ii--; }
                }
            }
            i = __builtin_amdgcn_readfirstlane(ii); idx = __builtin_amdgcn_readfirstlane(id);
            continue;
        }
        const int c = MT_N - idx < 64 ? MT_N - idx : 64;
        const uint32_t draw = lane < c ? mt_temper(mt[idx + lane]) : 0u;
        i -= fy_pass(draw, c, i, mask, J, lane);
        idx += c;
    }
    __builtin_amdgcn_wave_barrier();
    for (int k = lane; k < MT_N; k += 64) key_out[k] = mt[k];
    if (lane == 0) { result[0] = idx; result[1] = 0; }
}

__global__ void __launch_bounds__(256) k_fy_keys(const uint32_t *J, int64_t n, unsigned long long *keys) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x + 1;
    if (i < n) keys[i - 1] = (unsigned long long)J[i] * (unsigned long long)(n + 1) + (unsigned long long)i;
}
// smallest step i > t with J[i] == q, or -1: upper bound of (q, t) among the sorted (J[i], i) keys
__device__ __forceinline__ long long fy_next(const unsigned long long *comp, int64_t m, int64_t n, uint32_t q, int64_t t) {
    const unsigned long long key = (unsigned long long)q * (unsigned long long)(n + 1) + (unsigned long long)t;
    int64_t lo = 0, hi = m;
    while (lo < hi) { const int64_t mid = (lo + hi) >> 1; if (comp[mid] <= key) lo = mid + 1; else hi = mid; }
    if (lo >= m) return -1;
    const unsigned long long c = comp[lo];
    return (c / (unsigned long long)(n + 1) == (unsigned long long)q) ? (long long)(c % (unsigned long long)(n + 1)) : -1;
}
__global__ void __launch_bounds__(256) k_fy_parent(const unsigned long long *comp, int64_t n, uint32_t *f) {
    const int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (p >= n) return;
    const long long nx = fy_next(comp, n - 1, n, (uint32_t)p, p);
    f[p] = nx >= 0 ? (uint32_t)nx : (uint32_t)p;
}
__global__ void __launch_bounds__(256) k_fy_double(const uint32_t *g, uint32_t *out, int64_t n) {
    const int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (p < n) out[p] = g[g[p]];
}
__global__ void __launch_bounds__(256) k_fy_out(const unsigned long long *comp, const uint32_t *J, const uint32_t *root, int64_t n, int64_t *perm) {
    const int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (p >= n) return;
    const uint32_t q = p == 0 ? 0u : J[p];
    const long long first = fy_next(comp, n - 1, n, q, p);
    perm[p] = first >= 0 ? (int64_t)root[first] : (int64_t)q;
}

namespace {
struct Layout { size_t o_key0, o_keyf, o_res, o_J, o_ka, o_kb, o_g0, o_g1, o_tmp, tmp_bytes, total; };
inline size_t up256(size_t x) { return (x + 255) & ~(size_t)255; }
Layout layout(int64_t n) {
    Layout L{};
    size_t tb = 0;
    (void)rocprim::radix_sort_keys(nullptr, tb, (const unsigned long long *)nullptr, (unsigned long long *)nullptr, (size_t)(n > 1 ? n - 1 : 1), 0, 64, (hipStream_t)0);
    L.tmp_bytes = tb;
    size_t o = 0;
    L.o_key0 = o; o = up256(o + MT_N * 4);
    L.o_keyf = o; o = up256(o + MT_N * 4);
    L.o_res = o; o = up256(o + 32);
    L.o_J = o; o = up256(o + (size_t)n * 4);
    L.o_ka = o; o = up256(o + (size_t)n * 8);
    L.o_kb = o; o = up256(o + (size_t)n * 8);
    L.o_g0 = o; o = up256(o + (size_t)n * 4);
    L.o_g1 = o; o = up256(o + (size_t)n * 4);
    L.o_tmp = o; o = up256(o + tb);
    L.total = o;
    return L;
}
}  // namespace

extern "C" int64_t lgcn_np_shuffle_perm_device_workspace(int64_t n) {
    if (n < 2 || n > 0x7ffffff0LL) return 256;
    return (int64_t)layout(n).total;
}

extern "C" int lgcn_np_shuffle_perm_device(int64_t n, int64_t *d_perm, void *workspace, int64_t workspace_bytes, void *stream) {
    if (n < 0 || (n > 0 && !d_perm)) { lgcn_set_error("np_shuffle_perm_device: invalid argument"); return 3; }
    hipStream_t st = (hipStream_t)stream;
    if (n == 0) return 0;
    if (n == 1) { if (hipMemsetAsync(d_perm, 0, sizeof(int64_t), st) != hipSuccess) { lgcn_set_error("np_shuffle_perm_device: memset failed"); return 10; } return 0; }
    if (n > 0x7ffffff0LL) { lgcn_set_error("np_shuffle_perm_device: n too large for the device form (use lgcn_np_shuffle_perm)"); return 3; }
    const Layout L = layout(n);
    if (!workspace || workspace_bytes < (int64_t)L.total) { lgcn_set_error("np_shuffle_perm_device: workspace too small"); return 3; }
    char *ws = (char *)workspace;
    uint32_t *key0 = (uint32_t *)(ws + L.o_key0), *keyf = (uint32_t *)(ws + L.o_keyf);
    uint32_t *J = (uint32_t *)(ws + L.o_J), *g0 = (uint32_t *)(ws + L.o_g0), *g1 = (uint32_t *)(ws + L.o_g1);
    unsigned long long *ka = (unsigned long long *)(ws + L.o_ka), *kb = (unsigned long long *)(ws + L.o_kb);
    long long *res = (long long *)(ws + L.o_res);
    uint32_t hkey[MT_N], hpos = 0;
    lgcn_np_get_state(hkey, &hpos);
    if (hipMemcpyAsync(key0, hkey, sizeof hkey, hipMemcpyHostToDevice, st) != hipSuccess ||
        hipStreamSynchronize(st) != hipSuccess) { lgcn_set_error("np_shuffle_perm_device: state upload failed"); return 10; }
    hipLaunchKernelGGL(k_fy_stream, dim3(1), dim3(64), 0, st, (const uint32_t *)key0, hpos, n, J, keyf, res);
    const unsigned blocks = (unsigned)((n + 255) / 256);
    hipLaunchKernelGGL(k_fy_keys, dim3(blocks), dim3(256), 0, st, (const uint32_t *)J, n, ka);
    int bits = 1;
    while (bits < 64 && ((unsigned long long)(n + 1) * (unsigned long long)(n + 1)) >> bits) bits++;
    size_t tb = L.tmp_bytes;
    if (rocprim::radix_sort_keys(ws + L.o_tmp, tb, (const unsigned long long *)ka, kb, (size_t)(n - 1), 0, bits, st) != hipSuccess) {
        lgcn_set_error("np_shuffle_perm_device: radix sort failed"); return 10; }
    hipLaunchKernelGGL(k_fy_parent, dim3(blocks), dim3(256), 0, st, (const unsigned long long *)kb, n, g0);
    uint32_t *cur = g0, *nxt = g1;
    for (int64_t span = 1; span < n; span <<= 1) {                 // pointer doubling: after r rounds a node points 2^r hops up its chain
        hipLaunchKernelGGL(k_fy_double, dim3(blocks), dim3(256), 0, st, (const uint32_t *)cur, nxt, n);
        uint32_t *t = cur; cur = nxt; nxt = t;
    }
    hipLaunchKernelGGL(k_fy_out, dim3(blocks), dim3(256), 0, st, (const unsigned long long *)kb, (const uint32_t *)J, (const uint32_t *)cur, n, d_perm);
    long long hres[2] = {0, 0};
    if (hipMemcpyAsync(hkey, keyf, sizeof hkey, hipMemcpyDeviceToHost, st) != hipSuccess ||
        hipMemcpyAsync(hres, res, sizeof hres, hipMemcpyDeviceToHost, st) != hipSuccess ||
        hipStreamSynchronize(st) != hipSuccess || hipGetLastError() != hipSuccess) { lgcn_set_error("np_shuffle_perm_device: kernels failed"); return 10; }
    const uint32_t idx = (uint32_t)hres[0];
    lgcn_np_set_state(hkey, idx);          // the host generator continues where the device stopped
    return 0;
}
