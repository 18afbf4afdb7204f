// lgcn_device.hip -- hand-written gfx950 (CDNA4, wave64) kernels of the LightGCN/BPR
// training hot path and their C-ABI launchers (include/lgcn_hip.h).
//
// The path is sparse and HBM/cache-bandwidth bound (<= 0.5 flop/byte): no MFMA.
// What matters here: 16-byte (fp32) / 8-byte (bf16) per-lane row gathers that
// cover whole 128/256-byte embedding rows, fp32 accumulation, wavefront
// (ds_swizzle/DPP) reductions, fused epilogues so that no dense [N,d]
// intermediate is written twice, and a launch geometry of >> 256 workgroups.
//
// Reference semantics (LightGCN_work/code): model.py:201-231 (computer),
// model.py:162-183 (bpr_loss), utils.py:53-64 (stageOne = fwd + backward + Adam).
//
// Algebra used (exact restructuring):
//  * forward needs dense X_1..X_{K-1} only; the last layer X_K and the layer mean
//    are evaluated on the <= 3B rows the batch gathers (k_bpr).
//  * backward is the Horner chain h_{k-1} = Gs + A h_k, Gs = G/(K+1); Gs has
//    <= 3B non-zero rows, so the first backward SpMM skips zero rows by bitmap.
//  * the last backward SpMM applies Adam in its epilogue (no dense grad buffer).
//  * the scatter-add of per-triplet gradient rows uses 64-bit fixed-point integer
//    atomics (scale 2^50): integer addition is associative, so the result is
//    bitwise reproducible and independent of batch sharding.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <math.h>
#include <new>

#include "lgcn_hip.h"

extern "C" void lgcn_set_error(const char *msg);   // lgcn_host.cpp

#define HIP_OK(expr)                                                            \
    do {                                                                        \
        hipError_t e_ = (expr);                                                 \
        if (e_ != hipSuccess) {                                                 \
            char buf_[256];                                                     \
            snprintf(buf_, sizeof buf_, "%s failed: %s", #expr, hipGetErrorString(e_)); \
            lgcn_set_error(buf_);                                               \
            return 10;                                                          \
        }                                                                       \
    } while (0)

typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __bf16 bf16_t;

#define FIXED_SCALE 1125899906842624.0   /* 2^50 */
#define FIXED_INV   8.8817841970012523e-16 /* 2^-50 */

__device__ __forceinline__ f32x4 load4(const float *p) { return *reinterpret_cast<const f32x4 *>(p); }
__device__ __forceinline__ f32x4 load4(const bf16_t *p) {
    bf16x4 v = *reinterpret_cast<const bf16x4 *>(p);
    return __builtin_convertvector(v, f32x4);
}
__device__ __forceinline__ void store4(float *p, f32x4 v) { *reinterpret_cast<f32x4 *>(p) = v; }
__device__ __forceinline__ void store4(bf16_t *p, f32x4 v) {
    *reinterpret_cast<bf16x4 *>(p) = __builtin_convertvector(v, bf16x4);
}
__device__ __forceinline__ f32x4 shfl_xor4(f32x4 v, int m) {
    f32x4 r;
    r.x = __shfl_xor(v.x, m); r.y = __shfl_xor(v.y, m); r.z = __shfl_xor(v.z, m); r.w = __shfl_xor(v.w, m);
    return r;
}
__device__ __forceinline__ bool bit_set(const uint32_t *bm, int i) { return (bm[i >> 5] >> (i & 31)) & 1u; }

// ---------------------------------------------------------------------------------
// One CSR row of  A_hat * X  computed by one wavefront.
// Lane layout: LPR = D/4 lanes cover one embedding row with 4 columns each
// (16 B fp32 / 8 B bf16 per lane); the wave's 64/LPR lane groups walk the row's
// neighbours interleaved (group g takes nnz p = start+g, start+g+NPW, ...), U
// neighbours deep, so every lane keeps U independent row loads in flight.  The
// per-group partial sums are combined by xor-shuffles in a fixed order: the
// result is deterministic and identical wherever this function is used (dense
// SpMM, on-the-fly last layer in k_bpr).
// ---------------------------------------------------------------------------------
template <int D, typename TI, bool SPARSE>
__device__ __forceinline__ f32x4 row_gather(const int32_t *__restrict__ indices,
                                            const float *__restrict__ vals, int start, int end,
                                            const TI *__restrict__ X, const uint32_t *__restrict__ bm,
                                            int lane) {
    constexpr int LPR = D / 4, NPW = 64 / LPR, U = 4;
    const int g = lane / LPR, l = lane % LPR;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    int p = start + g;
    for (; p + (U - 1) * NPW < end; p += U * NPW) {
        int col[U]; float v[U]; f32x4 x[U];
#pragma unroll
        for (int u = 0; u < U; u++) { col[u] = indices[p + u * NPW]; v[u] = vals[p + u * NPW]; }
#pragma unroll
        for (int u = 0; u < U; u++) {
            if (!SPARSE || bit_set(bm, col[u])) x[u] = load4(X + (int64_t)col[u] * D + l * 4);
            else x[u] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int u = 0; u < U; u++) acc += v[u] * x[u];
    }
    for (; p < end; p += NPW) {
        const int col = indices[p]; const float v = vals[p];
        if (!SPARSE || bit_set(bm, col)) acc += v * load4(X + (int64_t)col * D + l * 4);
    }
#pragma unroll
    for (int off = LPR; off < 64; off <<= 1) acc += shfl_xor4(acc, off);
    return acc;   // every lane group holds the full sum for its 4 columns
}

// XCD-aware block->row-tile map: hardware deals workgroups round-robin over the 8
// XCDs (speed-only observation), so give XCD x the x-th contiguous eighth of the
// tiles: CSR / output streams of one L2 stay contiguous, and user rows (which
// gather item rows) and item rows (which gather user rows) land on disjoint XCDs.
__device__ __forceinline__ int64_t tile_of_block(int64_t bid, int64_t ntiles, int remap) {
    if (!remap) return bid;
    const int64_t per = (ntiles + 7) / 8;
    return (bid & 7) * per + (bid >> 3);   // may be >= ntiles: caller checks
}

struct SpmmArgs {
    const int32_t *indptr; const int32_t *indices; const float *vals;
    const void *X; void *Y;
    const float *Gs; const uint32_t *bitmap;
    float *P; float *M; float *V;
    int64_t n_rows;
    float step_size, bc2_sqrt, w1, beta2, omb2, eps;
    int remap;
};

enum { M_SPARSE = 1, M_ADDG = 2, M_ADAM = 4 };

// Y = [Gs +] A_hat X   (4 rows per 256-thread workgroup, one wave per row)
//   M_SPARSE: X is Gs (fp32) whose non-zero rows are flagged in `bitmap`
//   M_ADDG  : epilogue adds Gs[row] where flagged           (Horner term)
//   M_ADAM  : epilogue applies torch.optim.Adam to P/M/V with grad = result
template <int D, typename TI, typename TO, int MODE>
__global__ void __launch_bounds__(256) k_spmm(SpmmArgs a) {
    constexpr int LPR = D / 4;
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int64_t ntiles = (a.n_rows + 3) >> 2;
    const int64_t tile = tile_of_block(blockIdx.x, ntiles, a.remap);
    if (tile >= ntiles) return;
    const int64_t row = tile * 4 + wid;
    if (row >= a.n_rows) return;
    const int start = a.indptr[row], end = a.indptr[row + 1];
    f32x4 acc = row_gather<D, TI, (MODE & M_SPARSE) != 0>(a.indices, a.vals, start, end,
                                                         (const TI *)a.X, a.bitmap, lane);
    if (lane >= LPR) return;
    const int64_t off = row * D + lane * 4;
    if ((MODE & M_ADDG) && bit_set(a.bitmap, (int)row)) {
        f32x4 g = load4(a.Gs + off);
        acc = g + acc;
    }
    if (MODE & M_ADAM) {
        f32x4 p = load4(a.P + off), m = load4(a.M + off), v = load4(a.V + off);
        m = m + a.w1 * (acc - m);                       // exp_avg.lerp_(grad, 1-beta1)
        v = v * a.beta2 + (a.omb2 * acc) * acc;         // mul_(beta2).addcmul_(g,g,1-beta2)
        f32x4 denom;
        denom.x = sqrtf(v.x) / a.bc2_sqrt + a.eps; denom.y = sqrtf(v.y) / a.bc2_sqrt + a.eps;
        denom.z = sqrtf(v.z) / a.bc2_sqrt + a.eps; denom.w = sqrtf(v.w) / a.bc2_sqrt + a.eps;
        p = p - a.step_size * (m / denom);              // addcdiv_(exp_avg, denom, -step_size)
        store4(a.P + off, p); store4(a.M + off, m); store4(a.V + off, v);
    } else {
        store4((TO *)a.Y + off, acc);
    }
}

// out = (X_0 + X_1 + ... + X_{K-1} + A X_{K-1}) / (K+1)   -- last layer of computer()
struct MeanArgs {
    const int32_t *indptr; const int32_t *indices; const float *vals;
    const float *X0; const void *Xl[LGCN_MAX_LAYERS]; int K;
    float *out; int64_t n_rows; int remap;
};

template <int D, typename TI>
__global__ void __launch_bounds__(256) k_spmm_mean(MeanArgs a) {
    constexpr int LPR = D / 4;
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int64_t ntiles = (a.n_rows + 3) >> 2;
    const int64_t tile = tile_of_block(blockIdx.x, ntiles, a.remap);
    if (tile >= ntiles) return;
    const int64_t row = tile * 4 + wid;
    if (row >= a.n_rows) return;
    const int start = a.indptr[row], end = a.indptr[row + 1];
    f32x4 xk;
    if (a.K == 1) xk = row_gather<D, float, false>(a.indices, a.vals, start, end, a.X0, nullptr, lane);
    else xk = row_gather<D, TI, false>(a.indices, a.vals, start, end, (const TI *)a.Xl[a.K - 1], nullptr, lane);
    if (lane >= LPR) return;
    const int64_t off = row * D + lane * 4;
    f32x4 s = load4(a.X0 + off);
    for (int k = 1; k < a.K; k++) s += load4((const TI *)a.Xl[k] + off);
    s += xk;
    const float div = (float)(a.K + 1);
    store4(a.out + off, s / div);
}

// ---------------------------------------------------------------------------------
// Fused BPR: one 192-thread workgroup (3 waves) per triplet; wave c owns slot c
// (0 user, 1 positive item, 2 negative item).
//   1. e_c = mean_k X_k[row_c]; X_K[row_c] is computed on the fly from X_{K-1}
//   2. x = e_u.e_p - e_u.e_n ; l = logsigmoid(x) ; r = |e_u|^2+|e_p|^2+|e_n|^2
//   3. gradient row of slot c wrt the propagated table -> either
//      fixed-point atomics into G64 (single GPU) or the exchange buffer (DP).
// ---------------------------------------------------------------------------------
struct BprArgs {
    const int32_t *indptr; const int32_t *indices; const float *vals;
    const float *X0; const void *Xl[LGCN_MAX_LAYERS]; int K;
    int32_t n_users; int64_t N;
    const int32_t *users; const int32_t *pos; const int32_t *neg;   // already offset to the local shard
    int32_t B_local;      // triplets handled by this launch
    int32_t shard;        // row stride of the contrib block (>= B_local)
    float inv_B;          // 1 / global batch
    float lam;            // decay / global batch
    long long *G64;       // if non-null: atomics
    float *contrib;       // else: [3*shard*D | shard | shard]
    float *terms;         // single GPU: [2*B]  (loss terms, reg terms)
    int32_t *err;
};

__device__ __forceinline__ float logsigmoid_f(float x) { return fminf(x, 0.f) - log1pf(expf(-fabsf(x))); }
__device__ __forceinline__ float sigmoid_neg_f(float x) {
    const float z = expf(-fabsf(x));
    return x < 0.f ? 1.f / (1.f + z) : z / (1.f + z);
}

template <int D, typename TI>
__global__ void __launch_bounds__(192) k_bpr(BprArgs a) {
    constexpr int LPR = D / 4;
    __shared__ __attribute__((aligned(16))) float e_lds[3 * D];
    const int lane = threadIdx.x & 63, c = threadIdx.x >> 6;
    const int b = blockIdx.x;
    int64_t row;
    bool bad = false;
    if (c == 0) { int u = a.users[b]; bad = (u < 0 || u >= a.n_users); row = u; }
    else {
        int it = (c == 1) ? a.pos[b] : a.neg[b];
        bad = (it < 0 || (int64_t)it + a.n_users >= a.N); row = (int64_t)it + a.n_users;
    }
    if (bad) { if (lane == 0) atomicExch(a.err, 1); row = 0; }
    const int start = a.indptr[row], end = a.indptr[row + 1];
    f32x4 xk;
    if (a.K == 1) xk = row_gather<D, float, false>(a.indices, a.vals, start, end, a.X0, nullptr, lane);
    else xk = row_gather<D, TI, false>(a.indices, a.vals, start, end, (const TI *)a.Xl[a.K - 1], nullptr, lane);
    if (lane < LPR) {
        const int64_t off = row * D + lane * 4;
        f32x4 s = load4(a.X0 + off);
        for (int k = 1; k < a.K; k++) s += load4((const TI *)a.Xl[k] + off);
        s += xk;
        const float div = (float)(a.K + 1);
        store4(&e_lds[c * D + lane * 4], s / div);
    }
    __syncthreads();
    // every wave recomputes the (cheap) dots from LDS; lanes >= LPR contribute zeros
    f32x4 u4 = {0, 0, 0, 0}, p4 = u4, n4 = u4;
    if (lane < LPR) {
        u4 = load4(&e_lds[lane * 4]); p4 = load4(&e_lds[D + lane * 4]); n4 = load4(&e_lds[2 * D + lane * 4]);
    }
    float ps = u4.x * p4.x + u4.y * p4.y + u4.z * p4.z + u4.w * p4.w;
    float ns = u4.x * n4.x + u4.y * n4.y + u4.z * n4.z + u4.w * n4.w;
    float rr = (u4.x * u4.x + u4.y * u4.y + u4.z * u4.z + u4.w * u4.w) +
               (p4.x * p4.x + p4.y * p4.y + p4.z * p4.z + p4.w * p4.w) +
               (n4.x * n4.x + n4.y * n4.y + n4.z * n4.z + n4.w * n4.w);
#pragma unroll
    for (int off = 1; off < LPR; off <<= 1) {
        ps += __shfl_xor(ps, off); ns += __shfl_xor(ns, off); rr += __shfl_xor(rr, off);
    }
    const float x = ps - ns;
    const float gb = bad ? 0.f : -a.inv_B * sigmoid_neg_f(x);
    if (c == 0 && lane == 0) {
        float *lt = a.G64 ? a.terms : a.contrib + (int64_t)3 * a.shard * D;
        const int stride = a.G64 ? a.B_local : a.shard;
        lt[b] = bad ? 0.f : logsigmoid_f(x);
        lt[stride + b] = bad ? 0.f : rr;
    }
    if (lane < LPR) {
        f32x4 g;
        if (c == 0) g = gb * (p4 - n4) + a.lam * u4;
        else if (c == 1) g = gb * u4 + a.lam * p4;
        else g = (-gb) * u4 + a.lam * n4;
        if (bad) g = f32x4{0, 0, 0, 0};
        if (a.G64) {
            unsigned long long *dst = (unsigned long long *)(a.G64 + row * D + lane * 4);
            atomicAdd(dst + 0, (unsigned long long)__double2ll_rn((double)g.x * FIXED_SCALE));
            atomicAdd(dst + 1, (unsigned long long)__double2ll_rn((double)g.y * FIXED_SCALE));
            atomicAdd(dst + 2, (unsigned long long)__double2ll_rn((double)g.z * FIXED_SCALE));
            atomicAdd(dst + 3, (unsigned long long)__double2ll_rn((double)g.w * FIXED_SCALE));
        } else {
            store4(a.contrib + ((int64_t)c * a.shard + b) * D + lane * 4, g);
        }
    }
}

// slot -> destination row of the global batch
__device__ __forceinline__ int64_t slot_row(int c, int b, const int32_t *users, const int32_t *pos,
                                            const int32_t *neg, int32_t n_users, int64_t N) {
    int64_t r;
    if (c == 0) { int u = users[b]; r = (u < 0 || u >= n_users) ? -1 : u; }
    else { int it = (c == 1) ? pos[b] : neg[b]; r = (it < 0 || (int64_t)it + n_users >= N) ? -1 : (int64_t)it + n_users; }
    return r;
}

struct SlotArgs {
    const int32_t *users; const int32_t *pos; const int32_t *neg;
    int32_t B; int32_t n_users; int64_t N;
    long long *G64; float *Gs; uint32_t *bitmap;
    const float *gathered; int32_t shard; int32_t world;   // DP scatter
    const float *terms; float *loss_out; float decay; int K;
};

// DP: order-independent scatter of every rank's gradient rows into G64
template <int D>
__global__ void __launch_bounds__(256) k_scatter(SlotArgs a) {
    constexpr int LPR = D / 4, SPB = 256 / LPR;
    const int s = blockIdx.x * SPB + threadIdx.x / LPR, l = threadIdx.x % LPR;
    if (s >= 3 * a.B) return;
    const int c = s / a.B, b = s % a.B;
    const int64_t row = slot_row(c, b, a.users, a.pos, a.neg, a.n_users, a.N);
    if (row < 0) return;
    const int r = b / a.shard, i = b % a.shard;
    const int64_t blk = (int64_t)3 * a.shard * D + 2 * a.shard;
    f32x4 g = load4(a.gathered + r * blk + ((int64_t)c * a.shard + i) * D + l * 4);
    unsigned long long *dst = (unsigned long long *)(a.G64 + row * D + l * 4);
    atomicAdd(dst + 0, (unsigned long long)__double2ll_rn((double)g.x * FIXED_SCALE));
    atomicAdd(dst + 1, (unsigned long long)__double2ll_rn((double)g.y * FIXED_SCALE));
    atomicAdd(dst + 2, (unsigned long long)__double2ll_rn((double)g.z * FIXED_SCALE));
    atomicAdd(dst + 3, (unsigned long long)__double2ll_rn((double)g.w * FIXED_SCALE));
}

// Gs[row] = float(G64[row]) / (K+1), flag the row; block 0 also reduces the loss.
// Slots that share a row write identical bits (benign).
template <int D>
__global__ void __launch_bounds__(256) k_finalize(SlotArgs a) {
    constexpr int LPR = D / 4, SPB = 256 / LPR;
    const int s = blockIdx.x * SPB + threadIdx.x / LPR, l = threadIdx.x % LPR;
    if (s < 3 * a.B) {
        const int c = s / a.B, b = s % a.B;
        const int64_t row = slot_row(c, b, a.users, a.pos, a.neg, a.n_users, a.N);
        if (row >= 0) {
            const long long *src = a.G64 + row * D + l * 4;
            const float div = (float)(a.K + 1);
            f32x4 g;
            g.x = (float)((double)src[0] * FIXED_INV) / div; g.y = (float)((double)src[1] * FIXED_INV) / div;
            g.z = (float)((double)src[2] * FIXED_INV) / div; g.w = (float)((double)src[3] * FIXED_INV) / div;
            store4(a.Gs + row * D + l * 4, g);
            if (l == 0) atomicOr(a.bitmap + (row >> 5), 1u << (row & 31));
        }
    }
    if (blockIdx.x == 0) {   // deterministic loss reduction: fixed strided partials + LDS tree
        __shared__ float sl[256], sr[256];
        float fl = 0.f, fr = 0.f;
        if (a.gathered) {
            const int64_t blk = (int64_t)3 * a.shard * D + 2 * a.shard;
            for (int b = threadIdx.x; b < a.B; b += 256) {
                const float *t = a.gathered + (b / a.shard) * blk + (int64_t)3 * a.shard * D;
                fl += t[b % a.shard]; fr += t[a.shard + b % a.shard];
            }
        } else {
            for (int b = threadIdx.x; b < a.B; b += 256) { fl += a.terms[b]; fr += a.terms[a.B + b]; }
        }
        sl[threadIdx.x] = fl; sr[threadIdx.x] = fr;
        __syncthreads();
        for (int w = 128; w > 0; w >>= 1) {
            if ((int)threadIdx.x < w) { sl[threadIdx.x] += sl[threadIdx.x + w]; sr[threadIdx.x] += sr[threadIdx.x + w]; }
            __syncthreads();
        }
        if (threadIdx.x == 0) {
            const float bpr = -(sl[0] / (float)a.B);
            const float reg = (0.5f * sr[0]) / (float)a.B;
            a.loss_out[0] = bpr + a.decay * reg; a.loss_out[1] = bpr; a.loss_out[2] = reg;
        }
    }
}

// zero what the step touched: G64 / Gs rows and bitmap words of the batch rows
template <int D>
__global__ void __launch_bounds__(256) k_cleanup(SlotArgs a) {
    constexpr int LPR = D / 4, SPB = 256 / LPR;
    const int s = blockIdx.x * SPB + threadIdx.x / LPR, l = threadIdx.x % LPR;
    if (s >= 3 * a.B) return;
    const int c = s / a.B, b = s % a.B;
    const int64_t row = slot_row(c, b, a.users, a.pos, a.neg, a.n_users, a.N);
    if (row < 0) return;
    long long *q = a.G64 + row * D + l * 4;
    q[0] = 0; q[1] = 0; q[2] = 0; q[3] = 0;
    store4(a.Gs + row * D + l * 4, f32x4{0, 0, 0, 0});
    if (l == 0) a.bitmap[row >> 5] = 0u;
}

__global__ void __launch_bounds__(256) k_apply_perm(const int32_t *S, int cols, const int64_t *perm, int64_t T,
                                                   int32_t *users, int32_t *pos, int32_t *neg) {
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= T) return;
    const int64_t j = perm ? perm[t] : t;
    users[t] = S[j * cols]; pos[t] = S[j * cols + 1]; neg[t] = S[j * cols + 2];
}

// ---------------------------------------------------------------------------------
// launch helpers
// ---------------------------------------------------------------------------------
static inline unsigned grid_rows(int64_t n_rows, int remap) {
    int64_t ntiles = (n_rows + 3) / 4;
    if (remap) ntiles = ((ntiles + 7) / 8) * 8;
    return (unsigned)ntiles;
}

template <int D, typename TI, typename TO, int MODE>
static void launch_spmm_t(const SpmmArgs &a, hipStream_t st) {
    hipLaunchKernelGGL((k_spmm<D, TI, TO, MODE>), dim3(grid_rows(a.n_rows, a.remap)), dim3(256), 0, st, a);
}

template <int D, int MODE>
static int launch_spmm_d(const SpmmArgs &a, int x_dtype, int y_dtype, hipStream_t st) {
    if (MODE & M_SPARSE) x_dtype = LGCN_F32;           // Gs is always fp32
    if (MODE & M_ADAM) y_dtype = LGCN_F32;
    if (x_dtype == LGCN_F32 && y_dtype == LGCN_F32) launch_spmm_t<D, float, float, MODE>(a, st);
    else if (x_dtype == LGCN_F32 && y_dtype == LGCN_BF16) launch_spmm_t<D, float, bf16_t, MODE>(a, st);
    else if (x_dtype == LGCN_BF16 && y_dtype == LGCN_F32) launch_spmm_t<D, bf16_t, float, MODE>(a, st);
    else launch_spmm_t<D, bf16_t, bf16_t, MODE>(a, st);
    return 0;
}

template <int MODE>
static int launch_spmm(const SpmmArgs &a, int d, int x_dtype, int y_dtype, hipStream_t st) {
    switch (d) {
    case 32: return launch_spmm_d<32, MODE>(a, x_dtype, y_dtype, st);
    case 64: return launch_spmm_d<64, MODE>(a, x_dtype, y_dtype, st);
    case 128: return launch_spmm_d<128, MODE>(a, x_dtype, y_dtype, st);
    case 256: return launch_spmm_d<256, MODE>(a, x_dtype, y_dtype, st);
    }
    lgcn_set_error("embedding dim must be 32, 64, 128 or 256");
    return 3;
}

#define DISPATCH_D(d, CALL)                                                      \
    switch (d) {                                                                 \
    case 32: { constexpr int D = 32; CALL; } break;                              \
    case 64: { constexpr int D = 64; CALL; } break;                              \
    case 128: { constexpr int D = 128; CALL; } break;                            \
    case 256: { constexpr int D = 256; CALL; } break;                            \
    default: lgcn_set_error("embedding dim must be 32, 64, 128 or 256"); return 3; \
    }

static int check_dtype(int t) {
    if (t != LGCN_F32 && t != LGCN_BF16) { lgcn_set_error("dtype must be LGCN_F32 or LGCN_BF16"); return 3; }
    return 0;
}

extern "C" int lgcn_device_available(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) { (void)hipGetLastError(); return 0; }
    return n > 0;
}

extern "C" int lgcn_spmm_csr(const int32_t *indptr, const int32_t *indices, const float *vals, int64_t n_rows,
                             const void *X, int x_dtype, void *Y, int y_dtype, int d, void *stream) {
    if (!indptr || !indices || !vals || !X || !Y || n_rows < 0) { lgcn_set_error("lgcn_spmm_csr: null/invalid argument"); return 3; }
    if (check_dtype(x_dtype) || check_dtype(y_dtype)) return 3;
    if (n_rows == 0) return 0;
    SpmmArgs a{};
    a.indptr = indptr; a.indices = indices; a.vals = vals; a.X = X; a.Y = Y; a.n_rows = n_rows; a.remap = 0;
    int rc = launch_spmm<0>(a, d, x_dtype, y_dtype, (hipStream_t)stream);
    if (rc) return rc;
    HIP_OK(hipGetLastError());
    return 0;
}

static inline size_t esize(int dtype) { return dtype == LGCN_BF16 ? 2 : 4; }

extern "C" int lgcn_propagate_mean(const int32_t *indptr, const int32_t *indices, const float *vals, int64_t N,
                                   const float *E0, int K, int d, int act_dtype, void *work, float *out,
                                   void *stream) {
    if (!indptr || !indices || !vals || !E0 || !out || K < 1 || K > LGCN_MAX_LAYERS || N <= 0) {
        lgcn_set_error("lgcn_propagate_mean: invalid argument"); return 3;
    }
    if (K > 1 && !work) { lgcn_set_error("lgcn_propagate_mean: workspace required for K > 1"); return 3; }
    if (check_dtype(act_dtype)) return 3;
    hipStream_t st = (hipStream_t)stream;
    MeanArgs m{};
    m.indptr = indptr; m.indices = indices; m.vals = vals; m.X0 = E0; m.K = K; m.out = out; m.n_rows = N; m.remap = 0;
    const size_t stride = (size_t)N * d * esize(act_dtype);
    const void *prev = E0; int prev_dtype = LGCN_F32;
    for (int k = 1; k < K; k++) {
        void *y = (char *)work + (size_t)(k - 1) * stride;
        SpmmArgs a{};
        a.indptr = indptr; a.indices = indices; a.vals = vals; a.X = prev; a.Y = y; a.n_rows = N; a.remap = 0;
        int rc = launch_spmm<0>(a, d, prev_dtype, act_dtype, st);
        if (rc) return rc;
        m.Xl[k] = y; prev = y; prev_dtype = act_dtype;
    }
    DISPATCH_D(d, {
        if (act_dtype == LGCN_F32) hipLaunchKernelGGL((k_spmm_mean<D, float>), dim3(grid_rows(N, 0)), dim3(256), 0, st, m);
        else hipLaunchKernelGGL((k_spmm_mean<D, bf16_t>), dim3(grid_rows(N, 0)), dim3(256), 0, st, m);
    });
    HIP_OK(hipGetLastError());
    return 0;
}

extern "C" int lgcn_apply_perm(const int32_t *S, int s_cols, const int64_t *perm, int64_t T,
                               int32_t *users, int32_t *pos, int32_t *neg, void *stream) {
    if (!S || !users || !pos || !neg || s_cols < 3 || T < 0) { lgcn_set_error("lgcn_apply_perm: invalid argument"); return 3; }
    if (T == 0) return 0;
    hipLaunchKernelGGL(k_apply_perm, dim3((unsigned)((T + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       S, s_cols, perm, T, users, pos, neg);
    HIP_OK(hipGetLastError());
    return 0;
}

// ---------------------------------------------------------------------------------
// training context
// ---------------------------------------------------------------------------------
struct lgcn_ctx {
    lgcn_train_config c;
    int64_t step;
    void *act[LGCN_MAX_LAYERS];   // act[k] = X_k storage for k = 1..K-1 (also reused for H)
};

extern "C" int lgcn_ctx_create(const lgcn_train_config *cfg, lgcn_ctx **out) {
    if (!cfg || !out) { lgcn_set_error("lgcn_ctx_create: null argument"); return 3; }
    const lgcn_train_config &c = *cfg;
    if (!c.indptr || !c.indices || !c.vals || !c.E0 || !c.adam_m || !c.adam_v || !c.G64 || !c.Gs ||
        !c.bitmap || !c.terms || !c.err) { lgcn_set_error("lgcn_ctx_create: null buffer"); return 3; }
    if (c.K < 1 || c.K > LGCN_MAX_LAYERS) { lgcn_set_error("lgcn_ctx_create: K out of range"); return 3; }
    if (c.K > 1 && !c.act) { lgcn_set_error("lgcn_ctx_create: activation workspace missing"); return 3; }
    if (c.d != 32 && c.d != 64 && c.d != 128 && c.d != 256) { lgcn_set_error("embedding dim must be 32, 64, 128 or 256"); return 3; }
    if (check_dtype(c.act_dtype)) return 3;
    if (c.N <= 0 || c.n_users <= 0 || c.n_users >= c.N || c.max_batch <= 0) { lgcn_set_error("lgcn_ctx_create: bad sizes"); return 3; }
    lgcn_ctx *x = new (std::nothrow) lgcn_ctx;
    if (!x) { lgcn_set_error("out of memory"); return 4; }
    x->c = c; x->step = 0;
    const size_t stride = (size_t)c.N * c.d * esize(c.act_dtype);
    for (int k = 0; k < LGCN_MAX_LAYERS; k++) x->act[k] = nullptr;
    for (int k = 1; k < c.K; k++) x->act[k] = (char *)c.act + (size_t)(k - 1) * stride;
    *out = x;
    return 0;
}
extern "C" void lgcn_ctx_destroy(lgcn_ctx *ctx) { delete ctx; }
extern "C" int64_t lgcn_ctx_get_step(const lgcn_ctx *ctx) { return ctx ? ctx->step : -1; }
extern "C" void lgcn_ctx_set_step(lgcn_ctx *ctx, int64_t s) { if (ctx) ctx->step = s; }
extern "C" void lgcn_ctx_set_lr(lgcn_ctx *ctx, double lr) { if (ctx) ctx->c.lr = lr; }

static SpmmArgs base_spmm(const lgcn_ctx *x) {
    SpmmArgs a{};
    a.indptr = x->c.indptr; a.indices = x->c.indices; a.vals = x->c.vals; a.n_rows = x->c.N;
    a.Gs = x->c.Gs; a.bitmap = x->c.bitmap; a.remap = x->c.xcd_remap;
    return a;
}

// forward layers X_1..X_{K-1}
static int run_forward(lgcn_ctx *x, hipStream_t st) {
    const lgcn_train_config &c = x->c;
    const void *prev = c.E0; int prev_dt = LGCN_F32;
    for (int k = 1; k < c.K; k++) {
        SpmmArgs a = base_spmm(x);
        a.X = prev; a.Y = x->act[k];
        int rc = launch_spmm<0>(a, c.d, prev_dt, c.act_dtype, st);
        if (rc) return rc;
        prev = x->act[k]; prev_dt = c.act_dtype;
    }
    return 0;
}

static int run_bpr(lgcn_ctx *x, const int32_t *users, const int32_t *pos, const int32_t *neg,
                   int32_t B_global, int32_t b_off, int32_t B_local, int32_t shard, bool atomics, hipStream_t st) {
    const lgcn_train_config &c = x->c;
    BprArgs a{};
    a.indptr = c.indptr; a.indices = c.indices; a.vals = c.vals; a.X0 = c.E0; a.K = c.K;
    for (int k = 1; k < c.K; k++) a.Xl[k] = x->act[k];
    a.n_users = c.n_users; a.N = c.N;
    a.users = users + b_off; a.pos = pos + b_off; a.neg = neg + b_off;
    a.B_local = B_local; a.shard = shard;
    a.inv_B = 1.0f / (float)B_global; a.lam = c.decay / (float)B_global;
    a.G64 = atomics ? (long long *)c.G64 : nullptr; a.contrib = c.contrib; a.terms = c.terms; a.err = c.err;
    if (B_local <= 0) return 0;
    DISPATCH_D(c.d, {
        if (c.act_dtype == LGCN_F32) hipLaunchKernelGGL((k_bpr<D, float>), dim3(B_local), dim3(192), 0, st, a);
        else hipLaunchKernelGGL((k_bpr<D, bf16_t>), dim3(B_local), dim3(192), 0, st, a);
    });
    return 0;
}

// finalize + backward chain + Adam + cleanup
static int run_backward(lgcn_ctx *x, const int32_t *users, const int32_t *pos, const int32_t *neg, int32_t B,
                        const float *gathered, int32_t shard, int32_t world, float *loss_out, hipStream_t st) {
    const lgcn_train_config &c = x->c;
    SlotArgs s{};
    s.users = users; s.pos = pos; s.neg = neg; s.B = B; s.n_users = c.n_users; s.N = c.N;
    s.G64 = (long long *)c.G64; s.Gs = c.Gs; s.bitmap = c.bitmap; s.gathered = gathered; s.shard = shard; s.world = world;
    s.terms = c.terms; s.loss_out = loss_out; s.decay = c.decay; s.K = c.K;
    const int spb = 256 / (c.d / 4);
    const unsigned sgrid = (unsigned)((3 * (int64_t)B + spb - 1) / spb);
    if (gathered) { DISPATCH_D(c.d, hipLaunchKernelGGL((k_scatter<D>), dim3(sgrid), dim3(256), 0, st, s)); }
    DISPATCH_D(c.d, hipLaunchKernelGGL((k_finalize<D>), dim3(sgrid), dim3(256), 0, st, s));

    x->step += 1;
    const double bc1 = 1.0 - pow(c.beta1, (double)x->step);
    const double bc2 = 1.0 - pow(c.beta2, (double)x->step);
    // Horner: h_{K-1} = Gs + A Gs (sparse input); h_{k-1} = Gs + A h_k; last one feeds Adam
    const void *prev = c.Gs; int prev_dt = LGCN_F32;
    for (int k = c.K; k >= 1; k--) {
        SpmmArgs a = base_spmm(x);
        a.X = prev;
        const bool first = (k == c.K), last = (k == 1);
        void *y = nullptr;
        if (!last) {
            // ping-pong inside the activation workspace (forward activations are dead now)
            y = x->act[1 + ((c.K - k) & 1)];
            if (c.K == 2) y = x->act[1];
            a.Y = y;
        } else {
            a.P = c.E0; a.M = c.adam_m; a.V = c.adam_v;
            a.step_size = (float)(c.lr / bc1); a.bc2_sqrt = (float)sqrt(bc2);
            a.w1 = (float)(1.0 - c.beta1); a.beta2 = (float)c.beta2; a.omb2 = (float)(1.0 - c.beta2); a.eps = (float)c.eps;
        }
        int rc;
        if (first && last) rc = launch_spmm<M_SPARSE | M_ADDG | M_ADAM>(a, c.d, prev_dt, LGCN_F32, st);
        else if (first) rc = launch_spmm<M_SPARSE | M_ADDG>(a, c.d, prev_dt, c.act_dtype, st);
        else if (last) rc = launch_spmm<M_ADDG | M_ADAM>(a, c.d, prev_dt, LGCN_F32, st);
        else rc = launch_spmm<M_ADDG>(a, c.d, prev_dt, c.act_dtype, st);
        if (rc) return rc;
        prev = y; prev_dt = c.act_dtype;
    }
    DISPATCH_D(c.d, hipLaunchKernelGGL((k_cleanup<D>), dim3(sgrid), dim3(256), 0, st, s));
    return 0;
}

static int check_batch(const lgcn_ctx *x, const void *u, const void *p, const void *n, int32_t B) {
    if (!x || !u || !p || !n) { lgcn_set_error("train step: null argument"); return 3; }
    if (B <= 0 || B > x->c.max_batch) { lgcn_set_error("train step: batch size out of range (0 < B <= max_batch)"); return 3; }
    return 0;
}

extern "C" int lgcn_train_step(lgcn_ctx *x, const int32_t *users, const int32_t *pos, const int32_t *neg,
                               int32_t B, float *loss_out, void *stream) {
    int rc = check_batch(x, users, pos, neg, B);
    if (rc) return rc;
    if (!loss_out) { lgcn_set_error("train step: loss_out is null"); return 3; }
    hipStream_t st = (hipStream_t)stream;
    if ((rc = run_forward(x, st))) return rc;
    if ((rc = run_bpr(x, users, pos, neg, B, 0, B, B, true, st))) return rc;
    if ((rc = run_backward(x, users, pos, neg, B, nullptr, B, 1, loss_out, st))) return rc;
    HIP_OK(hipGetLastError());
    return 0;
}

extern "C" int lgcn_train_epoch(lgcn_ctx *x, const int32_t *users, const int32_t *pos, const int32_t *neg,
                                int64_t T, int32_t B, float *loss_out, void *stream) {
    if (T <= 0) return 0;
    int64_t i = 0;
    for (int64_t t = 0; t < T; t += B, i++) {
        const int32_t b = (int32_t)((T - t) < B ? (T - t) : B);
        int rc = lgcn_train_step(x, users + t, pos + t, neg + t, b, loss_out + 3 * i, stream);
        if (rc) return rc;
    }
    return 0;
}

extern "C" int lgcn_train_step_dp_part1(lgcn_ctx *x, const int32_t *users, const int32_t *pos, const int32_t *neg,
                                        int32_t B_global, int32_t world, int32_t rank, void *stream) {
    int rc = check_batch(x, users, pos, neg, B_global);
    if (rc) return rc;
    if (!x->c.contrib) { lgcn_set_error("dp step: cfg.contrib exchange buffer missing"); return 3; }
    if (world < 1 || rank < 0 || rank >= world) { lgcn_set_error("dp step: bad world/rank"); return 3; }
    const int32_t shard = (B_global + world - 1) / world;
    const int32_t b_off = rank * shard;
    int32_t B_local = B_global - b_off;
    if (B_local > shard) B_local = shard;
    if (B_local < 0) B_local = 0;
    hipStream_t st = (hipStream_t)stream;
    if ((rc = run_forward(x, st))) return rc;
    if ((rc = run_bpr(x, users, pos, neg, B_global, b_off, B_local, shard, false, st))) return rc;
    HIP_OK(hipGetLastError());
    return 0;
}

extern "C" int lgcn_train_step_dp_part2(lgcn_ctx *x, const int32_t *users, const int32_t *pos, const int32_t *neg,
                                        int32_t B_global, int32_t world, const float *gathered, float *loss_out,
                                        void *stream) {
    int rc = check_batch(x, users, pos, neg, B_global);
    if (rc) return rc;
    if (!gathered || !loss_out || world < 1) { lgcn_set_error("dp step part 2: invalid argument"); return 3; }
    const int32_t shard = (B_global + world - 1) / world;
    if ((rc = run_backward(x, users, pos, neg, B_global, gathered, shard, world, loss_out, (hipStream_t)stream))) return rc;
    HIP_OK(hipGetLastError());
    return 0;
}

extern "C" int lgcn_ctx_check(lgcn_ctx *x, void *stream) {
    if (!x) { lgcn_set_error("null context"); return 3; }
    int32_t flag = 0;
    HIP_OK(hipMemcpyAsync(&flag, x->c.err, sizeof flag, hipMemcpyDeviceToHost, (hipStream_t)stream));
    HIP_OK(hipStreamSynchronize((hipStream_t)stream));
    if (flag) {
        HIP_OK(hipMemsetAsync(x->c.err, 0, sizeof flag, (hipStream_t)stream));
        lgcn_set_error("device flagged an out-of-range user/item id in a batch");
    }
    return flag;
}
