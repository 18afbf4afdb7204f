// lgcn_eval.hip -- fused full-ranking evaluation for gfx950.
//
// Replaces, for one launch over all test users (reference: LightGCN_work/code):
//   model.py:114-123      getUsersRating: rating = U_b . I^T  (the reference re-propagates per 100 users;
//                         here the propagated table comes in once from lgcn_propagate_mean)
//   Procedure.py:177-181  rating[train positives] = -(1<<10)
//   Procedure.py:183      torch.topk(rating, k=max(topks))
//   Procedure.py:89-121 + utils.py:173-217   per-user hits, precision / recall / NDCG, summed over users
// without ever materialising the [users, m_items] score matrix.
//
// k_eval_topk: the one dense contraction of the system -> matrix cores.  fp32 in / fp32 accumulate
// (v_mfma_f32_32x32x2_f32: exact f32, 64 FLOP/clk/SIMD) so the ranking is the reference's fp32 ranking.
//   * workgroup = 4 waves x 32 users; the users' rows are the B operand and stay in registers for the
//     whole item sweep (D/2 VGPRs per lane: lane (j, h) holds elements [h*D/2, (h+1)*D/2) of user j --
//     the k order inside a dot product is free, so each half-row is one contiguous run);
//   * items stream through LDS in tiles of 32 rows (coalesced 16-byte global loads, double buffered,
//     rows padded by 16 B so the ds_read_b128 of 32 lanes at the same column hit different banks);
//     every wave multiplies the same item tile with its own users: D/2 MFMAs per 32 x 32 scores;
//   * the accumulator layout puts a user on a lane (col = lane & 31) and 16 of the tile's items in its
//     registers, so the train-positive mask is a 32-bit word per user (built one tile ahead from the
//     sorted train CSR by a cursor whose next entry is always already loaded) and the running top-K
//     is per lane: a threshold compare per score marks the candidates of a tile, and only those touch the
//     lane's list in LDS (replace the minimum, rescan for the new minimum), one per round for all lanes
//     together.  The lists of the two lanes of a user are merged and sorted once at the end.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <math.h>

#include "lgcn_hip.h"
#include "lgcn_internal.h"

typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

#define EVAL_KMAX 64
#define EVAL_NEG_INF (-3.0e38f)
#ifndef EVAL_PARTS
#define EVAL_PARTS 3          /* workgroups per user block (<= 4) = workgroups that fit a CU with the compact lists */
#endif
// Ids in the per-lane lists: int32 item ids, or (compact form) 16-bit offsets from the first item of the workgroup's
// part of the sweep -- 10 KB less LDS per workgroup, which is what lets a third workgroup share the CU.
template <typename IDT> struct ListId;
template <> struct ListId<int32_t> { static constexpr int32_t EMPTY = -1; static constexpr int PAD = 1; };
template <> struct ListId<uint16_t> { static constexpr uint16_t EMPTY = 0xffffu; static constexpr int PAD = 0; };
#ifndef EVAL_NBUF
#define EVAL_NBUF 2
#endif
#ifndef EVAL_PUB_TILES
#define EVAL_PUB_TILES 16        /* tiles between two exchanges of the parts' K-th best (a power of two; 0: never) */
#endif
#ifndef EVAL_SPLIT3
#define EVAL_SPLIT3 1          /* compact form: fp32 scores from six bf16 product planes (0: fp32 matrix instructions) */
#endif
#define EVAL_ID16_MAX_TILES 2047        /* 2047 * 32 + 31 < 0xffff */

struct EvalArgs {
    const float *E; int32_t n_users, m_items;
    const int32_t *users; int32_t n_eval;
    const int64_t *train_ptr; const int32_t *train_idx;
    int32_t K;
    int32_t *out_items; float *out_scores;
    // the item sweep split over gridDim.y workgroups per user block (two co-resident workgroups per CU overlap
    // one's list maintenance with the other's MFMAs): each writes its sorted partial list here, k_eval_merge picks
    int32_t *part_items; float *part_scores;      // [n_eval, gridDim.y, K] or NULL (gridDim.y == 1: straight to out_*)
    // The parts of a user's sweep tell each other their K-th best so far ([gridDim.y][n_eval], NaN = nothing yet; or NULL):
    // K items at or above a part's K-th best exist, so every other part may use it as a floor for what it still inserts.
    // Any value ever published is valid, however stale -- the exchange only prunes, it never decides.
    float *thr_pub;
    // Optional: the train-positive masks precomputed once per dataset (lgcn_eval_build_masks): masks[t * mask_stride + slot] =
    // bit i set when item 32 t + i is a train positive of the user in evaluation slot `slot`.  With them the item sweep holds no
    // global load besides the tile stream and one coalesced 128-byte mask load per wave and tile, issued a tile ahead (the
    // cursor over the sorted train CSR costs a dependent load and a wait inside the loop).  NULL: the cursor.
    const uint32_t *masks; int64_t mask_stride;
};

// c ? hi : lo.  The empty asm keeps the compiler from folding a tree of these over vector elements into ONE dynamically
// indexed element, which it lowers through scratch memory or an LDS copy of the vector.
static __device__ __forceinline__ float pick(bool c, float hi, float lo) {
    asm volatile("" : "+v"(hi), "+v"(lo));
    return c ? hi : lo;
}

// SPLIT3: the fp32 product on the bf16 matrix cores.  Every fp32 value is EXACTLY the sum of three bf16 values
// (x = h + m + l: 8 + 8 + 8 significant bits, by truncation: h = x & 0xffff0000, m = (x - h) & 0xffff0000, l = x - h - m,
// every subtraction exact), and every bf16 x bf16 product is exact in fp32, so
//     a . b = sum_k (ah + am + al)(bh + bm + bl)
// is accumulated from the six product planes hh, hm, mh, hl, lh, mm (smallest first) by v_mfma_f32_32x32x16_bf16 -- 24 MFMAs of
// 8 passes per 32 x 32 x 64 block instead of 32 of 16 passes.  The three planes left out (ml, lm, ll) are below 2^-23 of
// |a_k b_k| per term: the rounding an fp32 dot product of this length carries anyway (the ranking is fp32-accurate, not
// bit-identical to any one fp32 summation order -- neither is the reference's sgemm).
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
typedef __attribute__((ext_vector_type(2))) uint32_t u32x2;
struct Planes2 { uint32_t h, m, l; };                        // two consecutive values, bf16 pairs (low half = first value)
static __device__ __forceinline__ Planes2 split3(float x0, float x1) {
    const uint32_t M = 0xffff0000u;
    const float h0 = __uint_as_float(__float_as_uint(x0) & M), h1 = __uint_as_float(__float_as_uint(x1) & M);
    const float r0 = x0 - h0, r1 = x1 - h1;
    const float m0 = __uint_as_float(__float_as_uint(r0) & M), m1 = __uint_as_float(__float_as_uint(r1) & M);
    const float l0 = r0 - m0, l1 = r1 - m1;
    Planes2 o;
    o.h = (__float_as_uint(h0) >> 16) | __float_as_uint(h1);
    o.m = (__float_as_uint(m0) >> 16) | __float_as_uint(m1);
    o.l = (__float_as_uint(l0) >> 16) | (__float_as_uint(l1) & M);
    return o;
}
static __device__ __forceinline__ f32x16 mfma_bf16(const u32x4 &a, const u32x4 &b, const f32x16 &c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

template <int D, int KS, typename IDT, int WGS, int NBUF = 2, bool SPLIT3 = false, bool MASKS = false>
                                                                // KS: list slots per lane (a multiple of 4, >= K); WGS: workgroups per CU;
                                                                // NBUF: item tile buffers in LDS (1: one more barrier per tile)
__global__ void __launch_bounds__(256, WGS) k_eval_topk(EvalArgs a) {
    constexpr int HALF = D / 2, RS = D + 4;            // row stride of the LDS item tile (floats)
    // SPLIT3: three bf16 planes per tile, rows of D + 8 bf16 (144 bytes at d = 64: the 16-byte reads of 16 lanes at one
    // column start 36 banks apart and cover the 64 banks once)
    constexpr int RSB = D + 8, NCH = D / 16, PLANE_B = 32 * RSB * 2;
    constexpr int TILE_F = SPLIT3 ? 3 * PLANE_B / 4 : 32 * RS;
    constexpr int LPT = 32 * D * 4 / 16 / 256;          // 16-byte pieces per thread per item tile
    static_assert(LPT >= 1, "tile smaller than the workgroup");
    __shared__ __attribute__((aligned(16))) float tile_lds[NBUF][TILE_F];
    // per-lane candidate lists: KS slots (slots >= K hold +inf: never the minimum, never output),
    // 16-byte aligned rows so the minimum scan is KS/4 independent ds_read_b128
    // (row stride KS floats = 80 / 128 bytes: 16 lanes' 16-byte reads start 20 / 32 banks apart and cover the 64 banks once)
    constexpr int KSP = (KS + 7) / 8 * 8, KH = KSP / 2;       // slots padded so that each of a user's two lanes scans KH of them
    __shared__ __attribute__((aligned(16))) float list_s[128][KSP];
    __shared__ IDT list_i[128][KS + ListId<IDT>::PAD];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int j = lane & 31, h = lane >> 5;
    const int K = a.K;
    const float *items = a.E + (int64_t)a.n_users * D;

    // ---- this lane's user: half a row in registers (B operand)
    const int64_t slot = (int64_t)blockIdx.x * 128 + wid * 32 + j;
    const bool have = slot < a.n_eval;
    const int32_t uid = have ? a.users[slot] : 0;
    float b[SPLIT3 ? 1 : HALF];
    u32x4 bh[SPLIT3 ? NCH : 1], bm[SPLIT3 ? NCH : 1], bl[SPLIT3 ? NCH : 1];   // SPLIT3: k = 16 c + 8 h + [0, 8) of MFMA step c
    if constexpr (SPLIT3) {
#pragma unroll
        for (int c = 0; c < NCH; c++) {
            const float *up = a.E + (int64_t)uid * D + 16 * c + 8 * h;
            const f32x4 v0 = *reinterpret_cast<const f32x4 *>(up), v1 = *reinterpret_cast<const f32x4 *>(up + 4);
            const Planes2 p0 = split3(v0.x, v0.y), p1 = split3(v0.z, v0.w), p2 = split3(v1.x, v1.y), p3 = split3(v1.z, v1.w);
            bh[c] = u32x4{p0.h, p1.h, p2.h, p3.h}; bm[c] = u32x4{p0.m, p1.m, p2.m, p3.m}; bl[c] = u32x4{p0.l, p1.l, p2.l, p3.l};
        }
    } else {
        const float *up = a.E + (int64_t)uid * D + h * HALF;
#pragma unroll
        for (int s = 0; s < HALF; s += 4) {
            const f32x4 v = *reinterpret_cast<const f32x4 *>(up + s);
            b[s] = v.x; b[s + 1] = v.y; b[s + 2] = v.z; b[s + 3] = v.w;
        }
    }
    // ---- train-positive cursor (lower lane of the pair walks it; the mask is shared with the upper one)
    int64_t tp = 0, tend = 0;
    int32_t nid = 0x7fffffff, nnid = 0x7fffffff;
    const int ntiles_all = (a.m_items + 31) / 32;
    const int t_begin = (int)((int64_t)ntiles_all * blockIdx.y / gridDim.y), t_end = (int)((int64_t)ntiles_all * (blockIdx.y + 1) / gridDim.y);
    uint32_t mask_next = 0;        // MASKS: this user's word of the NEXT tile, in flight under this tile's work
    const uint32_t slot32 = (uint32_t)slot;     // (uniform row base + one 32-bit lane offset: slots past n_eval read their padding column, zeros)
    if (MASKS) { if (t_begin < t_end) mask_next = (a.masks + (int64_t)t_begin * a.mask_stride)[slot32]; }
    else if (have && h == 0) {
        tp = a.train_ptr[uid]; tend = a.train_ptr[uid + 1];
        if (t_begin > 0) {                                      // first train positive inside this workgroup's item range
            int64_t lo = tp, hi = tend;
            while (lo < hi) { const int64_t mid = (lo + hi) >> 1; if (a.train_idx[mid] < t_begin * 32) lo = mid + 1; else hi = mid; }
            tp = lo;
        }
        if (tp < tend) nid = a.train_idx[tp];
        if (tp + 1 < tend) nnid = a.train_idx[tp + 1];
    }
    const int ul = wid * 32 + j;                               // the user's list (shared by its two lanes)
    if (h == 0) for (int k = 0; k < KSP; k++) { list_s[ul][k] = k < K ? EVAL_NEG_INF : 3.0e38f; if (k < KS) list_i[ul][k] = ListId<IDT>::EMPTY; }
    const int id0 = sizeof(IDT) == 2 ? t_begin * 32 : 0;       // list ids are stored relative to this
    float thr = EVAL_NEG_INF;      // the user's K-th best so far (the same value in both lanes), or the other parts' if that is higher
    int pmin = 0;                  // where it sits in the list

    // piece p of a tile: item row p / (D/4), 16-byte column p % (D/4)
    f32x4 pre[LPT];
    auto load_tile = [&](int t) {
#pragma unroll
        for (int q = 0; q < LPT; q++) {
            const int p = tid + q * 256, r = p / (D / 4), c = p % (D / 4);
            const int64_t item = (int64_t)t * 32 + r;
            pre[q] = item < a.m_items ? *reinterpret_cast<const f32x4 *>(items + item * D + c * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
    };
    auto store_tile = [&](int buf) {
#pragma unroll
        for (int q = 0; q < LPT; q++) {
            const int p = tid + q * 256, r = p / (D / 4), c = p % (D / 4);
            if constexpr (SPLIT3) {                             // split once per workgroup, not once per wave
                const Planes2 p0 = split3(pre[q].x, pre[q].y), p1 = split3(pre[q].z, pre[q].w);
                char *dst = reinterpret_cast<char *>(tile_lds[buf]) + (r * RSB + 4 * c) * 2;
                *reinterpret_cast<uint2 *>(dst) = make_uint2(p0.h, p1.h);
                *reinterpret_cast<uint2 *>(dst + PLANE_B) = make_uint2(p0.m, p1.m);
                *reinterpret_cast<uint2 *>(dst + 2 * PLANE_B) = make_uint2(p0.l, p1.l);
            } else {
                *reinterpret_cast<f32x4 *>(&tile_lds[buf][r * RS + c * 4]) = pre[q];
            }
        }
    };
    load_tile(t_begin);
    store_tile(NBUF == 2 ? t_begin & 1 : 0);
    __syncthreads();
    typedef __attribute__((address_space(1))) float gf32;
    for (int t = t_begin; t < t_end; t++) {
        const int buf = NBUF == 2 ? t & 1 : 0;
        if (a.thr_pub && have && ((t - t_begin) & (EVAL_PUB_TILES - 1)) == EVAL_PUB_TILES - 1) {
            // (agent-scope accesses: the parts run on different XCDs, whose L2s do not see each other's plain stores)
            if (h == 0) __hip_atomic_store((gf32 *)(a.thr_pub + (int64_t)blockIdx.y * a.n_eval + slot), thr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            for (int q = 0; q < (int)gridDim.y; q++)
                if (q != (int)blockIdx.y)
                    thr = fmaxf(thr, __hip_atomic_load((gf32 *)(a.thr_pub + (int64_t)q * a.n_eval + slot), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));   // fmaxf drops a NaN
        }
        if (t + 1 < t_end) load_tile(t + 1);                   // in flight under this tile's MFMAs
        // ---- mask of this tile's train positives for my user
        const int base = t * 32;
        uint32_t mask = 0;
        if (MASKS) {
            mask = mask_next;                                   // both lanes of the user load the same word
            if (t + 1 < t_end) mask_next = (a.masks + (int64_t)(t + 1) * a.mask_stride)[slot32];
        } else {
            while (nid < base + 32) {                           // nid >= base always: tiles ascend
                mask |= 1u << (nid - base);
                nid = nnid; tp++;
                nnid = (tp + 1 < tend) ? a.train_idx[tp + 1] : 0x7fffffff;
            }
            mask = __shfl(mask, j);                             // lane j (h = 0) holds the user's mask
        }
        // ---- 32 items x 32 users: scores[item i][user j] = sum_k I[i][k] U[j][k]
        f32x16 acc = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        if constexpr (SPLIT3) {
            const char *arow = reinterpret_cast<const char *>(tile_lds[buf]) + (j * RSB + 8 * h) * 2;   // lane (i = j, h): k = 16 c + 8 h + [0, 8)
            u32x4 ah[NCH], am[NCH], al[NCH];
#pragma unroll
            for (int c = 0; c < NCH; c++) {
                am[c] = *reinterpret_cast<const u32x4 *>(arow + PLANE_B + 32 * c);
                ah[c] = *reinterpret_cast<const u32x4 *>(arow + 32 * c);
                al[c] = *reinterpret_cast<const u32x4 *>(arow + 2 * PLANE_B + 32 * c);
            }
#pragma unroll
            for (int c = 0; c < NCH; c++) acc = mfma_bf16(am[c], bm[c], acc);
#pragma unroll
            for (int c = 0; c < NCH; c++) { acc = mfma_bf16(ah[c], bl[c], acc); acc = mfma_bf16(al[c], bh[c], acc); }
#pragma unroll
            for (int c = 0; c < NCH; c++) { acc = mfma_bf16(ah[c], bm[c], acc); acc = mfma_bf16(am[c], bh[c], acc); }
#pragma unroll
            for (int c = 0; c < NCH; c++) acc = mfma_bf16(ah[c], bh[c], acc);
        } else {
            const float *arow = &tile_lds[buf][j * RS + h * HALF];  // A operand: lane (i = j, h) holds item i's half row
#pragma unroll
            for (int s = 0; s < HALF; s += 4) {
                const f32x4 av = *reinterpret_cast<const f32x4 *>(arow + s);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.x, b[s], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.y, b[s + 1], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.z, b[s + 2], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.w, b[s + 3], acc, 0, 0, 0);
            }
        }
        if (NBUF == 1) {                                        // every wave has read the tile: the next one may overwrite it
            __syncthreads();
            if (t + 1 < t_end) store_tile(0);
        }
        // ---- my 16 scores: item row = (reg & 3) + 8 * (reg >> 2) + 4 * h.  First only MARK the ones that can
        //      enter the user's top K: above this lane's K-th best AND not below the partner lane's (the other
        //      half of the user's items: K items at or above that score already exist).  Then the marked scores
        //      are inserted one per round, all lanes together: the rounds of a tile are the LARGEST number of
        //      marked scores any lane of the workgroup has (the waves meet at the tile's barrier), not the number
        //      of score positions where some lane inserts -- mid-sweep that is 1-2 rounds instead of 5-10.
        uint32_t pend = 0;
        // The reference's semantics (a train positive's score IS -1024 and competes, Procedure.py:181; rows past the table do
        // not exist) cost three instructions per score.  They can only matter while some user with a train positive in this
        // tile has no K scores above -1024 yet, or in the table's last tile: everywhere else a train positive is simply
        // never marked.
        const bool exact_mask = base + 32 > a.m_items || __any(mask != 0u && !(thr >= -1024.0f));
        if (exact_mask) {
#pragma unroll
            for (int reg = 0; reg < 16; reg++) {
                const int row = (reg & 3) + 8 * (reg >> 2) + 4 * h;
                float sc = acc[reg];
                if ((mask >> row) & 1u) sc = -1024.0f;                      // Procedure.py:181
                if (base + row >= a.m_items) sc = EVAL_NEG_INF;             // past the table
                acc[reg] = sc;
                if (sc > thr) pend |= 1u << reg;
            }
        } else {
#pragma unroll
            for (int reg = 0; reg < 16; reg++)
                if (acc[reg] > thr) pend |= 1u << reg;
            // my rows' bits of the train-positive mask, in register order: nibble g of the result = rows 8 g + 4 h + [0, 4)
            const uint32_t m2 = mask >> (4 * h);
            pend &= ~((m2 & 0xfu) | ((m2 >> 4) & 0xf0u) | ((m2 >> 8) & 0xf00u) | ((m2 >> 12) & 0xf000u));
        }
        // One list per USER: of its two lanes (the two halves of the tile's rows) at most one inserts per round, the lower
        // one first; then BOTH rescan half of the list each for the new K-th best and combine through v_permlane32_swap
        // (no LDS round trip, and neither lane idles during the scan).
        while (__any(pend != 0u)) {
            const uint32_t mine = pend != 0u ? 1u : 0u;
            const u32x2 mm = __builtin_amdgcn_permlane32_swap(mine, mine, false, false);      // .x: lower lane's, .y: upper lane's
            const bool act = mine && (h == 0 || mm.x == 0u);
            uint32_t did = 0u;
            if (act) {
                const int reg = __builtin_ctz(pend);
                pend &= pend - 1u;
                // acc[reg] for a per-lane reg: a binary select tree (4 bit tests + 15 selects; a linear chain is 15 + 15)
                const bool b0 = reg & 1, b1 = reg & 2, b2 = reg & 4, b3 = reg & 8;
                const float e0 = pick(b0, acc[1], acc[0]), e1 = pick(b0, acc[3], acc[2]), e2 = pick(b0, acc[5], acc[4]), e3 = pick(b0, acc[7], acc[6]);
                const float e4 = pick(b0, acc[9], acc[8]), e5 = pick(b0, acc[11], acc[10]), e6 = pick(b0, acc[13], acc[12]), e7 = pick(b0, acc[15], acc[14]);
                const float f0 = pick(b1, e1, e0), f1 = pick(b1, e3, e2), f2 = pick(b1, e5, e4), f3 = pick(b1, e7, e6);
                const float sc = pick(b3, pick(b2, f3, f2), pick(b2, f1, f0));
                if (sc > thr) {                                         // thr may have risen since the marking
                    const int row = (reg & 3) + 8 * (reg >> 2) + 4 * h;
                    list_s[ul][pmin] = sc; list_i[ul][pmin] = (IDT)(base + row - id0);
                    did = 1u;
                }
            }
            const u32x2 dd = __builtin_amdgcn_permlane32_swap(did, did, false, false);
            if (dd.x | dd.y) {
                // (the wave's LDS accesses execute in program order: the partner's write above is visible to these reads)
                f32x4 q[KH / 4];
#pragma unroll
                for (int k4 = 0; k4 < KH / 4; k4++) q[k4] = *reinterpret_cast<const f32x4 *>(&list_s[ul][h * KH + 4 * k4]);
                float m = q[0].x; int pm = 0;
#pragma unroll
                for (int k = 1; k < KH; k++) { const float v = q[k / 4][k % 4]; if (v < m) { m = v; pm = k; } }
                pm += h * KH;
                const u32x2 mv = __builtin_amdgcn_permlane32_swap(__float_as_uint(m), __float_as_uint(m), false, false);
                const u32x2 pv = __builtin_amdgcn_permlane32_swap((uint32_t)pm, (uint32_t)pm, false, false);
                const float m0 = __uint_as_float(mv.x), m1 = __uint_as_float(mv.y);
                const bool lower = m0 <= m1;                            // ties: the first slot, as one lane's scan would
                thr = fmaxf(lower ? m0 : m1, thr);          // (thr never falls: it may already stand above the list's minimum, on another part's word)
                pmin = (int)(lower ? pv.x : pv.y);
            }
        }
        if (NBUF == 2 && t + 1 < t_end) store_tile(buf ^ 1);
        __syncthreads();
    }
    // ---- sort the user's list descending (ties: lower item id first)
    __syncthreads();
    if (h == 0 && have) {
        for (int r = 0; r < K; r++) {
            float best = EVAL_NEG_INF * 2.0f; int bi = 0x7fffffff, bk = -1;
            for (int k = 0; k < K; k++) {
                const float v = list_s[ul][k]; const IDT raw = list_i[ul][k];
                const int id = id0 + (int)raw;
                if (raw != ListId<IDT>::EMPTY && (v > best || (v == best && id < bi))) { best = v; bi = id; bk = k; }
            }
            if (bk >= 0) list_i[ul][bk] = ListId<IDT>::EMPTY;   // taken
            if (a.part_items) {
                const int64_t o = (slot * gridDim.y + blockIdx.y) * K + r;
                a.part_items[o] = bk >= 0 ? bi : -1; a.part_scores[o] = best;
            } else {
                a.out_items[slot * K + r] = bk >= 0 ? bi : -1;
                if (a.out_scores) a.out_scores[slot * K + r] = best;
            }
        }
    }
}

// top K of a user's P sorted partial lists (descending, ties: lower item id first): a P-way merge by one thread
__global__ void __launch_bounds__(256) k_eval_merge(EvalArgs a, int P) {
    const int64_t s = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (s >= a.n_eval) return;
    const int K = a.K;
    int cur[4] = {0, 0, 0, 0};
    for (int r = 0; r < K; r++) {
        float best = EVAL_NEG_INF * 2.0f; int bi = 0x7fffffff, bp = -1;
        for (int p = 0; p < P; p++) {
            if (cur[p] >= K) continue;
            const int64_t o = (s * P + p) * K + cur[p];
            const int id = a.part_items[o]; const float v = a.part_scores[o];
            if (id >= 0 && (v > best || (v == best && id < bi))) { best = v; bi = id; bp = p; }
        }
        if (bp >= 0) cur[bp]++;
        a.out_items[s * K + r] = bp >= 0 ? bi : -1;
        if (a.out_scores) a.out_scores[s * K + r] = best;
    }
}

// per-user metrics from the ranked ids: r_j = [id_j in the user's test list] (sorted: binary search)
struct MetricArgs {
    const int32_t *topk; int32_t n_eval, K;
    const int64_t *test_ptr; const int32_t *test_idx;      // per evaluated slot, ids ascending
    int32_t ks[8]; int32_t n_ks;
    double *per_user;                                        // [n_eval, 3 * n_ks]: precision | recall | ndcg
};

__global__ void __launch_bounds__(256) k_eval_metrics(MetricArgs a) {
    const int64_t s = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (s >= a.n_eval) return;
    const int64_t b = a.test_ptr[s], e = a.test_ptr[s + 1];
    const int len = (int)(e - b);
    uint64_t hits = 0;
    for (int r = 0; r < a.K; r++) {
        const int32_t id = a.topk[s * a.K + r];
        int64_t lo = b, hi = e;
        while (lo < hi) { const int64_t mid = (lo + hi) >> 1; if (a.test_idx[mid] < id) lo = mid + 1; else hi = mid; }
        if (lo < e && a.test_idx[lo] == id) hits |= 1ull << r;
    }
    for (int q = 0; q < a.n_ks; q++) {
        const int k = a.ks[q];
        double right = 0.0, dcg = 0.0, idcg = 0.0;
        for (int r = 0; r < k; r++) {
            const double disc = 1.0 / log2((double)(r + 2));
            if ((hits >> r) & 1ull) { right += 1.0; dcg += disc; }
            if (r < len) idcg += disc;
        }
        if (idcg == 0.0) idcg = 1.0;
        double *o = a.per_user + s * 3 * a.n_ks;
        o[q] = right / (double)k;                         // utils.py:173-187
        o[a.n_ks + q] = len > 0 ? right / (double)len : 0.0;
        o[2 * a.n_ks + q] = dcg / idcg;                   // utils.py:190-203
    }
}

// sums over users in a fixed order (one workgroup: deterministic)
__global__ void __launch_bounds__(256) k_eval_sum(const double *per_user, int32_t n_eval, int32_t width, double *sums) {
    __shared__ double part[256];
    for (int c = 0; c < width; c++) {
        double acc = 0.0;
        for (int64_t s = threadIdx.x; s < n_eval; s += 256) acc += per_user[s * width + c];
        part[threadIdx.x] = acc;
        __syncthreads();
        for (int w = 128; w > 0; w >>= 1) {
            if ((int)threadIdx.x < w) part[threadIdx.x] += part[threadIdx.x + w];
            __syncthreads();
        }
        if (threadIdx.x == 0) sums[c] = part[0];
        __syncthreads();
    }
}

// masks[t * stride + slot]: one wave per evaluation slot walks the user's train positives
__global__ void __launch_bounds__(256) k_eval_masks(const int32_t *users, int32_t n_eval, const int64_t *train_ptr, const int32_t *train_idx,
                                                    int32_t m_items, uint32_t *masks, int64_t stride) {
    const int64_t slot = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (slot >= n_eval) return;
    const int32_t u = users[slot];
    const int64_t b = train_ptr[u], e = train_ptr[u + 1];
    for (int64_t i = b + (threadIdx.x & 63); i < e; i += 64) {
        const int32_t it = train_idx[i];
        if (it >= 0 && it < m_items) atomicOr(masks + (int64_t)(it >> 5) * stride + slot, 1u << (it & 31));
    }
}

static inline int64_t eval_mask_stride(int32_t n_eval) { return ((int64_t)n_eval + 127) / 128 * 128; }

extern "C" int64_t lgcn_eval_mask_words(int32_t m_items, int32_t n_eval) {
    if (m_items <= 0 || n_eval <= 0) return 0;
    return (int64_t)((m_items + 31) / 32) * eval_mask_stride(n_eval);
}

extern "C" int lgcn_eval_build_masks(const int32_t *users, int32_t n_eval, const int64_t *train_indptr, const int32_t *train_indices,
                                     int32_t m_items, uint32_t *masks, void *stream) {
    if (!users || !train_indptr || !train_indices || !masks || m_items <= 0 || n_eval < 0) { lgcn_set_error("lgcn_eval_build_masks: invalid argument"); return 3; }
    if (n_eval == 0) return 0;
    hipStream_t st = (hipStream_t)stream;
    const int64_t stride = eval_mask_stride(n_eval);
    if (hipMemsetAsync(masks, 0, sizeof(uint32_t) * (size_t)lgcn_eval_mask_words(m_items, n_eval), st) != hipSuccess) { lgcn_set_error("lgcn_eval_build_masks: memset failed"); return 10; }
    hipLaunchKernelGGL(k_eval_masks, dim3((unsigned)((n_eval + 3) / 4)), dim3(256), 0, st, users, n_eval, train_indptr, train_indices, m_items, masks, stride);
    if (hipGetLastError() != hipSuccess) { lgcn_set_error("lgcn_eval_build_masks: launch failed"); return 10; }
    return 0;
}

static int eval_topk(const float *E, int32_t n_users, int32_t m_items, int32_t d,
                     const int32_t *users, int32_t n_eval,
                     const int64_t *train_indptr, const int32_t *train_indices,
                     int32_t K, int32_t *topk_items, float *topk_scores, const uint32_t *masks, void *stream, bool split3) {
    if (!E || !users || !train_indptr || !train_indices || !topk_items || n_users <= 0 || m_items <= 0 || n_eval < 0) {
        lgcn_set_error("lgcn_eval_topk: invalid argument"); return 3;
    }
    if (K < 1 || K > EVAL_KMAX || K > m_items) { lgcn_set_error("lgcn_eval_topk: K must be in 1..64 and <= m_items"); return 3; }
    if (n_eval == 0) return 0;
    EvalArgs a{E, n_users, m_items, users, n_eval, train_indptr, train_indices, K, topk_items, topk_scores, nullptr, nullptr, nullptr,
               masks, eval_mask_stride(n_eval)};
    const unsigned blocks = (unsigned)((n_eval + 127) / 128);
    hipStream_t st = (hipStream_t)stream;
    // The item sweep is split over `parts` workgroups per user block, as many as fit a CU together, so that one's list
    // maintenance runs under the others' MFMAs.  Compact lists (16-bit ids relative to the part's first item: d <= 64,
    // at most 2047 tiles per part) make that three for K <= 20 and two for K <= 64 (the lists grow with K); d = 128: two
    // for K <= 20; otherwise one workgroup per CU with int32 ids and the fp32 matrix instructions.
    // (measured on Gowalla with int32 ids, two per CU: 1 / 2 / 3 / 4 parts = 3.40 / 2.54 / 2.79 / 2.70 ms)
    const int ntiles = (m_items + 31) / 32;
    const bool small = K <= 20;
    const bool split = m_items >= 4096 && ((d <= 128 && small) || d <= 64);
    const int want = (d <= 64 && small) ? EVAL_PARTS : 2;      // d = 128: 256 registers; K > 20: 64-slot lists -- two workgroups per CU
    const bool id16 = split && (ntiles + want - 1) / want + 1 <= EVAL_ID16_MAX_TILES;
    const int parts = id16 ? want : (split && small) ? 2 : 1;
    void *tmp = nullptr;
    bool tmp_sync = false;
    if (parts > 1) {
        const size_t n = (size_t)n_eval * parts * K;
        const size_t lists = (n * (sizeof(int32_t) + sizeof(float)) + 255) & ~(size_t)255;
        const size_t pub = EVAL_PUB_TILES > 0 ? (size_t)parts * n_eval * sizeof(float) : 0;
        if (hipMallocAsync(&tmp, lists + pub, st) != hipSuccess) {
            (void)hipGetLastError();
            tmp = nullptr;
            if (hipMalloc(&tmp, lists + pub) != hipSuccess) { lgcn_set_error("lgcn_eval_topk: cannot allocate the partial lists"); return 4; }
            tmp_sync = true;       // no stream-ordered pool on this runtime: plain allocation, freed after a synchronise
        }
        a.part_items = (int32_t *)tmp; a.part_scores = (float *)((int32_t *)tmp + n);
        if (pub) {
            a.thr_pub = (float *)((char *)tmp + lists);
            if (hipMemsetAsync(a.thr_pub, 0xff, pub, st) != hipSuccess) { lgcn_set_error("lgcn_eval_topk: memset failed"); return 10; }     // NaN = nothing published
        }
    }
    const dim3 grid(blocks, parts);
    const bool mk = masks != nullptr;
#define EV(...) do { if (mk) hipLaunchKernelGGL((k_eval_topk<__VA_ARGS__, true>), grid, dim3(256), 0, st, a); \
                     else hipLaunchKernelGGL((k_eval_topk<__VA_ARGS__, false>), grid, dim3(256), 0, st, a); } while (0)
    // the generic form: int32 ids, one workgroup per CU, fp32 matrix instructions
#define EVAL_GENERIC(DD) do { if (K <= 20) EV(DD, 20, int32_t, 1, 2, false); else if (K <= 32) EV(DD, 32, int32_t, 1, 2, false); \
                              else EV(DD, 64, int32_t, 1, (DD >= 256 ? 1 : 2), false); } while (0)
#define EVAL_SMALL(DD) do { if (id16 && small && split3) EV(DD, 20, uint16_t, EVAL_PARTS, EVAL_NBUF, true); \
                            else if (id16 && small) EV(DD, 20, uint16_t, EVAL_PARTS, EVAL_NBUF, false); \
                            else if (id16 && split3) EV(DD, 64, uint16_t, 2, 2, true); \
                            else if (id16) EV(DD, 64, uint16_t, 2, 2, false); \
                            else EVAL_GENERIC(DD); } while (0)
    switch (d) {
    case 32: EVAL_SMALL(32); break;
    case 64: EVAL_SMALL(64); break;
    case 128:
        if (id16 && small && split3) EV(128, 20, uint16_t, 2, 1, true);
        else if (id16 && small) EV(128, 20, uint16_t, 2, 2, false);
        else EVAL_GENERIC(128);
        break;
    case 256: EVAL_GENERIC(256); break;
    default: if (tmp) { if (tmp_sync) (void)hipFree(tmp); else (void)hipFreeAsync(tmp, st); } lgcn_set_error("embedding dim must be 32, 64, 128 or 256"); return 3;
    }
#undef EVAL_SMALL
#undef EVAL_GENERIC
#undef EV
    if (parts > 1) {
        hipLaunchKernelGGL(k_eval_merge, dim3((unsigned)((n_eval + 255) / 256)), dim3(256), 0, st, a, parts);
        if (tmp_sync) { (void)hipStreamSynchronize(st); (void)hipFree(tmp); }
        else (void)hipFreeAsync(tmp, st);
    }
    if (hipGetLastError() != hipSuccess) { lgcn_set_error("lgcn_eval_topk: launch failed"); return 10; }
    return 0;
}

extern "C" int lgcn_eval_topk(const float *E, int32_t n_users, int32_t m_items, int32_t d, const int32_t *users, int32_t n_eval,
                              const int64_t *train_indptr, const int32_t *train_indices, int32_t K, int32_t *topk_items,
                              float *topk_scores, void *stream) {
    return eval_topk(E, n_users, m_items, d, users, n_eval, train_indptr, train_indices, K, topk_items, topk_scores, nullptr, stream, EVAL_SPLIT3 != 0);
}

extern "C" int lgcn_eval_topk_masked(const float *E, int32_t n_users, int32_t m_items, int32_t d, const int32_t *users, int32_t n_eval,
                                     const int64_t *train_indptr, const int32_t *train_indices, int32_t K, int32_t *topk_items,
                                     float *topk_scores, const uint32_t *masks, void *stream) {
    return eval_topk(E, n_users, m_items, d, users, n_eval, train_indptr, train_indices, K, topk_items, topk_scores, masks, stream, EVAL_SPLIT3 != 0);
}

extern "C" int lgcn_eval_topk_fp32(const float *E, int32_t n_users, int32_t m_items, int32_t d, const int32_t *users, int32_t n_eval,
                                   const int64_t *train_indptr, const int32_t *train_indices, int32_t K, int32_t *topk_items,
                                   float *topk_scores, void *stream) {
    return eval_topk(E, n_users, m_items, d, users, n_eval, train_indptr, train_indices, K, topk_items, topk_scores, nullptr, stream, false);
}

extern "C" int lgcn_eval_metrics(const int32_t *topk_items, int32_t n_eval, int32_t K,
                                 const int64_t *test_indptr, const int32_t *test_items_sorted,
                                 const int32_t *ks, int32_t n_ks, double *per_user, double *sums, void *stream) {
    if (!topk_items || !test_indptr || !test_items_sorted || !ks || !per_user || !sums || n_eval < 0) {
        lgcn_set_error("lgcn_eval_metrics: invalid argument"); return 3;
    }
    if (n_ks < 1 || n_ks > 8 || K < 1 || K > 64) { lgcn_set_error("lgcn_eval_metrics: 1..8 cut-offs, K <= 64"); return 3; }
    MetricArgs a{};
    a.topk = topk_items; a.n_eval = n_eval; a.K = K; a.test_ptr = test_indptr; a.test_idx = test_items_sorted;
    a.n_ks = n_ks; a.per_user = per_user;
    for (int q = 0; q < n_ks; q++) {
        if (ks[q] < 1 || ks[q] > K) { lgcn_set_error("lgcn_eval_metrics: a cut-off exceeds K"); return 3; }
        a.ks[q] = ks[q];
    }
    hipStream_t st = (hipStream_t)stream;
    if (n_eval > 0) hipLaunchKernelGGL(k_eval_metrics, dim3((unsigned)((n_eval + 255) / 256)), dim3(256), 0, st, a);
    hipLaunchKernelGGL(k_eval_sum, dim3(1), dim3(256), 0, st, per_user, n_eval, 3 * n_ks, sums);
    if (hipGetLastError() != hipSuccess) { lgcn_set_error("lgcn_eval_metrics: launch failed"); return 10; }
    return 0;
}
