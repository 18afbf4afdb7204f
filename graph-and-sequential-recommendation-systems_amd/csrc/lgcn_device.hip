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
//    are evaluated on the <= 3B rows the batch gathers (k_triplet).
//  * backward is the Horner chain h_{k-1} = Gs + A h_k, Gs = G/(K+1); Gs has
//    <= 3B non-zero rows, so the first backward SpMM skips zero rows by bitmap
//    and gathers the flagged rows from a fp32 copy made once per step (k_g32).
//  * the last backward SpMM applies Adam in its epilogue (no dense grad buffer).
//  * the scatter-add of per-triplet gradient rows uses 64-bit fixed-point integer
//    atomics (scale 2^50): integer addition is associative, so the result is
//    bitwise reproducible and independent of batch sharding.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <algorithm>
#include <array>
#include <new>
#include <vector>

#include "lgcn_hip.h"
#include "lgcn_internal.h"

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
typedef __attribute__((ext_vector_type(8))) float f32x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __bf16 bf16_t;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(16))) __bf16 bf16x16;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
typedef __attribute__((ext_vector_type(2))) float f32x2;
// LGCN_FP8: a table of n rows is  n * D bytes of OCP E4M3 values  followed by  n fp32 ROW SCALES  (value = scale * fp8): one
// pointer names both.  Scales are powers of two with max|row| / scale in [64, 128) (inside the range of both E4M3 flavours;
// a division by a power of two is exact, so the quantisation is a pure fp8 rounding the tests restate bit for bit).
// The columns of an fp8 row are CHUNK-INTERLEAVED: with L = D / 16 lanes per row, the 16 bytes at l*16 hold the four 4-column
// chunks q = j*L + l (j = 0..3): byte l*16 + 4j + e = column (j*L + l)*4 + e.  A lane that gathers 16 bytes therefore owns four
// float4 chunks that lie L chunks apart, and the fp32 side of every epilogue (G32, P / M / V, fp32 outputs) is four fully
// coalesced float4 accesses per row instead of 64 contiguous bytes per lane (measured on the 10M x 1M graph: Adam's operands of
// the fp8 step streamed at 4.0 TB/s with the natural order against 6.6 TB/s in the fp32 step).  The table is an opaque blob of
// the ABI (lgcn_to_fp8 writes it, the kernels read it); lgcn_fp8_col_of_byte gives the order for anyone who decodes it.
struct fp8_t { uint8_t v; };
__host__ __device__ __forceinline__ int fp8_byte_of_col(int c, int D) { const int L = D / 16, q = c >> 2; return (q % L) * 16 + 4 * (q / L) + (c & 3); }
__device__ __forceinline__ const float *fp8_scales(const void *tab, int64_t n_rows, int D) { return (const float *)((const char *)tab + n_rows * D); }
__device__ __forceinline__ float *fp8_scales(void *tab, int64_t n_rows, int D) { return (float *)((char *)tab + n_rows * D); }

#ifndef LGCN_GATHER_U
#define LGCN_GATHER_U 8      /* max row gathers in flight per lane (8 x 16 B raw = 32 VGPRs) */
#endif
#define FIXED_SCALE 1125899906842624.0   /* 2^50 */
#define FIXED_INV   8.8817841970012523e-16 /* 2^-50 */

// C fp32 values per lane (C = 4: 16 B of fp32 / 8 B of bf16; C = 8: 32 B of fp32 / 16 B of bf16)
template <int C> struct VecF;
template <> struct VecF<4> { typedef f32x4 T; typedef bf16x4 B; };
template <> struct VecF<8> { typedef f32x8 T; typedef bf16x8 B; };
template <> struct VecF<16> { typedef f32x16 T; typedef bf16x16 B; };

template <int C> __device__ __forceinline__ typename VecF<C>::T zerov() {
    typename VecF<C>::T z;
#pragma unroll
    for (int i = 0; i < C; i++) z[i] = 0.f;
    return z;
}
template <int C> __device__ __forceinline__ typename VecF<C>::T loadv(const float *p) {
    return *reinterpret_cast<const typename VecF<C>::T *>(p);
}
template <int C> __device__ __forceinline__ typename VecF<C>::T loadv(const bf16_t *p) {
    return __builtin_convertvector(*reinterpret_cast<const typename VecF<C>::B *>(p), typename VecF<C>::T);
}
template <int C> __device__ __forceinline__ void storev(float *p, typename VecF<C>::T v) {
    *reinterpret_cast<typename VecF<C>::T *>(p) = v;
}
template <int C> __device__ __forceinline__ void storev(bf16_t *p, typename VecF<C>::T v) {
    *reinterpret_cast<typename VecF<C>::B *>(p) = __builtin_convertvector(v, typename VecF<C>::B);
}
__device__ __forceinline__ f32x4 load4(const float *p) { return loadv<4>(p); }
__device__ __forceinline__ f32x4 load4(const bf16_t *p) { return loadv<4>(p); }
__device__ __forceinline__ void store4(float *p, f32x4 v) { storev<4>(p, v); }
__device__ __forceinline__ void store4(bf16_t *p, f32x4 v) { storev<4>(p, v); }
// one fp8 element of a table, decoded (k_triplet's lower-layer rows: lane = column)
__device__ __forceinline__ float fp8_decode(uint8_t b) { return __builtin_amdgcn_cvt_pk_f32_fp8((int)b, false)[0]; }
template <typename TI> __device__ __forceinline__ float tab_elem(const void *tab, int64_t n_rows, int D, int64_t row, int col);
template <> __device__ __forceinline__ float tab_elem<float>(const void *tab, int64_t, int D, int64_t row, int col) { return ((const float *)tab)[row * D + col]; }
template <> __device__ __forceinline__ float tab_elem<bf16_t>(const void *tab, int64_t, int D, int64_t row, int col) { return (float)((const bf16_t *)tab)[row * D + col]; }
template <> __device__ __forceinline__ float tab_elem<fp8_t>(const void *tab, int64_t n_rows, int D, int64_t row, int col) {
    return fp8_scales(tab, n_rows, D)[row] * fp8_decode(((const uint8_t *)tab)[row * D + fp8_byte_of_col(col, D)]);
}
// scale of a row whose largest magnitude is amax: 2^(e - 6) for amax = 1.f * 2^e  ->  amax / scale in [64, 128); rows
// below 2^-100 (and zero rows) are stored as zeros with scale 1; every other finite row, up to the top of fp32, is scaled
__device__ __forceinline__ void fp8_row_scale(float amax, float &scale, float &inv) {
    const uint32_t E = __float_as_uint(amax) >> 23;           // biased exponent (amax >= 0)
    if (E < 27u || E == 255u) { scale = 1.f; inv = E == 255u ? 1.f : 0.f; return; }      // (inf / NaN rows stay inf / NaN)
    scale = __uint_as_float((E - 6u) << 23); inv = __uint_as_float((260u - E) << 23);
}
// Store one row piece of C values per lane as fp8: the row's LPR = D / C lanes (contiguous, aligned, all active) agree on
// the row maximum through xor shuffles, every lane quantises its piece, lane l == 0 writes the scale.
// IL: v holds the lane's four interleaved chunks (an fp8 table was gathered: C = 16) -- its 16 bytes are contiguous at l*16;
// otherwise v holds C consecutive columns (an fp32 / bf16 table was gathered) and every 4-column chunk goes to its own word.
template <int D, int C, bool IL>
__device__ __forceinline__ void store_row_fp8(void *tab, int64_t n_rows, int64_t row, int l, typename VecF<C>::T v) {
    static_assert(!IL || C == 16, "interleaved accumulators come from fp8 gathers: 16 per lane");
    constexpr int LPR = D / C;
    float amax = 0.f;
#pragma unroll
    for (int i = 0; i < C; i++) amax = fmaxf(amax, fabsf(v[i]));
#pragma unroll
    for (int o = 1; o < LPR; o <<= 1) amax = fmaxf(amax, __shfl_xor(amax, o));
    float scale, inv;
    fp8_row_scale(amax, scale, inv);
    uint32_t w[C / 4];
#pragma unroll
    for (int i = 0; i < C / 4; i++) {
        int p = 0;
        p = __builtin_amdgcn_cvt_pk_fp8_f32(v[4 * i] * inv, v[4 * i + 1] * inv, p, false);
        p = __builtin_amdgcn_cvt_pk_fp8_f32(v[4 * i + 2] * inv, v[4 * i + 3] * inv, p, true);
        w[i] = (uint32_t)p;
    }
    if (IL) {
        uint32_t *dst = (uint32_t *)((char *)tab + row * D + l * 16);
#pragma unroll
        for (int i = 0; i < C / 4; i++) dst[i] = w[i];
    } else {
#pragma unroll
        for (int i = 0; i < C / 4; i++) *(uint32_t *)((char *)tab + row * D + fp8_byte_of_col((l * (C / 4) + i) * 4, D)) = w[i];
    }
    if (l == 0) fp8_scales(tab, n_rows, D)[row] = scale;
}
// row piece store by output type
// a lane's piece of an fp32 row: C consecutive columns at l*C, or (IL: the accumulators of an fp8 gather) the four float4 chunks j*L + l
template <int D, int C, bool IL> __device__ __forceinline__ typename VecF<C>::T row_load(const float *base, int64_t row, int l) {
    if constexpr (!IL) return loadv<C>(base + row * D + l * C);
    else {
        typename VecF<C>::T r;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const f32x4 t = load4(base + row * D + (j * (D / 16) + l) * 4);
            r[4 * j] = t.x; r[4 * j + 1] = t.y; r[4 * j + 2] = t.z; r[4 * j + 3] = t.w;
        }
        return r;
    }
}
template <int D, int C, bool IL> __device__ __forceinline__ void row_store(float *base, int64_t row, int l, typename VecF<C>::T v) {
    if constexpr (!IL) storev<C>(base + row * D + l * C, v);
    else {
#pragma unroll
        for (int j = 0; j < 4; j++) store4(base + row * D + (j * (D / 16) + l) * 4, f32x4{v[4 * j], v[4 * j + 1], v[4 * j + 2], v[4 * j + 3]});
    }
}
template <int D, int C, typename TO, bool IL> struct RowStore {
    static __device__ __forceinline__ void put(void *Y, int64_t, int64_t row, int l, typename VecF<C>::T v) {
        static_assert(!IL, "an fp8 gather writes fp32 or fp8");
        storev<C>((TO *)Y + row * D + l * C, v); }
};
template <int D, int C, bool IL> struct RowStore<D, C, float, IL> {
    static __device__ __forceinline__ void put(void *Y, int64_t, int64_t row, int l, typename VecF<C>::T v) { row_store<D, C, IL>((float *)Y, row, l, v); }
};
template <int D, int C, bool IL> struct RowStore<D, C, fp8_t, IL> {
    static __device__ __forceinline__ void put(void *Y, int64_t n_rows, int64_t row, int l, typename VecF<C>::T v) { store_row_fp8<D, C, IL>(Y, n_rows, row, l, v); }
};
__device__ __forceinline__ bool bit_set(const uint32_t *bm, int i) { return (bm[i >> 5] >> (i & 31)) & 1u; }

// fixed-point gradient row -> fp32 Gs row:  (float)(q * 2^-50) / (K+1)
template <int C> __device__ __forceinline__ typename VecF<C>::T loadv_fixed(const long long *p, float div) {
    typedef __attribute__((ext_vector_type(2))) long long i64x2_;
    typename VecF<C>::T r;
#pragma unroll
    for (int i = 0; i < C; i += 2) {
        const i64x2_ a = *reinterpret_cast<const i64x2_ *>(p + i);
        r[i] = (float)((double)a.x * FIXED_INV) / div; r[i + 1] = (float)((double)a.y * FIXED_INV) / div;
    }
    return r;
}

struct GatherSrc {           // what a row gather reads
    const float *S;          // fp8 table: its row scales (an entry's weight is multiplied by S[col] when it is staged), else NULL
    const void *X;           // [N,D] of TI; SPARSE: the fp32 copy of the flagged gradient rows (k_g32)
    const uint32_t *bm;      // SPARSE: non-zero-row bitmap (global, or the workgroup's LDS copy)
    float div;               // unused by the gathers (K+1; the epilogues take it from SpmmArgs)
};

// Raw (unconverted) piece of a gathered row: 16 bytes per lane for fp32 AND bf16 tables (4 / 8
// columns), so one wave-instruction always moves 1 KiB = 4 fp32 rows / 8 bf16 rows at d = 64 and the
// bf16 table needs HALF the gather instructions of the fp32 one (with 8 B per lane the bf16 kernel
// issued as many gathers as the fp32 one and was instruction-bound: +15 % for half the bytes).
// The conversion to fp32 is kept OUT of the load block: hipcc otherwise waits vmcnt(0) inside every
// block and the U gathers serialise (measured: bf16 SpMM 2x slower than fp32).
typedef __attribute__((ext_vector_type(2))) long long i64x2;
struct fixed4 { i64x2 a, b; };
template <typename TI, bool SPARSE> struct Raw;
template <> struct Raw<float, false> {
    static constexpr int CPL = 4;
    typedef f32x4 T;
    // BIG = false (table under 4 GiB): uniform base + ONE 32-bit byte offset per gather (v_lshl_add_u32, SGPR-base
    // addressing) instead of a sign extension, a 64-bit shift and a 64-bit add per gathered row
    template <bool BIG> static __device__ __forceinline__ T load(const GatherSrc &s, int col, int D, int l) {
        if (BIG) return *reinterpret_cast<const T *>((const float *)s.X + (int64_t)col * D + l * 4);
        return *reinterpret_cast<const T *>((const char *)s.X + ((uint32_t)col * (uint32_t)(D * 4) + (uint32_t)(l * 16))); }
    static __device__ __forceinline__ f32x4 cvt(const T &r, float) { return r; }
};
template <> struct Raw<bf16_t, false> {
    static constexpr int CPL = 8;
    typedef bf16x8 T;
    template <bool BIG> static __device__ __forceinline__ T load(const GatherSrc &s, int col, int D, int l) {
        if (BIG) return *reinterpret_cast<const T *>((const bf16_t *)s.X + (int64_t)col * D + l * 8);
        return *reinterpret_cast<const T *>((const char *)s.X + ((uint32_t)col * (uint32_t)(D * 2) + (uint32_t)(l * 16))); }
    static __device__ __forceinline__ f32x8 cvt(const T &r, float) { return __builtin_convertvector(r, f32x8); }
};
template <> struct Raw<fp8_t, false> {
    static constexpr int CPL = 16;
    typedef u32x4 T;
    template <bool BIG> static __device__ __forceinline__ T load(const GatherSrc &s, int col, int D, int l) {
        if (BIG) return *reinterpret_cast<const T *>((const char *)s.X + (int64_t)col * D + l * 16);
        return *reinterpret_cast<const T *>((const char *)s.X + ((uint32_t)col * (uint32_t)D + (uint32_t)(l * 16))); }
    // the row scale is NOT applied here: the entry's weight was multiplied by it when the entry was staged
    static __device__ __forceinline__ f32x16 cvt(const T &r, float) {
        f32x16 o;
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const auto a = __builtin_amdgcn_cvt_pk_f32_fp8((int)r[i], false), b = __builtin_amdgcn_cvt_pk_f32_fp8((int)r[i], true);
            o[4 * i] = a[0]; o[4 * i + 1] = a[1]; o[4 * i + 2] = b[0]; o[4 * i + 3] = b[1];
        }
        return o;
    }
};
// SPARSE: the flagged rows of Gs, converted once per step from the fixed-point accumulator to fp32 by k_g32
// (gathering the 512-byte int64 rows and converting every gathered copy -- int64 -> fp64 -> fp32 and a
// division per element -- made this launch cost as much as a dense layer for a quarter of its gathers)
template <typename TI> struct Raw<TI, true> : Raw<float, false> {};
// geometry of one kernel instance: columns per lane, lanes per row, neighbour rows per wave-instruction
template <int D, typename TI, bool SPARSE> struct Geo {
    static constexpr int CPL = Raw<TI, SPARSE>::CPL, LPR = D / CPL, NPW = 64 / LPR;
    typedef typename VecF<CPL>::T Acc;
};

// ---------------------------------------------------------------------------------
// One CSR row segment [start,end) of  A_hat * X  computed by one wavefront.
//
// The segment is walked in tiles of 64 non-zeros.  Per tile: every lane loads one
// (col,val) pair -- one coalesced 256-byte read of each CSR stream -- and the tile is
// staged in a wave-private LDS slot.  Then LPR lanes cover one embedding row with 16
// bytes each and the wave's NPW = 64/LPR lane groups take the staged neighbours
// interleaved, U deep, so every lane has up to U independent row gathers in flight
// behind ONE index round trip.  SPARSE: only neighbours whose row is flagged in the
// bitmap are staged (ballot + prefix-popcount compaction), the rest cost no gather.
// Partial sums are combined across the lane groups in a fixed order: deterministic,
// and identical wherever this function is used.
// ---------------------------------------------------------------------------------
// stage one tile of n <= 64 (col,val) pairs held one per lane; returns the staged count.  The tile is followed by
// zero-weight entries (column 0) up to TILE_PAD positions past its end, so that the gather batches below read
// past-the-end slots WITHOUT a per-gather clamp / compare / select (3 vector instructions per gathered row):
// positions [cnt, 64) are written here, positions [64, 64 + TILE_PAD) once per wave by tile_pad_init().
#define TILE_PAD 64           /* >= 4 * NPW of every geometry (d = 32 with a bf16 table: NPW = 16) */
#define TILE_ST (64 + TILE_PAD)
__device__ __forceinline__ void tile_pad_init(int2 *stage, int lane) { stage[64 + lane] = make_int2(0, 0); }
template <bool SPARSE>
__device__ __forceinline__ int tile_stage(int col, float val, int n, const GatherSrc &src, int lane, int2 *stage) {
    if (!SPARSE) {
        stage[lane] = lane < n ? make_int2(col, __float_as_int(val)) : make_int2(0, 0);
        return n;
    }
    const bool act = (lane < n) && bit_set(src.bm, col);
    const unsigned long long mask = __ballot(act);
    const int below = __builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0));
    const int cnt = __popcll(mask);
    // flagged entries at their compacted position, the others behind them as zero-weight padding: all 64 slots written
    stage[act ? below : cnt + (lane - below)] = act ? make_int2(col, __float_as_int(val)) : make_int2(0, 0);
    return cnt;
}

// One batch of U gathers per lane: entries j0+g, j0+g+NPW, ... of the staged tile (past the end: zero-weight padding).
template <int D, typename TI, bool SPARSE, int U, bool BIG>
__device__ __forceinline__ void gather_batch(const int2 *stage, int j0, const GatherSrc &src, int lane,
                                             typename Geo<D, TI, SPARSE>::Acc &acc) {
    typedef Geo<D, TI, SPARSE> G;
    typedef Raw<TI, SPARSE> R;
    const int g = lane / G::LPR, l = lane % G::LPR;
    int2 cv[U]; typename R::T x[U];
    const int2 *p = stage + j0 + g;
#pragma unroll
    for (int u = 0; u < U; u++) cv[u] = p[u * G::NPW];                   // all LDS reads first (one base, immediate offsets)
#pragma unroll
    for (int u = 0; u < U; u++) x[u] = R::template load<BIG>(src, cv[u].x, D, l);      // U gathers in flight
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int u = 0; u < U; u++) acc += __int_as_float(cv[u].y) * R::cvt(x[u], src.div);
}

// Gather-accumulate the staged tile.  The batch depth follows the (wave-uniform) number of staged
// neighbours instead of always issuing the deepest batch: a 64-lane gather instruction costs the
// CU's address path the same whether 1 or all of its row slots are useful.
template <int D, typename TI, bool SPARSE, bool BIG, int UCAP = LGCN_GATHER_U>
__device__ __forceinline__ void tile_gather(const int2 *stage, int cnt, const GatherSrc &src, int lane,
                                            typename Geo<D, TI, SPARSE>::Acc &acc) {
    constexpr int NPW = Geo<D, TI, SPARSE>::NPW, UMAX = SPARSE ? 4 : UCAP;
    static_assert(4 * NPW <= TILE_PAD, "the padding behind a tile must cover the deepest batch's overshoot");
    cnt = __builtin_amdgcn_readfirstlane(cnt);
    int j = 0;
    if (UMAX >= 8) for (; cnt - j > 4 * NPW; j += 8 * NPW) gather_batch<D, TI, SPARSE, (UMAX >= 8 ? 8 : UMAX), BIG>(stage, j, src, lane, acc);
    if (UMAX >= 4) for (; cnt - j > 2 * NPW; j += 4 * NPW) gather_batch<D, TI, SPARSE, (UMAX >= 4 ? 4 : UMAX), BIG>(stage, j, src, lane, acc);
    for (; cnt - j > NPW; j += 2 * NPW) gather_batch<D, TI, SPARSE, 2, BIG>(stage, j, src, lane, acc);
    if (cnt - j > 0) gather_batch<D, TI, SPARSE, 1, BIG>(stage, j, src, lane, acc);
}

// Sum over the lanes that hold the same columns (lane % LPR equal): every lane ends with the full
// sum.  VALU only -- v_permlane32_swap / v_permlane16_swap (gfx950) exchange wave halves / 16-lane rows
// without the LDS crossbar, DPP row rotations finish inside a row.  Fixed order: bitwise reproducible.
__device__ __forceinline__ float sum_xor32(float v) {
    typedef __attribute__((ext_vector_type(2))) unsigned u32x2;
    const unsigned a = __float_as_uint(v);
    const u32x2 r = __builtin_amdgcn_permlane32_swap(a, a, false, false);
    return __uint_as_float(r.x) + __uint_as_float(r.y);
}
__device__ __forceinline__ float sum_xor16(float v) {
    typedef __attribute__((ext_vector_type(2))) unsigned u32x2;
    const unsigned a = __float_as_uint(v);
    const u32x2 r = __builtin_amdgcn_permlane16_swap(a, a, false, false);
    return __uint_as_float(r.x) + __uint_as_float(r.y);
}
template <int ROT> __device__ __forceinline__ float sum_ror(float v) {      // + the lane ROT positions lower in the 16-lane row (cyclic)
    return v + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x120 + ROT, 0xf, 0xf, false));
}
template <int LPR, typename Acc>
__device__ __forceinline__ Acc reduce_groups(Acc acc) {
    constexpr int C = sizeof(Acc) / sizeof(float);
#pragma unroll
    for (int i = 0; i < C; i++) {
        float v = acc[i];
        if (LPR <= 32) v = sum_xor32(v);
        if (LPR <= 16) v = sum_xor16(v);
        if (LPR <= 8) v = sum_ror<8>(v);
        if (LPR <= 4) v = sum_ror<4>(v);
        acc[i] = v;
    }
    return acc;   // every lane group holds the full sum for its columns
}

// where a wave reads the (col,val) pairs of a row segment from: the CSR arrays (two 4-byte loads per
// entry) or the graph plan's packed stream (one 8-byte load)
struct CsrSrc {
    const int32_t *indices; const float *vals;
    __device__ __forceinline__ int2 at(int64_t i) const { return make_int2(indices[i], __float_as_int(vals[i])); }
};
struct PackedSrc {
    const int2 *pk;
    __device__ __forceinline__ int2 at(int64_t i) const { return pk[i]; }
};

template <int D, typename TI, bool SPARSE, bool BIG, typename ES>
__device__ __forceinline__ typename Geo<D, TI, SPARSE>::Acc
row_gather(const ES &es, int64_t start, int64_t end, const GatherSrc &src, int lane, int2 *stage) {
    typedef Geo<D, TI, SPARSE> G;
    typename G::Acc acc = zerov<G::CPL>();
    tile_pad_init(stage, lane);
    // the next tile's (col,val) pairs are in flight while this tile gathers
    int2 cv = make_int2(0, 0);
    if (start + lane < end) cv = es.at(start + lane);
    for (int64_t base = start; base < end; base += 64) {
        const int n = (int)min((int64_t)64, end - base);
        if (!SPARSE && src.S) cv.y = __float_as_int(__int_as_float(cv.y) * src.S[cv.x]);     // fp8 table: weight x row scale of the column
        const int cnt = tile_stage<SPARSE>(cv.x, __int_as_float(cv.y), n, src, lane, stage);
        __builtin_amdgcn_wave_barrier();
        cv = make_int2(0, 0);
        if (base + 64 + lane < end) cv = es.at(base + 64 + lane);
        tile_gather<D, TI, SPARSE, BIG>(stage, cnt, src, lane, acc);
        __builtin_amdgcn_wave_barrier();
    }
    return reduce_groups<G::LPR>(acc);
}

// ---------------------------------------------------------------------------------
// Work plan of one graph (built on the host at lgcn_graph_create).
//
// The processing order of the rows is cut into 8 slices, one per XCD (hardware deals
// workgroups round-robin over the 8 XCDs -- speed-only observation -- so block b works on
// slice b & 7).  A slice's rows mostly gather rows of the same part of the graph
// (reorder.py), so the part of the table an XCD touches is a fraction of the whole and
// lives in that XCD's 4 MiB L2.  Inside a slice:
//   * rows with more than LONG_T (= one 64-entry tile) non-zeros run as independent waves at
//     the FRONT of the slice (they start first: their serial tile walk overlaps everything
//     else).  Rows up to LONG_CH non-zeros are finished by one wave; longer rows are cut into
//     chunks of LONG_CH (a single wave walking Gowalla's 1415-nnz row = 23 serial tiles was
//     the critical path = the entire 57 us of an un-split launch): each chunk writes a partial
//     row and takes a ticket, and the LAST arriver sums the partials in chunk order (fixed
//     order: bitwise reproducible whoever is last) and runs the epilogue.  The hand-off is not
//     free (write-through stores, a drain, an atomic round trip, an L1 invalidate), which is
//     why the chunk is 512 and not one tile.  Long rows stay in the slice of their part of the
//     graph: they are 29 % of Gowalla's non-zeros, and dealt round-robin over the XCDs (round 1)
//     they alone produced half of the L2 misses.
//   * the other rows go NPW at a time per wave, ONE ROW PER LANE GROUP (NPW = 64 / lanes per row:
//     4 fp32 rows or 8 bf16 rows at d = 64), 4 waves per workgroup, in order.  The plan sorts every
//     window of SHORT_WIN consecutive short rows by length, so the rows of a pack are about equally
//     long (the window is far smaller than what an XCD has in flight: locality is unaffected).
// ---------------------------------------------------------------------------------
#define LONG_T 64             /* rows with more non-zeros than one tile leave the short path */
#define SLICE_PAD 64          /* short rows of a slice are padded to a multiple of this (>= rows per workgroup of every variant) */
#ifndef SHORT_WIN
#define SHORT_WIN 2048        /* window of the length sort (measured 128 / 512 / 1024 / 2048 / 4096 / 8192: 6082 / 6330 / 6403 / 6400 / 6378 / 6329 steps/s) */
#endif
#ifndef LONG_CH
#define LONG_CH 512           /* measured on Gowalla: 64 -> 58 us, 128 -> 40, 256 -> 34, 512 -> 32.5, 768 -> 39, none -> 57 */
#endif
#define XCDS 8
// The (col,val) pairs of the planned rows are kept a second time as ONE packed stream of 8-byte entries in
// plan order (chunks of a slice, then its short rows): the rows of a pack are contiguous there, so a wave
// loads a whole pack's index tiles with one or two 512-byte instructions instead of two 4-byte loads per
// row.  What bounds these kernels is the number of vector-memory instructions a CU retires (a 13-lane index
// load holds its address/return path as long as a 64-lane 1-KiB row gather, 26-30 cycles per instruction
// measured on both the fp32 and the bf16 table), so every instruction saved is time saved.
struct LongPlan {
    const int4 *chunks;           // [n_chunk_slots] (index o into long_row | -1 = padding, first entry, end entry in the stream, ordinal in row)
    const int32_t *long_row;      // [n_long] row ids with nnz > LONG_T
    const int32_t *long_nch;      // [n_long] chunks of that row
    float *partials;              // [n_chunk_slots, D]  (slot = position in `chunks`)
    int32_t *counters;            // [n_long] arrival tickets (zero between launches)
    int32_t n_long, n_chunk_slots;
};
struct SlicePlan {                // per XCD slice x: chunk blocks [cblk[x], cblk[x+1]) then short rows [rows[x], rows[x+1]) of rowinfo
    int32_t cblk[XCDS + 1];
    int32_t rows[XCDS + 1];       // multiples of SLICE_PAD
};

struct SpmmArgs {
    const int2 *pk;               // packed (col, val bits) stream in plan order
    const int4 *rowinfo;          // short rows of the slices, padded per slice: (row | -1, first entry in the stream, count, 0)
    LongPlan lp; SlicePlan sp;
    const void *X; void *Y;
    int64_t n_rows;               // rows of X (picks 32- or 64-bit gather offsets)
    long long *G64; uint32_t *bitmap; float div;   // sparse gradient rows (fixed point), K+1
    const float *G32;             // their fp32 copy Gs = G64 / 2^50 / (K+1), valid on the flagged rows (k_g32)
    float *P; float *M; float *V;
    bf16_t *Pb;                   // optional bf16 shadow of P written by the Adam epilogue
    void *Pq;                     // optional fp8 shadow of P (rows + scales), same
    // last kernel of a step (K >= 2): its epilogue zeroes the G64 rows / bitmap bits it consumes and one
    // wave reduces the per-triplet loss terms, so no separate clean-up launch is needed
    int clear;
    const float *terms; const float *gathered; float *loss_out; int32_t B, shard; float decay;
    float ent_coeff;              // popularity gate: coefficient of the gates' entropy term in the loss (0: no gate)
    int64_t blk;                  // data parallel: floats per rank in `gathered` (0: the default layout 3*shard*D + 2*shard)
    float step_size, bc2_sqrt, w1, beta2, omb2, eps;
    int remap;
    const float *selfX;           // M_ADDSELF: Y[row] = selfX[row] + (A X)[row]  (item-item smoothing, model.py:228-229)
    int32_t *cnt; float lam;      // reg_ego: slots of the batch naming each row (zeroed as consumed), decay / B: grad += lam * cnt[row] * P[row]
};

enum { M_SPARSE = 1, M_ADDG = 2, M_ADAM = 4, M_ADDSELF = 8 };

//   M_ADDG  : add Gs[row] where flagged (Horner term)
//   M_ADAM  : apply torch.optim.Adam to P/M/V with grad = result, else store to Y
// Adam's operands of one row piece, fetched under the LAST gather batch of a pack (see pack_batch)
template <int C> struct AdamPre { typename VecF<C>::T p, m, v; bool have; };

// IL: the accumulators are the four interleaved chunks of an fp8 gather (see fp8_t): the fp32 rows are addressed chunk by chunk
template <int D, typename TO, int MODE, int C, bool IL = false>
__device__ __forceinline__ void spmm_epilogue(const SpmmArgs &a, int64_t row, int l, typename VecF<C>::T acc, bool flagged,
                                              const AdamPre<C> *pre = nullptr) {
    typedef typename VecF<C>::T V;
    const int64_t off = row * D + l * C;
    if (MODE & M_ADDSELF) acc = row_load<D, C, IL>(a.selfX, row, l) + acc;
    if ((MODE & M_ADDG) && flagged) {      // (the row's bitmap word was fetched before the gathers -- a dependent load here measured +0.3-0.6 % on the step)
        V g = row_load<D, C, IL>(a.G32, row, l);          // = (float)(G64 * 2^-50) / (K+1), converted once by k_g32
        acc = g + acc;
        if ((MODE & M_ADAM) && !(MODE & M_SPARSE) && a.clear) {      // consumed: leave the workspace clean
            // (the bitmap is NOT cleared here: an atomic on words that every row's epilogue reads keeps
            //  dropping those lines from L2 -- measured +22 us; the two bitmaps alternate per step and
            //  k_triplet of the next step zeroes the stale one with plain stores)
            if constexpr (IL) {
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    i64x2 *q = reinterpret_cast<i64x2 *>(a.G64 + row * D + (j * (D / 16) + l) * 4);
                    q[0] = i64x2{0, 0}; q[1] = i64x2{0, 0};
                }
            } else {
                i64x2 *q = reinterpret_cast<i64x2 *>(a.G64 + off);
#pragma unroll
                for (int i = 0; i < C / 2; i++) q[i] = i64x2{0, 0};
            }
        }
    }
    if (MODE & M_ADAM) {
        V p, m, v;
        if (pre && pre->have) { p = pre->p; m = pre->m; v = pre->v; }
        else { p = row_load<D, C, IL>(a.P, row, l); m = row_load<D, C, IL>(a.M, row, l); v = row_load<D, C, IL>(a.V, row, l); }
        if ((MODE & M_ADDG) && a.cnt && flagged) {
            // upstream LightGCN's L2 term (cfg.reg_ego): d(decay * reg)/dE0[row] = decay/B * (slots naming the row) * E0[row],
            // added to the propagated gradient here, where E0[row] is in registers anyway
            const int n_slots = a.cnt[row];
            acc = acc + (a.lam * (float)n_slots) * p;
            if (l == 0) a.cnt[row] = 0;                 // consumed (all lanes of the row belong to this wave: read before write)
        }
        m = m + a.w1 * (acc - m);                       // exp_avg.lerp_(grad, 1-beta1)
        v = v * a.beta2 + (a.omb2 * acc) * acc;         // mul_(beta2).addcmul_(g,g,1-beta2)
        V denom;
#pragma unroll
        for (int i = 0; i < C; i++) denom[i] = sqrtf(v[i]) / a.bc2_sqrt + a.eps;
        p = p - a.step_size * (m / denom);              // addcdiv_(exp_avg, denom, -step_size)
        row_store<D, C, IL>(a.P, row, l, p); row_store<D, C, IL>(a.M, row, l, m); row_store<D, C, IL>(a.V, row, l, v);
        if (a.Pb) storev<C>(a.Pb + off, p);          // (bf16 shadow: only with bf16 tables, never IL)
        if (a.Pq) store_row_fp8<D, C, IL>(a.Pq, a.n_rows, row, l, p);
    } else {
        RowStore<D, C, TO, IL>::put(a.Y, a.n_rows, row, l, acc);
    }
}

// deterministic reduction of the per-triplet loss / reg terms by ONE wave (fixed strided
// partials, then an xor-shuffle tree): loss_out = {bpr + decay*reg, bpr, reg}   (model.py:168-173)
__device__ __forceinline__ void reduce_loss_wave(const float *terms, const float *gathered, int B, int shard,
                                                 int D, float decay, float *loss_out, int lane, float ent_coeff = 0.f, int64_t blk_in = 0) {
    // Every lane adds its terms b = lane, lane + 64, ... in that order (the order is part of the result).
    float fl = 0.f, fr = 0.f, fe = 0.f;
    if (ent_coeff != 0.f) {        // popularity gate: entropy of the 2B gates, one sum per triplet (model.py:176-181)
        if (!gathered) { for (int b = lane; b < B; b += 64) fe += terms[2 * B + b]; }
        else for (int b = lane; b < B; b += 64) fe += gathered[(int64_t)(b / shard) * blk_in + (int64_t)3 * shard * D + 2 * shard + b % shard];
    }
    if (!gathered) {
        // one GPU: the plain loop (deeper explicit batches made the launch it rides in slower: 3214-3278 steps/s at
        // B = 8192 for 32 ... 4 loads in flight, 3296 for this form)
#pragma unroll 8
        for (int b = lane; b < B; b += 64) { fl += terms[b]; fr += terms[B + b]; }
    } else {
        // data parallel: the terms of the GLOBAL batch lie in the ranks' blocks.  16 loads in flight: with one load per
        // round trip a 16 384-triplet batch (8 ranks) took 100 us and held the +Adam launch back (38 -> 108 us).
        // (r, i) = (block, position in it) of this lane's next term, advanced without divisions.
        constexpr int UL = 16;
        const int64_t blk = blk_in ? blk_in : (int64_t)3 * shard * D + 2 * shard;
        int r = lane / shard, i = lane % shard;
        for (int b0 = 0; b0 < B; b0 += 64 * UL) {
            float tl[UL], tr[UL];
#pragma unroll
            for (int u = 0; u < UL; u++) {
                const bool in = b0 + u * 64 + lane < B;
                const float *t = gathered + (in ? r : 0) * blk + (int64_t)3 * shard * D;      // past the end: a valid address, weight 0
                const int ii = in ? i : 0;
                tl[u] = t[ii]; tr[u] = t[shard + ii];
                i += 64;
                while (i >= shard) { i -= shard; r++; }
            }
#pragma unroll
            for (int u = 0; u < UL; u++) {
                const bool in = b0 + u * 64 + lane < B;
                fl += in ? tl[u] : 0.f; fr += in ? tr[u] : 0.f;
            }
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { fl += __shfl_xor(fl, off); fr += __shfl_xor(fr, off); fe += __shfl_xor(fe, off); }
    if (lane == 0) {
        const float bpr = -(fl / (float)B) - ent_coeff * (fe / (float)(2 * B));       // loss - coeff * entropy.mean()
        const float reg = (0.5f * fr) / (float)B;
        loss_out[0] = bpr + decay * reg; loss_out[1] = bpr; loss_out[2] = reg;
    }
}

// Y = [Gs +] A_hat X.  256-thread workgroups = 4 waves.  Block b works on slice b & 7 (b >> 3 -th
// block of it): first the slice's long-row chunks, one per wave, then its short rows.
//   M_SPARSE: X is Gs, read from the fixed-point table G64 for rows flagged in `bitmap`
#ifndef SPMM_MIN_WAVES
#define SPMM_MIN_WAVES 6      /* waves per SIMD the register allocation must allow (<= 80 VGPRs): a wave keeps up to
                                 NPW x 8 row gathers in flight, so residency is not what hides the latency */
#endif
#ifndef SPMM_MIN_WAVES_ADAM
#define SPMM_MIN_WAVES_ADAM 5    /* the fp32 +Adam variant holds P/M/V pieces across its last batch: 84 VGPRs */
#endif
#ifndef SPMM_ADAM_PREFETCH
#define SPMM_ADAM_PREFETCH 1   /* Adam's P/M/V of a short row are loaded under the pack's last gather batch */
#endif
#ifndef SPMM_U_SP
#define SPMM_U_SP 4           /* the same for the sparse first backward layer (few flagged neighbours per row) */
#endif
#ifndef SPMM_U
#define SPMM_U 8              /* gathers in flight per lane in the short-row path */
#endif
// One batch of U gathers of a lane group walking ITS OWN row: the next U of its entries, p[0], p[GPR], ... (GPR lane
// groups share a row and take alternate entries).  Past the row's end the staged row continues with zero-weight
// entries (written once per pack), so a gather costs its address, its load and its multiply-adds and nothing else
// (before: clamp, select, compare, select, LDS address and a 64-bit address per gathered row -- 9 of the 12 / 21 vector
// instructions per fp32 / bf16 gather; loads are unconditional either way: a predicated load makes hipcc wait).
template <int D, typename TI, bool SPARSE, int U, int GPR, bool PRE, bool BIG, int PC>
__device__ __forceinline__ void pack_batch(const int2 *p, const GatherSrc &src, int l,
                                           typename Geo<D, TI, SPARSE>::Acc &acc,
                                           const SpmmArgs *a, int64_t off, bool final_batch,
                                           AdamPre<PC> *pre) {
    typedef Raw<TI, SPARSE> R;
    int2 cv[U]; typename R::T xr[U];
#pragma unroll
    for (int u = 0; u < U; u++) cv[u] = p[u * GPR];
#pragma unroll
    for (int u = 0; u < U; u++) xr[u] = R::template load<BIG>(src, cv[u].x, D, l);
    if (PRE && final_batch && off >= 0) {       // the pack's last gathers are in flight: Adam's operands ride the same round trip
        pre->p = loadv<PC>(a->P + off); pre->m = loadv<PC>(a->M + off); pre->v = loadv<PC>(a->V + off);
        pre->have = true;
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int u = 0; u < U; u++) acc += __int_as_float(cv[u].y) * R::cvt(xr[u], src.div);
}

// rows per workgroup of a kernel variant: 4 waves x PACKS packs x NPW rows
#ifndef SPMM_WAVE_ROWS
#define SPMM_WAVE_ROWS 1      /* a wave walks packs one after the other until it has done this many rows (1: one pack per wave --
                                 measured: 4 -> 1 takes the d = 128 layer from 221 to 199 us; more, shorter waves win) */
#endif
#ifndef SPMM_WPB
#define SPMM_WPB 1            /* waves per workgroup (1, 2 or 4; the plan pads a slice's chunks to 4 and its rows to 64).  One-wave
                                 workgroups measured best: 4 -> 2 -> 1 = 5650 -> 5890 -> 5965 steps/s, dense layer 28.7 -> 27.5 -> 26.6 us */
#endif
template <int NPW> struct PackGeo { static constexpr int PACKS = NPW >= SPMM_WAVE_ROWS ? 1 : SPMM_WAVE_ROWS / NPW, RPW = NPW * PACKS, RPB = SPMM_WPB * RPW; };
// lane groups that share one short row.  A bf16 table row is 8 lanes wide, so 8 rows fit a wave -- but more,
// shorter waves are what this kernel wants (measured): two groups per bf16 row = 4 rows per wave like fp32.
#ifndef SPMM_GPR_BF16
#define SPMM_GPR_BF16 2
#endif
#ifndef SPMM_SCALAR_ROWINFO
#define SPMM_SCALAR_ROWINFO 1  /* a pack's row descriptors and bitmap words through the scalar cache instead of vector loads + readlane */
#endif
#ifndef SPMM_ADAM_SPLIT
#define SPMM_ADAM_SPLIT 1     /* bf16 tables: the two lane groups of a row share its Adam epilogue (see k_spmm) */
#endif
#ifndef SPMM_ADAM_PREFETCH_BF16
#define SPMM_ADAM_PREFETCH_BF16 0
#endif
#ifndef SPMM_GPR_F32
#define SPMM_GPR_F32 1
#endif
template <int D, typename TI, bool SP> struct RowGeo {
    static constexpr int NPW = Geo<D, TI, SP>::NPW;
    // (an fp8 row is D / 16 lanes wide: NPW / 4 groups per row keep 4 rows per wave at every d)
    static constexpr int WANT = SP ? 1 : (Geo<D, TI, SP>::CPL == 16 ? (NPW >= 4 ? NPW / 4 : 1) : Geo<D, TI, SP>::CPL == 8 ? SPMM_GPR_BF16 : SPMM_GPR_F32);
    static constexpr int GPR = NPW >= WANT ? WANT : NPW;
    static constexpr int RPK = NPW / GPR;          // rows of a pack
};
// v[lane] + v[lane ^ O], VALU only
template <int O> __device__ __forceinline__ float sum_xor(float v, int lane) {
    if (O == 32) return sum_xor32(v);
    if (O == 16) return sum_xor16(v);
    if (O == 8) return sum_ror<8>(v);
    // O == 4: row_ror moves data to HIGHER lanes (lane i receives lane i - n): the partner is 4 below or 4 above
    const float dn = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x120 + 4, 0xf, 0xf, false));
    const float up = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x120 + 12, 0xf, 0xf, false));
    return v + ((lane & 4) ? dn : up);
}
// sum over the GPR lane groups of a row (groups LPR lanes apart): every one of them ends with the total
template <int LPR, int GPR> __device__ __forceinline__ float sum_row_groups(float v, int lane) {
    if (GPR >= 2) v = sum_xor<LPR>(v, lane);
    if (GPR >= 4) v = sum_xor<(2 * LPR > 32 ? 32 : 2 * LPR)>(v, lane);
    return v;
}

template <int D, typename TI, typename TO, int MODE, bool BIG>
// (fp8 tables: 16 accumulators per lane and, with Adam, 3 x 16 operands: a 128-register budget, 4 waves per SIMD)
__global__ void __launch_bounds__(64 * SPMM_WPB, sizeof(TI) == 1 ? 4 : ((MODE & M_ADAM) && sizeof(TI) == 4) ? SPMM_MIN_WAVES_ADAM : SPMM_MIN_WAVES) k_spmm(SpmmArgs a) {
    constexpr bool SP = (MODE & M_SPARSE) != 0;
    typedef Geo<D, TI, SP> G;
    typedef typename G::Acc Acc;
    typedef Raw<TI, SP> R;
    constexpr int LPR = G::LPR, NPW = G::NPW, C = G::CPL;
    constexpr bool IL = !SP && sizeof(TI) == 1;        // accumulators of an fp8 gather: chunk-interleaved columns
    constexpr int GPR = RowGeo<D, TI, SP>::GPR, RPK = RowGeo<D, TI, SP>::RPK;
    static_assert(GPR == 1 || GPR == 2 || GPR == 4, "one, two or four lane groups per row");
    static_assert(LPR * GPR <= 64 && (GPR == 1 || LPR >= 4), "lane groups of a row must fit the wave");
    constexpr int PACKS = PackGeo<RPK>::PACKS, RPW = PackGeo<RPK>::RPW, RPB = PackGeo<RPK>::RPB;
    // stage row stride (entries): a row's <= 64 entries + the zero-weight tail the deepest batch may read (3 per lane
    // group of the row); lane groups reading the same position of different rows hit different banks (76 * 2 mod 64 = 24)
    constexpr int ST = 76;
    static_assert(64 + 3 * GPR <= ST, "staged row + padding");
    constexpr int U = SP ? SPMM_U_SP : (C == 16 ? 4 : SPMM_U);      // fp8: 16 accumulators per lane; 4 x 16 B in flight per lane
    __shared__ int2 stage_lds[SPMM_WPB][(RPK * ST > TILE_ST ? RPK * ST : TILE_ST)];
    const int lane = threadIdx.x & 63;
    const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    GatherSrc src;
    src.X = SP ? (const void *)a.G32 : a.X; src.bm = a.bitmap; src.div = a.div;
    src.S = (!SP && sizeof(TI) == 1) ? fp8_scales(a.X, a.n_rows, D) : nullptr;
    // (a per-workgroup LDS copy of the row bitmap -- 9 KiB on Gowalla -- was measured: no gain, the copy's
    //  own round trip per workgroup costs what the per-neighbour tests save once a pack's tests are batched)
    // (block 0 is dispatched first: the reduction overlaps the whole launch; on the last block it
    //  sat on the tail and cost +25 us)
    if ((MODE & M_ADAM) && !SP && a.clear && blockIdx.x == 0 && wid == SPMM_WPB - 1)
        reduce_loss_wave(a.terms, a.gathered, a.B, a.shard, D, a.decay, a.loss_out, lane, a.ent_coeff, a.blk);
    int x, j;
    if (a.remap) { x = blockIdx.x & (XCDS - 1); j = blockIdx.x >> 3; }
    else {        // slices as contiguous block ranges (placement-independent either way: speed only)
        x = 0; j = blockIdx.x;
        while (x < XCDS - 1) {
            const int nb = (a.sp.cblk[x + 1] - a.sp.cblk[x]) * (4 / SPMM_WPB) + (a.sp.rows[x + 1] - a.sp.rows[x]) / RPB;
            if (j < nb) break;
            j -= nb; x++;
        }
    }
    const int ncb = (a.sp.cblk[x + 1] - a.sp.cblk[x]) * (4 / SPMM_WPB);      // the plan counts groups of 4 chunks
    if (j < ncb) {
        // ---- one chunk of a long row, the whole wave on it ----
        const int c = a.sp.cblk[x] * 4 + j * SPMM_WPB + wid;
#if SPMM_SCALAR_ROWINFO
        // (the chunk's descriptor, its row and the row's bitmap word are wave-uniform plan data: scalar loads, as for the packs below)
        typedef int i32x4c_ __attribute__((ext_vector_type(4)));
        const i32x4c_ chv = ((const __attribute__((address_space(4))) i32x4c_ *)a.lp.chunks)[c];
        const int4 ch = make_int4(chv.x, chv.y, chv.z, chv.w);
        const int o = ch.x;
        if (o < 0) return;
        const int64_t row = ((const __attribute__((address_space(4))) int32_t *)a.lp.long_row)[o];
        const int nch = ((const __attribute__((address_space(4))) int32_t *)a.lp.long_nch)[o];
        const bool rflag = (MODE & M_ADDG) ? ((((const __attribute__((address_space(4))) uint32_t *)a.bitmap)[row >> 5] >> (row & 31)) & 1u) != 0u : false;
#else
        const int4 ch = a.lp.chunks[c];
        const int o = ch.x;
        if (o < 0) return;
        const int64_t row = a.lp.long_row[o];
        const int nch = a.lp.long_nch[o];
        const bool rflag = (MODE & M_ADDG) ? bit_set(a.bitmap, (int)row) : false;
#endif
        Acc acc = row_gather<D, TI, SP, BIG>(PackedSrc{a.pk}, ch.y, ch.z, src, lane, stage_lds[wid]);
        if (nch == 1) {                                   // LONG_T < nnz <= LONG_CH: one wave, no hand-off
            if (lane < LPR) spmm_epilogue<D, TO, MODE, C, IL>(a, row, lane, acc, rflag);
            return;
        }
        // Publish the partial WRITE-THROUGH (sc1: 8-byte agent-scope stores, no release fence -- a
        // release would write back this XCD's whole dirty L2, measured 2x on the launch), drain, take
        // a ticket; the last arriver invalidates its L1 once and reads the partials with sc1 loads
        // (cdna guide G16: R1 form with a counter, both the acquire AND the sc1-load variant).
        typedef __attribute__((address_space(1))) unsigned long long gu64;
        if (lane < LPR) {
            union { Acc v; unsigned long long q[C / 2]; } pk; pk.v = acc;
            gu64 *dst = (gu64 *)(a.lp.partials + (int64_t)c * D + lane * C);
#pragma unroll
            for (int i = 0; i < C / 2; i++) __hip_atomic_store(dst + i, pk.q[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        int ticket = 0;
        if (lane == 0) ticket = __hip_atomic_fetch_add(a.lp.counters + o, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        ticket = __builtin_amdgcn_readfirstlane(ticket);
        if (ticket != nch - 1) return;
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (lane == 0) __hip_atomic_store(a.lp.counters + o, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // next launch
        if (lane >= LPR) return;
        const int first = c - ch.w;
        Acc tot = zerov<C>();
        for (int k = 0; k < nch; k++) {
            union { Acc v; unsigned long long q[C / 2]; } pk;
            gu64 *sp_ = (gu64 *)(a.lp.partials + (int64_t)(first + k) * D + lane * C);
#pragma unroll
            for (int i = 0; i < C / 2; i++) pk.q[i] = __hip_atomic_load(sp_ + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (k == 0) tot = pk.v; else tot += pk.v;
        }
        spmm_epilogue<D, TO, MODE, C, IL>(a, row, lane, tot, rflag);
        return;
    }
    // ---- short rows (<= 64 non-zeros): a pack of NPW rows at a time, ONE ROW PER LANE GROUP.  All
    //      index tiles of the pack are loaded back to back, then every group walks its own row U
    //      gathers deep: NPW x U row gathers in flight per wave (32 at d = 64 fp32, 64 with a bf16
    //      table) instead of one row's worth, no cross-group reduction, no divergence (the trip
    //      count is the pack's longest row; shorter rows pad with zero-weight re-reads of their
    //      last entry, and the plan sorts windows of rows by length so there is little to pad).
    //      A row is summed in CSR order by one accumulator chain -- the reference's own order.
    const int t = j - ncb;
    if (t >= (a.sp.rows[x + 1] - a.sp.rows[x]) / RPB) return;
    const int g = lane / LPR, l = lane % LPR;
    int2 *stage = stage_lds[wid];
#pragma unroll 1
    for (int pk = 0; pk < PACKS; pk++) {
        const int64_t pos0 = (int64_t)a.sp.rows[x] + (int64_t)t * RPB + (wid * PACKS + pk) * RPK;
        int off[RPK + 1];
        off[0] = 0;
#if SPMM_SCALAR_ROWINFO
        // The pack's row descriptors (row id, first entry, count) are the same for every lane and constant during the launch:
        // SCALAR loads (constant address space -> s_load_dwordx4), no vector-memory instruction and no readlane for them -- these
        // kernels are bound by the vector-memory instructions a CU retires (experiments.txt run 29-31).  The rows' bitmap words
        // likewise (k_spmm only reads the bitmap).
        typedef int i32x4_ __attribute__((ext_vector_type(4)));          // (a plain vector type: HIP's int4 is a class on the host pass)
        typedef const __attribute__((address_space(4))) i32x4_ *kint4p;
        typedef const __attribute__((address_space(4))) uint32_t *ku32p;
        const kint4p rip = (kint4p)(a.rowinfo + pos0);
        int row_s[RPK];
        int64_t base = 0;
#pragma unroll
        for (int r = 0; r < RPK; r++) {
            const i32x4_ ri = rip[r];
            row_s[r] = ri.x; off[r + 1] = off[r] + ri.z;
            if (r == 0) base = ri.y;
        }
        if (row_s[0] < 0) break;                                   // padding is at the end of a slice
        uint32_t fw_s[RPK];
#pragma unroll
        for (int r = 0; r < RPK; r++) fw_s[r] = ((MODE & M_ADDG) && row_s[r] >= 0) ? ((ku32p)a.bitmap)[row_s[r] >> 5] : 0u;
        const int tot = off[RPK];
#else
        // lane r < RPK fetches (row id, first entry, count) of row r with ONE 16-byte load from the plan
        int my_row = -1, my_s = 0, my_n = 0;
        if (lane < RPK) {
            const int4 ri = a.rowinfo[pos0 + lane];
            my_row = ri.x; my_s = ri.y; my_n = ri.z;
        }
        uint32_t my_fw = 0u;           // bitmap word of this lane's row, in flight under the stream loads
        if ((MODE & M_ADDG) && my_row >= 0) my_fw = a.bitmap[my_row >> 5];
        if (__builtin_amdgcn_readlane(my_row, 0) < 0) break;       // padding is at the end of a slice
#pragma unroll
        for (int r = 0; r < RPK; r++) off[r + 1] = off[r] + __builtin_amdgcn_readlane(my_n, r);
        const int tot = off[RPK];
        const int64_t base = __builtin_amdgcn_readlane(my_s, 0);
#endif
        // the pack's rows are contiguous in the stream: entry e of the pack belongs to the row r with
        // off[r] <= e < off[r+1]; whole 512-byte loads, all issued before the first is staged
        int2 cvr[RPK];
#pragma unroll
        for (int it = 0; it < RPK; it++) {
            cvr[it] = make_int2(0, 0);
            if (it * 64 < tot) { const int e = it * 64 + lane; if (e < tot) cvr[it] = a.pk[base + e]; }
        }
        const int myr = g / GPR, sub = g % GPR;      // this lane group's row of the pack, and its share of it
        int maxcnt = 0;
        if (!SP && sizeof(TI) == 1) {                // fp8 table: weight x row scale of the column (padding entries: column 0, weight 0)
#pragma unroll
            for (int it = 0; it < RPK; it++)
                if (it * 64 < tot) cvr[it].y = __float_as_int(__int_as_float(cvr[it].y) * src.S[cvr[it].x]);
        }
        if (!SP) {
#pragma unroll
            for (int it = 0; it < RPK; it++) {
                if (it * 64 < tot) {
                    const int e = it * 64 + lane;
                    int r = 0;
#pragma unroll
                    for (int k = 1; k < RPK; k++) r += (e >= off[k]) ? 1 : 0;
                    int o_r = 0;
#pragma unroll
                    for (int k = 1; k < RPK; k++) o_r = (e >= off[k]) ? off[k] : o_r;
                    if (e < tot) stage[r * ST + (e - o_r)] = cvr[it];
                }
            }
#pragma unroll
            for (int r = 0; r < RPK; r++) maxcnt = max(maxcnt, (off[r + 1] - off[r] + GPR - 1) / GPR);
            // zero-weight tail of every row up to what the longest row's batches read (see pack_batch)
            const int padto = GPR * (maxcnt + 3);
#pragma unroll
            for (int r = 0; r < RPK; r++)
                for (int q = off[r + 1] - off[r] + lane; q < padto; q += 64) stage[r * ST + q] = make_int2(0, 0);
        } else {
            // keep only the neighbours whose row is flagged.  The bitmap words of ALL the pack's entries are
            // fetched in one round trip straight from the registers the stream landed in, and every entry is
            // staged at its compacted position in one pass: ballot, popcount of the flagged lanes below it
            // inside its own row's lane range, plus the row's count from the earlier 64-entry pieces (uniform).
            uint32_t w[RPK];
#pragma unroll
            for (int it = 0; it < RPK; it++) w[it] = (it * 64 < tot) ? src.bm[cvr[it].x >> 5] : 0u;      // col 0 past the end: valid
            int cntr[RPK];
#pragma unroll
            for (int r = 0; r < RPK; r++) cntr[r] = 0;
#pragma unroll
            for (int it = 0; it < RPK; it++) {
                if (it * 64 < tot) {
                    const int e = it * 64 + lane;
                    const bool flag = e < tot && ((w[it] >> (cvr[it].x & 31)) & 1u);
                    const unsigned long long m = __ballot(flag);
                    int r = 0, o_r = 0, before = cntr[0];
#pragma unroll
                    for (int k = 1; k < RPK; k++) { const bool ge = e >= off[k]; r += ge ? 1 : 0; o_r = ge ? off[k] : o_r; before = ge ? cntr[k] : before; }
                    const int lo = max(o_r - it * 64, 0);                 // first lane of this lane's row in this piece (<= lane)
                    const unsigned long long below = m & ((1ull << lane) - 1ull) & ~((1ull << lo) - 1ull);
                    if (flag) stage[r * ST + before + __popcll(below)] = cvr[it];
#pragma unroll
                    for (int k = 0; k < RPK; k++) {
                        const int lk = min(max(off[k] - it * 64, 0), 64), hk = min(max(off[k + 1] - it * 64, 0), 64);
                        const unsigned long long mk = (hk >= 64 ? ~0ull : (1ull << hk) - 1ull) & ~(lk >= 64 ? ~0ull : (1ull << lk) - 1ull);
                        cntr[k] += __popcll(m & mk);
                    }
                }
            }
#pragma unroll
            for (int r = 0; r < RPK; r++) maxcnt = max(maxcnt, cntr[r]);
            const int padto = GPR * (maxcnt + 3);
#pragma unroll
            for (int r = 0; r < RPK; r++)
                for (int q = cntr[r] + lane; q < padto; q += 64) stage[r * ST + q] = make_int2(0, 0);
        }
        __builtin_amdgcn_wave_barrier();
        maxcnt = __builtin_amdgcn_readfirstlane(maxcnt);
        const int2 *mystage = stage + myr * ST + sub;
        Acc acc = zerov<C>();
        // batch depth follows the pack's longest row (8 / 4 / 2 / 1 gathers per lane): a padded gather costs
        // the address path as much as a useful one
#if SPMM_SCALAR_ROWINFO
        int mrow = row_s[0];
        uint32_t mfw = fw_s[0];
#pragma unroll
        for (int r = 1; r < RPK; r++) { mrow = myr == r ? row_s[r] : mrow; mfw = myr == r ? fw_s[r] : mfw; }
#else
        const int mrow = __shfl(my_row, myr);
        const uint32_t mfw = (MODE & M_ADDG) ? (uint32_t)__shfl((int)my_fw, myr) : 0u;
#endif
        // (fp32 tables only: with a bf16 table the kernel is already at its register budget and the operands spill --
        //  measured 6940 vs 7420 steps/s; fp32: 6339 vs 6306)
        //  with the two lane groups of a bf16 row sharing the epilogue -- SPLIT below -- the operands are 12 registers, not 24)
        constexpr bool SPLIT = SPMM_ADAM_SPLIT && (MODE & M_ADAM) != 0 && C == 8 && GPR == 2 && !IL;
        constexpr bool PRE = SPMM_ADAM_PREFETCH && (MODE & M_ADAM) != 0 && (sizeof(TI) == 4 || (SPLIT && SPMM_ADAM_PREFETCH_BF16));
        constexpr int PC = SPLIT ? 4 : C;
        AdamPre<PC> pre; pre.have = false;
        const int64_t poff = (PRE && mrow >= 0 && (SPLIT || sub == 0)) ? (int64_t)mrow * D + (SPLIT ? (2 * l + sub) * 4 : l * C) : -1;
        int u0 = 0;
#define PACK_BATCH(UU) pack_batch<D, TI, SP, UU, GPR, PRE, BIG, PC>(mystage + GPR * u0, src, l, acc, &a, poff, maxcnt - u0 <= UU, &pre)
        if (U >= 8) for (; maxcnt - u0 > 4; u0 += 8) PACK_BATCH((U >= 8 ? 8 : U));
        if (U >= 4) for (; maxcnt - u0 > 2; u0 += 4) PACK_BATCH((U >= 4 ? 4 : U));
        for (; maxcnt - u0 > 1; u0 += 2) PACK_BATCH(2);
        if (maxcnt - u0 > 0) PACK_BATCH(1);
#undef PACK_BATCH
        if (GPR > 1) {
#pragma unroll
            for (int i = 0; i < C; i++) acc[i] = sum_row_groups<LPR, GPR>(acc[i], lane);     // fixed order: bitwise reproducible
        }
        const bool mflag = (MODE & M_ADDG) ? ((mfw >> (mrow & 31)) & 1u) != 0u : false;
        if constexpr (SPLIT) {
            // bf16 table, Adam: both lane groups of the row hold its sums, so BOTH run the fp32 epilogue, group `sub`
            // on the 4-column chunk 2l + sub.  One float4 per operand and lane, whole 64-byte lines per instruction
            // (8 columns per lane = two instructions that each touch every line of the row and use half of it);
            // the same arithmetic per element, so the same bits.
            if (mrow >= 0) {
                const f32x4 h = sub ? f32x4{acc[4], acc[5], acc[6], acc[7]} : f32x4{acc[0], acc[1], acc[2], acc[3]};
                spmm_epilogue<D, TO, MODE, 4, false>(a, mrow, 2 * l + sub, h, mflag, PRE ? &pre : nullptr);
            }
        } else
        if (mrow >= 0 && sub == 0) spmm_epilogue<D, TO, MODE, C, IL>(a, mrow, l, acc, mflag, PRE ? &pre : nullptr);
        if (PACKS > 1) __builtin_amdgcn_wave_barrier();
    }
}

// out = (X_0 + X_1 + ... + X_{K-1} + out) / (K+1), `out` holding X_K on entry -- the stack + mean of
// computer() (model.py:221-222) as one streaming pass.  The K-th layer itself runs through k_spmm
// like the others, so evaluation gets the split long rows too.
struct MeanArgs {
    const float *X0; const void *Xl[LGCN_MAX_LAYERS + 1]; int K;
    float *out; int64_t n4;      // number of 4-element pieces
    int32_t d; int64_t N;        // row width and rows of the tables (fp8 tables: where a piece's row scale lies)
};
// piece i (4 consecutive elements) of an [N,d] table of TI
template <typename TI> __device__ __forceinline__ f32x4 tab_load4(const void *tab, int64_t i, int d, int64_t N) { return load4((const TI *)tab + i * 4); }
template <> __device__ __forceinline__ f32x4 tab_load4<fp8_t>(const void *tab, int64_t i, int d, int64_t N) {
    const int64_t row = (i * 4) / d;
    const int w = *(const int *)((const char *)tab + row * d + fp8_byte_of_col((int)(i * 4 - row * d), d));      // a 4-column chunk is one aligned word
    const auto a = __builtin_amdgcn_cvt_pk_f32_fp8(w, false), b = __builtin_amdgcn_cvt_pk_f32_fp8(w, true);
    const float sc = ((const float *)((const char *)tab + N * d))[row];
    return f32x4{a[0] * sc, a[1] * sc, b[0] * sc, b[1] * sc};
}

template <typename TI>
__global__ void __launch_bounds__(256) k_layer_mean(MeanArgs a) {
    const float div = (float)(a.K + 1);
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < a.n4; i += (int64_t)gridDim.x * 256) {
        f32x4 s = load4(a.X0 + i * 4);
        for (int k = 1; k < a.K; k++) s += tab_load4<TI>(a.Xl[k], i, a.d, a.N);
        s += load4(a.out + i * 4);
        store4(a.out + i * 4, s / div);
    }
}

// bf16 copy of the parameter table: with bf16 activation storage (K >= 2) layer 1 gathers THIS instead of the fp32
// rows -- every SpMM input is then a 2-byte table.  Made at the start of a training call and kept current by the
// Adam epilogue (SpmmArgs::Pb) inside a multi-step call; evaluation (lgcn_propagate_mean) makes its own.
__global__ void __launch_bounds__(256) k_to_bf16(const float *src, bf16_t *dst, int64_t n4) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256)
        store4(dst + i * 4, load4(src + i * 4));
}
// fp8 copy of an fp32 table (rows + row scales): the input of layer 1 with fp8 activation storage.  One lane group of
// d / 16 lanes per row (64 B read, 16 B written per lane).
template <int D>
__global__ void __launch_bounds__(256) k_to_fp8(const float *src, void *dst, int64_t n_rows) {
    constexpr int LPR = D / 16, RPB = 256 / LPR;
    const int l = threadIdx.x % LPR;
    const int64_t rows_pad = (n_rows + RPB - 1) / RPB * RPB;          // whole lane groups stay together (the shuffles need them)
    for (int64_t r = (int64_t)blockIdx.x * RPB + threadIdx.x / LPR; r < rows_pad; r += (int64_t)gridDim.x * RPB) {
        const int64_t row = r < n_rows ? r : n_rows - 1;
        const f32x16 v = row_load<D, 16, true>(src, row, l);          // the lane's four chunks j*L + l: coalesced float4 loads
        if (r < n_rows) store_row_fp8<D, 16, true>(dst, n_rows, row, l, v);
        else { float amax = 0.f; for (int o = 1; o < LPR; o <<= 1) amax = fmaxf(amax, __shfl_xor(amax, o)); }   // (keep the group's shuffles matched)
    }
}
static int launch_to_fp8(const float *src, void *dst, int64_t n_rows, int d, hipStream_t st) {
    const int64_t blocks = (n_rows * (d / 16) + 255) / 256;
    const unsigned grid = (unsigned)(blocks < 8192 ? (blocks > 0 ? blocks : 1) : 8192);
    switch (d) {
    case 64: hipLaunchKernelGGL((k_to_fp8<64>), dim3(grid), dim3(256), 0, st, src, dst, n_rows); return 0;
    case 128: hipLaunchKernelGGL((k_to_fp8<128>), dim3(grid), dim3(256), 0, st, src, dst, n_rows); return 0;
    case 256: hipLaunchKernelGGL((k_to_fp8<256>), dim3(grid), dim3(256), 0, st, src, dst, n_rows); return 0;
    }
    lgcn_set_error("fp8 storage needs an embedding dim of 64, 128 or 256");
    return 3;
}
static void launch_to_bf16(const float *src, bf16_t *dst, int64_t n, hipStream_t st) {
    const int64_t n4 = n / 4, blocks = (n4 + 255) / 256;
    hipLaunchKernelGGL(k_to_bf16, dim3((unsigned)(blocks < 4096 ? blocks : 4096)), dim3(256), 0, st, src, dst, n4);
}

// ---------------------------------------------------------------------------------
// BPR on the batch.
//
// k_triplet (default): ONE 256-thread workgroup per triplet does everything the batch needs from it:
//   the three slot rows e = mean_k X_k[row] (user, positive, negative item; the last layer
//   X_K[row] = (A_hat X_{K-1})[row] is gathered on the fly), then the loss terms and the three gradient
//   rows (triplet_loss_regs).  A row is cut into units of ROWS_UNIT_TILES 64-entry tiles; unit u of slot c
//   belongs to wave (c + u) mod 4: a typical triplet costs one unit per slot on waves 0-2 and wave 3 ends at
//   once, while a hub positive (sampled proportionally to popularity: 1000-neighbour rows are common) is
//   shared by all four waves.  Partial rows meet in LDS in a fixed order; wave 0 finishes the triplet.
//   (Round 1/2 history: one workgroup per SLOT + a loss launch = 24.6K waves of which 18K ended after reading
//   two indices; the launch skeleton alone -- ids, indptr, nothing else -- measured 7.3 of its 16.7 us, and
//   the slot rows made a round trip through HBM to the 6.7 us loss launch.)
// k_triplet_dense: the same when the last layer was propagated densely (cfg.dense_last).
// ---------------------------------------------------------------------------------
struct BprArgs {
    const int32_t *indptr; const int32_t *indices; const float *vals;
    const float *X0; const void *Xl[LGCN_MAX_LAYERS + 1]; int K;
    int32_t n_users; int64_t N;
    const int32_t *users; const int32_t *pos; const int32_t *neg;   // already offset to the local shard
    int32_t B_local;      // triplets handled by this launch
    int32_t shard;        // row stride of the contrib block (>= B_local)
    float inv_B;          // 1 / global batch
    float lam;            // decay / global batch
    long long *G64;       // if non-null: atomics
    uint32_t *bitmap;
    uint32_t *stale_bitmap; int64_t bitmap_words;   // last step's bitmap: zeroed here (plain stores)
    uint32_t *item_bitmap;  // item-item smoothing: bit i = item i is named by the batch (or NULL)
    float *contrib;       // exchange block [3*shard*D | shard | shard] (data parallel)
    int32_t exchange;     // write the gradient rows and loss terms to `contrib` (instead of / besides the atomics)
    const float *Xhub; int32_t hub_nnz;     // rows with more than hub_nnz non-zeros: X_K[row] is read from Xhub (0: none)
    float *terms;         // atomics mode: [2*terms_stride] (loss terms | reg terms), this launch at terms_off
    int32_t terms_off, terms_stride;
    int32_t *err;
    int32_t dense_last;   // X_K exists densely (Xl[K]): the slot rows are read, not gathered
    int32_t reg_ego;      // L2 term on the tables' own rows X0[row] (upstream LightGCN) instead of the propagated rows (the fork)
    int32_t *cnt;         // reg_ego: per-row count of the batch slots naming the row (bumped where the row is flagged), or NULL
    // column-sharded data parallelism: this rank holds D of the table's columns.  cols_phase 1: the batch-row kernel stops after
    // the slot rows -- it writes them to erows [3][B_local][D] and the PARTIAL dot products / squared norms over its columns to
    // colsum [3][B_local] (pos score, neg score, reg term); the ranks all-reduce colsum; k_cols_finish does the rest.
    int32_t cols_phase; float *erows; float *colsum;
};

__device__ __forceinline__ float logsigmoid_f(float x) { return fminf(x, 0.f) - log1pf(expf(-fabsf(x))); }
__device__ __forceinline__ float sigmoid_neg_f(float x) {
    const float z = expf(-fabsf(x));
    return x < 0.f ? 1.f / (1.f + z) : z / (1.f + z);
}

__device__ __forceinline__ bool triplet_bad(const BprArgs &a, int b) {
    const int u = a.users[b], p = a.pos[b], n = a.neg[b];
    return u < 0 || u >= a.n_users || p < 0 || (int64_t)p + a.n_users >= a.N || n < 0 || (int64_t)n + a.n_users >= a.N;
}

// Loss terms and the three gradient rows of ONE triplet by one lane group.  Lane = column (mod 64): every
// load, store and atomic wave-instruction covers min(D,64) contiguous elements -- 512 contiguous bytes
// per 64-bit atomic instruction at d = 64.  (With 4 columns per lane the same atomics were 16 lanes x 8 B
// at a 32-byte stride and cost 9 of a 13.7 us kernel.)  x = e_u.e_p - e_u.e_n ; l = logsigmoid(x) ;
// r = |e_u|^2+|e_p|^2+|e_n|^2 ; gradient rows w.r.t. the propagated table -> fixed-point
// atomics into G64 + row flags (single GPU / dense DP) or the exchange block (DP rows).
template <int D>
__device__ __forceinline__ void triplet_finish(const BprArgs &a, int b, int l, const float *u, const float *p, const float *n,
                                               float ps, float ns, float rr);
// reg_ego: u0 / p0 / n0 = the slots' rows of the table itself; the L2 term is theirs and its gradient never enters G
// (the Adam epilogue adds it from the slot counts), so the gradient rows carry no lam term.
template <int D>
__device__ __forceinline__ void triplet_loss_regs(const BprArgs &a, int b, int l, const float *u, const float *p, const float *n,
                                                  const float *u0 = nullptr, const float *p0 = nullptr, const float *n0 = nullptr) {
    constexpr int LPT = D < 64 ? D : 64, CPL = D / LPT;
    float ps = 0.f, ns = 0.f, ru = 0.f, rp = 0.f, rn = 0.f;
    const bool ego = a.reg_ego != 0;
    const float lam = ego ? 0.f : a.lam;
#pragma unroll
    for (int j = 0; j < CPL; j++) {
        ps += u[j] * p[j]; ns += u[j] * n[j];
        const float ur = ego ? u0[j] : u[j], pr = ego ? p0[j] : p[j], nr = ego ? n0[j] : n[j];
        ru += ur * ur; rp += pr * pr; rn += nr * nr;
    }
    float rr = ru + rp + rn;
#pragma unroll
    for (int off = 1; off < LPT; off <<= 1) {
        ps += __shfl_xor(ps, off); ns += __shfl_xor(ns, off); rr += __shfl_xor(rr, off);
    }
    if (a.cols_phase == 1) {                    // column shard: rows + partial sums out, the all-reduce comes next
        const bool bad1 = triplet_bad(a, b);
        if (l == 0) { a.colsum[b] = bad1 ? 0.f : ps; a.colsum[a.B_local + b] = bad1 ? 0.f : ns; a.colsum[2 * a.B_local + b] = bad1 ? 0.f : rr; }
#pragma unroll
        for (int j = 0; j < CPL; j++) {
            a.erows[((int64_t)0 * a.B_local + b) * D + j * LPT + l] = u[j];
            a.erows[((int64_t)1 * a.B_local + b) * D + j * LPT + l] = p[j];
            a.erows[((int64_t)2 * a.B_local + b) * D + j * LPT + l] = n[j];
        }
        return;
    }
    triplet_finish<D>(a, b, l, u, p, n, ps, ns, rr);
}

// loss terms + the three gradient rows of one triplet from its slot rows and its (complete) dot products
template <int D>
__device__ __forceinline__ void triplet_finish(const BprArgs &a, int b, int l, const float *u, const float *p, const float *n,
                                               float ps, float ns, float rr) {
    constexpr int LPT = D < 64 ? D : 64, CPL = D / LPT;
    const float lam = a.reg_ego ? 0.f : a.lam;
    const bool tbad = triplet_bad(a, b);        // an out-of-range id voids the whole triplet
    const float x = ps - ns;
    const float gb = tbad ? 0.f : -a.inv_B * sigmoid_neg_f(x);
    if (l == 0) {
        float *lt = a.exchange ? a.contrib + (int64_t)3 * a.shard * D : a.terms + a.terms_off;
        const int stride = a.exchange ? a.shard : a.terms_stride;
        lt[b] = tbad ? 0.f : logsigmoid_f(x);
        lt[stride + b] = tbad ? 0.f : rr;
    }
    if (tbad && !a.exchange) return;
    const int64_t rows[3] = {(int64_t)a.users[b], (int64_t)a.pos[b] + a.n_users, (int64_t)a.neg[b] + a.n_users};
#pragma unroll
    for (int c = 0; c < 3; c++) {
#pragma unroll
        for (int j = 0; j < CPL; j++) {
            float g = c == 0 ? gb * (p[j] - n[j]) + lam * u[j] : (c == 1 ? gb * u[j] + lam * p[j] : (-gb) * u[j] + lam * n[j]);
            if (a.G64 && !tbad)
                atomicAdd((unsigned long long *)(a.G64 + rows[c] * D + j * LPT + l),
                          (unsigned long long)__double2ll_rn((double)g * FIXED_SCALE));
            if (a.exchange) a.contrib[((int64_t)c * a.shard + b) * D + j * LPT + l] = tbad ? 0.f : g;
        }
        if (a.G64 && !tbad && l == 0) {
            atomicOr(a.bitmap + (rows[c] >> 5), 1u << (rows[c] & 31));
            if (a.cnt) atomicAdd(a.cnt + rows[c], 1);
            if (a.item_bitmap && c > 0) { const int64_t it = rows[c] - a.n_users; atomicOr(a.item_bitmap + (it >> 5), 1u << (it & 31)); }
        }
    }
}

#ifndef TRIPLET_HUB_NNZ
#define TRIPLET_HUB_NNZ 131072  /* rows longer than this get their last-layer row from the hub plan (whole chip) instead of one workgroup
                                  (10M x 1M graph, ms per step: off 264, 8192: 242, 32768: 236, 65536: 239, 131072: 228, 262144: 231, 524288: 244) */
#endif
#ifndef TRIPLET_HUB_CHUNK
#define TRIPLET_HUB_CHUNK 2048   /* non-zeros per chunk of the hub plan's rows */
#endif
#ifndef TRIPLET_MIN_WAVES
#define TRIPLET_MIN_WAVES 4    /* bf16 tables: a 64-VGPR cap spills (7290 vs 7610 steps/s); fp32 tables fit 64 without (8 workgroups per CU, +0.4 %) */
#endif
#ifndef TRIPLET_MIN_WAVES_F32
#define TRIPLET_MIN_WAVES_F32 7 /* fp32 tables: 72 registers without scratch; the 64-register budget of round 2 (8 workgroups per CU) spills 20 bytes
                                   since the gather offsets became 32-bit + SGPR base (k_triplet 20.0 -> 21.9 us) */
#endif
#ifndef TRIPLET_U
#define TRIPLET_U LGCN_GATHER_U   /* gathers in flight per lane in k_triplet */
#endif
#ifndef TRIPLET_WAVES
#define TRIPLET_WAVES 4        /* waves per k_triplet workgroup: 4 = three gathering waves + one that reads the lower layers' rows; 3 = every wave
                                  reads the lower-layer rows of its own slot before it gathers (more workgroups resident per CU) */
#endif
#ifndef TRIPLET_WAVES_BF16
#define TRIPLET_WAVES_BF16 3   /* bf16 tables: three (measured again in run 27: Gowalla bf16 8 312 / 8 260 -> 8 398 / 8 417 steps/s; fp32 tables
                                  6 572 / 6 523 vs 6 518 / 6 560: no difference, four stays) */
#endif
// waves per k_triplet workgroup for a table type
template <typename TI> struct TripletGeo { static constexpr int NW = sizeof(TI) == 2 ? TRIPLET_WAVES_BF16 : TRIPLET_WAVES; };
#ifndef ROWS_UNIT_TILES
#define ROWS_UNIT_TILES 2      /* 64-entry tiles per unit of a slot row */
#endif
// units u_first, u_first + NW, ... of the row [start, start + n): one wave's share of a slot row.  The next
// tile's (col,val) pairs are in flight while this tile gathers, across the unit boundaries too.
template <int D, typename TG, bool BIG, int NW, typename ES>
__device__ __forceinline__ typename Geo<D, TG, false>::Acc
units_gather(const ES &es, int64_t start, int n, int u_first, const GatherSrc &src, int lane, int2 *stage) {
    typedef Geo<D, TG, false> G;
    constexpr int UT = ROWS_UNIT_TILES;
    typename G::Acc acc = zerov<G::CPL>();
    const int ntiles = (n + 63) >> 6;
    int t = u_first * UT;
    if (t >= ntiles) return acc;
    int2 cv = make_int2(0, 0);
    if (t * 64 + lane < n) cv = es.at(start + t * 64 + lane);
    for (;;) {
        if (src.S) cv.y = __float_as_int(__int_as_float(cv.y) * src.S[cv.x]);      // fp8 table: weight x row scale of the column
        const int cnt = tile_stage<false>(cv.x, __int_as_float(cv.y), min(64, n - t * 64), src, lane, stage);
        __builtin_amdgcn_wave_barrier();
        int tn = t + 1;
        if (tn % UT == 0) tn += (NW - 1) * UT;
        const bool more = tn < ntiles;
        cv = make_int2(0, 0);
        if (more && tn * 64 + lane < n) cv = es.at(start + tn * 64 + lane);
        tile_gather<D, TG, false, BIG, TRIPLET_U>(stage, cnt, src, lane, acc);
        __builtin_amdgcn_wave_barrier();
        if (!more) break;
        t = tn;
    }
    return reduce_groups<G::LPR>(acc);
}

// TG: type of the table the last layer gathers from (X_{K-1}; E0 itself when K == 1)
template <int D, typename TG, typename TI, bool BIG>
__device__ __forceinline__ void triplet_body(const BprArgs &a, const void *Xg, int2 *stage, float (*part)[TripletGeo<TI>::NW][D], float (*base)[D], float (*ego)[D]) {
    constexpr int NW = TripletGeo<TI>::NW;
    typedef Geo<D, TG, false> G;
    constexpr int C = G::CPL, LPR = G::LPR, UN = 64 * ROWS_UNIT_TILES;
    constexpr int LPT = D < 64 ? D : 64, CPT = D / LPT;      // lane = column (mod 64) layout of the finishing steps
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int b = blockIdx.x;
    const bool bad = triplet_bad(a, b);                 // an out-of-range id voids the whole triplet (err flag; zero terms)
    if (bad && threadIdx.x == 0) atomicExch(a.err, 1);
    // everything below is wave-uniform (scalar registers)
    const int64_t row0 = bad ? 0 : (int64_t)a.users[b], row1 = bad ? 0 : (int64_t)a.pos[b] + a.n_users, row2 = bad ? 0 : (int64_t)a.neg[b] + a.n_users;
    const int st0 = a.indptr[row0], st1 = a.indptr[row1], st2 = a.indptr[row2];
    const int n0 = a.indptr[row0 + 1] - st0, n1 = a.indptr[row1 + 1] - st1, n2 = a.indptr[row2 + 1] - st2;
    // the K lower layers' rows of a slot, summed in layer order (lane = column)
    auto base_row = [&](int c) {
        const int64_t row = c == 0 ? row0 : (c == 1 ? row1 : row2);
#pragma unroll
        for (int j = 0; j < CPT; j++) {
            const int col = j * LPT + lane;
            float s = a.X0[row * D + col];
            if (a.reg_ego) ego[c][col] = s;
            for (int k = 1; k < a.K; k++) s += tab_elem<TI>(a.Xl[k], a.N, D, row, col);
            base[c][col] = s;
        }
    };
    if (NW == 4 && w == 3 && lane < LPT) {              // the spare wave does all three while waves 0-2 gather
#pragma unroll
        for (int c = 0; c < 3; c++) base_row(c);
    }
    GatherSrc src; src.bm = nullptr; src.div = 1.f; src.X = Xg;
    src.S = sizeof(TG) == 1 ? fp8_scales(Xg, a.N, D) : nullptr;
    tile_pad_init(stage, threadIdx.x & 63);
    bool any = NW == 3 || (w == 0 || w == 3);
#pragma unroll 1
    for (int c = 0; c < 3; c++) {                       // (not unrolled: three inlined copies of the gather loop cost 20 VGPRs)
        const int stc = c == 0 ? st0 : (c == 1 ? st1 : st2), nc = c == 0 ? n0 : (c == 1 ? n1 : n2);
        const int u0 = NW == 4 ? ((w - c + 4) & 3) : ((w - c + 3) % 3), units = (a.hub_nnz && nc > a.hub_nnz) ? 0 : (nc + UN - 1) / UN;      // a hub row: computed by the hub plan
        if (u0 < units) {
            const typename G::Acc x = units_gather<D, TG, BIG, NW>(CsrSrc{a.indices, a.vals}, stc, nc, u0, src, lane, stage);
            if (lane < LPR) {
                if constexpr (sizeof(TG) == 1) {       // an fp8 gather: the lane's values are the chunks j*LPR + lane (see fp8_t)
#pragma unroll
                    for (int j = 0; j < 4; j++) store4(&part[c][w][(j * LPR + lane) * 4], f32x4{x[4 * j], x[4 * j + 1], x[4 * j + 2], x[4 * j + 3]});
                } else storev<C>(&part[c][w][lane * C], x);
            }
            any = true;
        }
    }
    if (NW == 3 && lane < LPT) base_row(w);             // three-wave form: wave w owns slot w's lower-layer rows (behind its gathers)
    if (!any) return;                  // s_barrier counts only the surviving waves
    __syncthreads();
    if (w != 0 || lane >= LPT) return;
    // wave 0: finish the three rows and the triplet
    const float div = (float)(a.K + 1);
    float e[3][CPT];
#pragma unroll
    for (int c = 0; c < 3; c++) {
        const int nc = c == 0 ? n0 : (c == 1 ? n1 : n2);
        const bool hub = a.hub_nnz && nc > a.hub_nnz;
        const int units = hub ? 0 : (nc + UN - 1) / UN;
        const int64_t rowc = c == 0 ? row0 : (c == 1 ? row1 : row2);
#pragma unroll
        for (int j = 0; j < CPT; j++) {
            const int col = j * LPT + lane;
            float xk = hub ? a.Xhub[rowc * D + col] : 0.f;
#pragma unroll
            for (int i = 0; i < NW; i++) if (i < units) xk += part[c][(c + i) % NW][col];     // unit 0's wave first
            e[c][j] = (base[c][col] + xk) / div;
        }
    }
    if (a.reg_ego) {
        float e0[3][CPT];
#pragma unroll
        for (int c = 0; c < 3; c++)
#pragma unroll
            for (int j = 0; j < CPT; j++) e0[c][j] = ego[c][j * LPT + lane];
        triplet_loss_regs<D>(a, b, lane, e[0], e[1], e[2], e0[0], e0[1], e0[2]);
    } else triplet_loss_regs<D>(a, b, lane, e[0], e[1], e[2]);
}

template <int D, typename TI, bool BIG>
__global__ void __launch_bounds__(64 * TripletGeo<TI>::NW, sizeof(TI) == 4 ? TRIPLET_MIN_WAVES_F32 : TRIPLET_MIN_WAVES) k_triplet(BprArgs a) {
    constexpr int NW = TripletGeo<TI>::NW;
    __shared__ int2 stage_lds[NW][TILE_ST];
    __shared__ __attribute__((aligned(32))) float part_lds[3][NW][D];
    __shared__ float base_lds[3][D];
    __shared__ float ego_lds[3][D];          // reg_ego: the slots' rows of the table itself
    // last step's row bitmap is dead: zero it here with plain stores (the two bitmaps alternate per step)
    for (int64_t i = (int64_t)blockIdx.x * (64 * NW) + threadIdx.x; i < a.bitmap_words; i += (int64_t)gridDim.x * (64 * NW))
        a.stale_bitmap[i] = 0u;
    int2 *stage = stage_lds[threadIdx.x >> 6];
    if (a.K == 1) triplet_body<D, float, TI, BIG>(a, a.X0, stage, part_lds, base_lds, ego_lds);
    else triplet_body<D, TI, TI, BIG>(a, a.Xl[a.K - 1], stage, part_lds, base_lds, ego_lds);
}

// The same when the last layer was propagated densely (cfg.dense_last): e = mean_k X_k[row] is K+1 row reads per slot,
// so one lane group (lane = column mod 64) does a whole triplet: 3 x (K+1) coalesced row reads, loss, atomics.  On graphs
// whose positives concentrate on hub items (a popularity-weighted mean item degree in the thousands: the synthetic Yelp /
// Amazon shapes) the slots together hold several times the graph's non-zeros -- one more dense SpMM is then far cheaper
// than per-slot gathers (236 -> ~50 us at B = 8192).  (Until run 56 this was two launches with the slot rows in HBM between.)
template <int D, typename TI>
__global__ void __launch_bounds__(256) k_triplet_dense(BprArgs a) {
    constexpr int LPT = D < 64 ? D : 64, CPT = D / LPT, TPB = 256 / LPT;     // lanes per triplet, triplets per workgroup
    // last step's row bitmap is dead: zero it here with plain stores (the two bitmaps alternate per step)
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < a.bitmap_words; i += (int64_t)gridDim.x * 256)
        a.stale_bitmap[i] = 0u;
    const int b = blockIdx.x * TPB + threadIdx.x / LPT, l = threadIdx.x % LPT;
    if (b >= a.B_local) return;                 // whole lane groups leave together
    const bool bad = triplet_bad(a, b);
    if (bad && l == 0) atomicExch(a.err, 1);
    const int64_t rows[3] = {bad ? 0 : (int64_t)a.users[b], bad ? 0 : (int64_t)a.pos[b] + a.n_users, bad ? 0 : (int64_t)a.neg[b] + a.n_users};
    const float div = (float)(a.K + 1);
    float e[3][CPT], e0[3][CPT];
#pragma unroll
    for (int c = 0; c < 3; c++) {
#pragma unroll
        for (int j = 0; j < CPT; j++) {
            const int64_t o = rows[c] * D + j * LPT + l;
            float s = a.X0[o];
            e0[c][j] = s;
            for (int k = 1; k <= a.K; k++) s += tab_elem<TI>(a.Xl[k], a.N, D, rows[c], j * LPT + l);
            e[c][j] = s / div;
        }
    }
    triplet_loss_regs<D>(a, b, l, e[0], e[1], e[2], e0[0], e0[1], e0[2]);
}

// Column-sharded step, after the all-reduce of the partial sums: one lane group per triplet reads its three slot rows (this rank's
// columns) and the COMPLETE scores / reg term, writes the loss terms and scatters the gradient rows of its columns.
template <int D>
__global__ void __launch_bounds__(256) k_cols_finish(BprArgs a) {
    constexpr int LPT = D < 64 ? D : 64, CPT = D / LPT, TPB = 256 / LPT;
    const int b = blockIdx.x * TPB + threadIdx.x / LPT, l = threadIdx.x % LPT;
    if (b >= a.B_local) return;
    float e[3][CPT];
#pragma unroll
    for (int c = 0; c < 3; c++)
#pragma unroll
        for (int j = 0; j < CPT; j++) e[c][j] = a.erows[((int64_t)c * a.B_local + b) * D + j * LPT + l];
    triplet_finish<D>(a, b, l, e[0], e[1], e[2], a.colsum[b], a.colsum[a.B_local + b], a.colsum[2 * a.B_local + b]);
}

// ---------------------------------------------------------------------------------
// The fork's optional branches in the fused step.
//
// Popularity gate (model.py:66-96,139-157,176-181).  For an item i with propagated row e_i and popularity scalar s_i:
//     a      = relu(W1 s_i + b1)                      pop_mlp[0]: Linear(1, Hp)
//     pv     = W2 a + b2                               pop_mlp[2]: Linear(Hp, d)
//     h      = relu(V1 [e_i ; pv] + c1)                gate_mlp[0]: Linear(2d, Hg)
//     g      = sigmoid((V2 h + c2) / T)                gate_mlp[2]: Linear(Hg, 1)
//     f_i    = g e_i + (1 - g) pv                      the row bpr_loss scores with
//     loss   = bpr(u, f_p, f_n) - coeff * mean over the 2B gates of H(clamp(g, 1e-6, 1 - 1e-6))
// bpr_loss reads f only on the 2B item slots of the batch, so the gate and its backward run on those slots: one wave per
// triplet (lane = column for the row work, lane = hidden unit for the two MLPs), the MLP weights in LDS (V1 and W2 row
// padded by one float: conflict-free both for "lane = unit walks a row" and "lane = column walks a column"), the
// gradient rows w.r.t. e_u, e_p, e_n scattered with the same fixed-point atomics as k_triplet, and the parameter
// gradients summed in two deterministic stages: per workgroup over its slots (phase 2 below, fixed slot order),
// then over the workgroups in index order inside k_gate_adam, which also applies torch.optim.Adam to the MLP parameters.
// ---------------------------------------------------------------------------------
#define GATE_TPB 8            /* triplets per workgroup (two per wave) */
#define GATE_S (2 * GATE_TPB) /* item slots per workgroup */
#define GATE_HMAX 64          /* pop_hidden, gate_hidden <= 64 (lane = hidden unit) */
struct GateArgs {
    const float *E;               // [N,d] final propagated table (layer mean, item-item smoothing applied)
    const float *item_pop;        // [m_items]
    const float *params;          // flat: W1[Hp] b1[Hp] W2[d*Hp] b2[d] V1[Hg*2d] c1[Hg] V2[Hg] c2[1]  (torch.nn.Linear layouts)
    long long *partials;          // [grid, P] per-workgroup parameter-gradient sums, FIXED POINT (2^50): every slot's contribution is
                                  // converted before it is added, so the total is independent of how slots fall into workgroups and
                                  // ranks (integer sums are associative) -- the data-parallel step equals the single-GPU step bit for bit
    int32_t Hp, Hg, P;
    float inv_temp, ent_scale;    // 1 / pop_gate_temp ; gate_entropy_coeff / (2 B)
    int32_t n_users; int64_t N;
    const int32_t *users; const int32_t *pos; const int32_t *neg;
    int32_t B; float inv_B, lam;
    long long *G64; uint32_t *bitmap; uint32_t *item_bitmap;      // item_bitmap: bit i = item i named by the batch (or NULL)
    uint32_t *stale_bitmap; int64_t bitmap_words;
    float *terms; int32_t terms_stride;
    int32_t *err;
    float *contrib; int32_t shard; int32_t exchange;   // data parallel: gradient rows and loss terms of this rank's shard go to the exchange
                                                        // block [3*shard*D | shard | shard | shard] (besides / instead of the atomics)
};
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

template <int D>
__global__ void __launch_bounds__(256) k_triplet_gate(GateArgs a) {
    constexpr int LPT = D < 64 ? D : 64, CPT = D / LPT, D2 = 2 * D, V1S = D2 + 1;
    extern __shared__ __attribute__((aligned(16))) float gsm[];
    const int Hp = a.Hp, Hg = a.Hg, W2S = Hp + 1;
    // LDS carve-up
    float *V1p = gsm;                                  // [Hg][2d + 1]
    float *W2p = V1p + GATE_HMAX * V1S;                // [d][Hp + 1]
    float *W1 = W2p + D * (GATE_HMAX + 1);             // [Hp]
    float *b1 = W1 + GATE_HMAX, *b2 = b1 + GATE_HMAX;  // [Hp], [d]
    float *c1 = b2 + D, *V2 = c1 + GATE_HMAX;          // [Hg], [Hg]
    float *IN = V2 + GATE_HMAX;                        // [S][2d]  gate input [e ; pv]
    float *DH = IN + GATE_S * D2;                      // [S][Hmax] d loss / d (pre-activation of the gate's hidden layer)
    float *HR = DH + GATE_S * GATE_HMAX;               // [S][Hmax] relu(h)
    float *DPV = HR + GATE_S * GATE_HMAX;              // [S][d]   d loss / d pv
    float *AA = DPV + GATE_S * D;                      // [S][Hmax] relu(a)
    float *DA = AA + GATE_S * GATE_HMAX;               // [S][Hmax] d loss / d (pre-activation of the pop hidden layer)
    float *SC = DA + GATE_S * GATE_HMAX;               // [S] popularity scalar
    float *DLOG = SC + GATE_S;                         // [S] d loss / d logit
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    for (int64_t i = (int64_t)blockIdx.x * 256 + tid; i < a.bitmap_words; i += (int64_t)gridDim.x * 256) a.stale_bitmap[i] = 0u;
    const int oW2 = 2 * Hp, ob2 = oW2 + D * Hp, oV1 = ob2 + D, oc1 = oV1 + Hg * D2, oV2 = oc1 + Hg, oc2 = oV2 + Hg;
    // (the weight copies are unrolled: with one load per round trip the 8 192 + 2 048 words took a third of the launch)
#pragma unroll 16
    for (int i = tid; i < Hg * D2; i += 256) V1p[(i / D2) * V1S + i % D2] = a.params[oV1 + i];
#pragma unroll 8
    for (int i = tid; i < D * Hp; i += 256) W2p[(i / Hp) * W2S + i % Hp] = a.params[oW2 + i];
    if (tid < Hp) { W1[tid] = a.params[tid]; b1[tid] = a.params[Hp + tid]; }
    for (int i = tid; i < D; i += 256) b2[i] = a.params[ob2 + i];
    if (tid < Hg) { c1[tid] = a.params[oc1 + tid]; V2[tid] = a.params[oV2 + tid]; }
    const float c2 = a.params[oc2];
    // records of slots that end up unused (batch tail, bad ids) must read as zero in phase 2
    for (int i = tid; i < GATE_S * GATE_HMAX; i += 256) { DH[i] = 0.f; HR[i] = 0.f; AA[i] = 0.f; DA[i] = 0.f; }
    for (int i = tid; i < GATE_S * D; i += 256) DPV[i] = 0.f;
    for (int i = tid; i < GATE_S * D2; i += 256) IN[i] = 0.f;
    if (tid < GATE_S) { SC[tid] = 0.f; DLOG[tid] = 0.f; }
    __syncthreads();
    // ---- phase 1: one wave per triplet
    for (int tt = w; tt < GATE_TPB; tt += 4) {
        const int b = blockIdx.x * GATE_TPB + tt;
        if (b >= a.B) break;
        const int iu = a.users[b], ip = a.pos[b], in_ = a.neg[b];
        const bool bad = iu < 0 || iu >= a.n_users || ip < 0 || (int64_t)ip + a.n_users >= a.N || in_ < 0 || (int64_t)in_ + a.n_users >= a.N;
        float *lt = a.exchange ? a.contrib + (int64_t)3 * a.shard * D : a.terms;
        const int lts = a.exchange ? a.shard : a.terms_stride;
        if (bad) {
            if (lane == 0) { atomicExch(a.err, 1); lt[b] = 0.f; lt[lts + b] = 0.f; lt[2 * lts + b] = 0.f; }
            if (a.exchange && lane < LPT)
                for (int c3 = 0; c3 < 3; c3++)
                    for (int j = 0; j < CPT; j++) a.contrib[((int64_t)c3 * a.shard + b) * D + j * LPT + lane] = 0.f;
            continue;
        }
        const bool col = lane < LPT;
        float u[CPT], e[2][CPT], pv[2][CPT], f[2][CPT], g[2], ent[2];
        const int item[2] = {ip, in_};
#pragma unroll
        for (int j = 0; j < CPT; j++) u[j] = col ? a.E[(int64_t)iu * D + j * LPT + lane] : 0.f;
#pragma unroll
        for (int q = 0; q < 2; q++) {
            const int sidx = 2 * tt + q;
            const int64_t row = (int64_t)item[q] + a.n_users;
#pragma unroll
            for (int j = 0; j < CPT; j++) e[q][j] = col ? a.E[row * D + j * LPT + lane] : 0.f;
            const float sc = a.item_pop[item[q]];
            if (lane < Hp) AA[sidx * GATE_HMAX + lane] = fmaxf(W1[lane] * sc + b1[lane], 0.f);
            if (lane == 0) SC[sidx] = sc;
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int j = 0; j < CPT; j++) {
                const int c = j * LPT + lane;
                float acc = 0.f;
                if (col) {
                    acc = b2[c];
#pragma unroll 8
                    for (int k = 0; k < Hp; k++) acc += W2p[c * W2S + k] * AA[sidx * GATE_HMAX + k];
                }
                pv[q][j] = acc;
                if (col) { IN[sidx * D2 + c] = e[q][j]; IN[sidx * D2 + D + c] = acc; }
            }
            __builtin_amdgcn_wave_barrier();
            float hr = 0.f;
            if (lane < Hg) {
                float h = c1[lane];
#pragma unroll 16
                for (int c = 0; c < D2; c++) h += V1p[lane * V1S + c] * IN[sidx * D2 + c];       // (16 LDS reads in flight, one accumulator chain)
                hr = fmaxf(h, 0.f);
                HR[sidx * GATE_HMAX + lane] = hr;
            }
            const float logit = (wave_sum(lane < Hg ? V2[lane] * hr : 0.f) + c2) * a.inv_temp;
            g[q] = 1.f / (1.f + expf(-logit));                                 // torch.sigmoid
            const float gc = fminf(fmaxf(g[q], 1e-6f), 1.f - 1e-6f);           // torch.clamp(gates, 1e-6, 1 - 1e-6)
            ent[q] = -(gc * logf(gc) + (1.f - gc) * logf(1.f - gc));
#pragma unroll
            for (int j = 0; j < CPT; j++) f[q][j] = g[q] * e[q][j] + (1.f - g[q]) * pv[q][j];
        }
        // loss terms (model.py:168-173 on the fused rows)
        float ps = 0.f, ns = 0.f, rr = 0.f;
#pragma unroll
        for (int j = 0; j < CPT; j++) {
            ps += u[j] * f[0][j]; ns += u[j] * f[1][j];
            rr += u[j] * u[j] + f[0][j] * f[0][j] + f[1][j] * f[1][j];
        }
        ps = wave_sum(ps); ns = wave_sum(ns); rr = wave_sum(rr);
        const float x = ps - ns;
        const float gb = -a.inv_B * sigmoid_neg_f(x);
        if (lane == 0) { lt[b] = logsigmoid_f(x); lt[lts + b] = rr; lt[2 * lts + b] = ent[0] + ent[1]; }
        // backward: user row
        const int64_t urow = iu;
#pragma unroll
        for (int j = 0; j < CPT; j++) {
            if (col) {
                const float du = gb * (f[0][j] - f[1][j]) + a.lam * u[j];
                if (a.G64) atomicAdd((unsigned long long *)(a.G64 + urow * D + j * LPT + lane), (unsigned long long)__double2ll_rn((double)du * FIXED_SCALE));
                if (a.exchange) a.contrib[((int64_t)0 * a.shard + b) * D + j * LPT + lane] = du;
            }
        }
        if (a.G64 && lane == 0) atomicOr(a.bitmap + (urow >> 5), 1u << (urow & 31));
#pragma unroll
        for (int q = 0; q < 2; q++) {
            const int sidx = 2 * tt + q;
            const int64_t row = (int64_t)item[q] + a.n_users;
            float dF[CPT], part = 0.f;
#pragma unroll
            for (int j = 0; j < CPT; j++) {
                dF[j] = (q == 0 ? gb : -gb) * u[j] + a.lam * f[q][j];
                part += dF[j] * (e[q][j] - pv[q][j]);
            }
            const bool inside = g[q] > 1e-6f && g[q] < 1.f - 1e-6f;            // the clamp passes gradient only inside its range
            const float dent = inside ? a.ent_scale * (logf(g[q]) - logf(1.f - g[q])) : 0.f;
            const float dlogit = (wave_sum(part) + dent) * g[q] * (1.f - g[q]) * a.inv_temp;
            if (lane < Hg) DH[sidx * GATE_HMAX + lane] = HR[sidx * GATE_HMAX + lane] > 0.f ? dlogit * V2[lane] : 0.f;
            if (lane == 0) DLOG[sidx] = dlogit;
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int j = 0; j < CPT; j++) {
                const int c = j * LPT + lane;
                float die = 0.f, dip = 0.f;
                if (col) {
#pragma unroll 8
                    for (int k = 0; k < Hg; k++) { const float dh = DH[sidx * GATE_HMAX + k]; die += V1p[k * V1S + c] * dh; dip += V1p[k * V1S + D + c] * dh; }
                }
                const float de = g[q] * dF[j] + die, dpv = (1.f - g[q]) * dF[j] + dip;
                if (col) {
                    DPV[sidx * D + c] = dpv;
                    if (a.G64) atomicAdd((unsigned long long *)(a.G64 + row * D + c), (unsigned long long)__double2ll_rn((double)de * FIXED_SCALE));
                    if (a.exchange) a.contrib[((int64_t)(1 + q) * a.shard + b) * D + c] = de;
                }
            }
            if (a.G64 && lane == 0) {
                atomicOr(a.bitmap + (row >> 5), 1u << (row & 31));
                if (a.item_bitmap) atomicOr(a.item_bitmap + (item[q] >> 5), 1u << (item[q] & 31));
            }
            __builtin_amdgcn_wave_barrier();
            if (lane < Hp) {
                float da = 0.f;
#pragma unroll 16
                for (int c = 0; c < D; c++) da += W2p[c * W2S + lane] * DPV[sidx * D + c];
                DA[sidx * GATE_HMAX + lane] = AA[sidx * GATE_HMAX + lane] > 0.f ? da : 0.f;
            }
        }
    }
    __syncthreads();
    // ---- phase 2: this workgroup's parameter-gradient sums over its slots.  The two item slots of a triplet (2 tt, 2 tt + 1) are
    //      added in fp32 -- the same pair on every rank whatever the shard -- and each triplet's term enters the sum in fixed point:
    //      the total does not depend on which triplets share a workgroup or a rank (half the conversions of one per slot)
    long long *out = a.partials + (int64_t)blockIdx.x * a.P;
    auto fx = [](float v) { return __double2ll_rn((double)v * FIXED_SCALE); };
#define GATE_SUM(EXPR) _Pragma("unroll") for (int w2 = 0; w2 < GATE_S; w2 += 2) { float pr_; { const int sI = w2; pr_ = (EXPR); } { const int sI = w2 + 1; pr_ += (EXPR); } v += fx(pr_); }
    for (int i = tid; i < a.P; i += 256) {
        long long v = 0;
        if (i < Hp) { GATE_SUM(DA[sI * GATE_HMAX + i] * SC[sI]) }
        else if (i < oW2) { const int k = i - Hp; GATE_SUM(DA[sI * GATE_HMAX + k]) }
        else if (i < ob2) { const int c = (i - oW2) / Hp, k = (i - oW2) % Hp; GATE_SUM(DPV[sI * D + c] * AA[sI * GATE_HMAX + k]) }
        else if (i < oV1) { const int c = i - ob2; GATE_SUM(DPV[sI * D + c]) }
        else if (i < oc1) { const int jj = (i - oV1) / D2, c = (i - oV1) % D2; GATE_SUM(DH[sI * GATE_HMAX + jj] * IN[sI * D2 + c]) }
        else if (i < oV2) { const int jj = i - oc1; GATE_SUM(DH[sI * GATE_HMAX + jj]) }
        else if (i < oc2) { const int jj = i - oV2; GATE_SUM(DLOG[sI] * HR[sI * GATE_HMAX + jj]) }
        else { GATE_SUM(DLOG[sI]) }
        out[i] = v;
    }
#undef GATE_SUM
}
static size_t gate_lds_bytes(int D) {
    const size_t fl = (size_t)GATE_HMAX * (2 * D + 1) + (size_t)D * (GATE_HMAX + 1) + 2 * GATE_HMAX + D + 2 * GATE_HMAX
                    + (size_t)GATE_S * 2 * D + 2 * GATE_S * GATE_HMAX + (size_t)GATE_S * D + 2 * GATE_S * GATE_HMAX + 2 * GATE_S;
    return fl * sizeof(float);
}

// sum of the fixed-point partial sums (of the workgroups on one GPU; of the ranks' totals in the exchange blocks under data
// parallelism -- integer sums: any grouping gives the same bits) + torch.optim.Adam on the MLP parameters (same arithmetic as
// spmm_epilogue's); grad_out keeps the reduced gradient (tests, inspection)
struct GateAdamArgs {
    const long long *src; int32_t n_src; int64_t stride; int32_t P;      // n_src vectors of P sums, `stride` int64 apart
    float *params, *m, *v, *grad_out;
    float step_size, bc2_sqrt, w1, beta2, omb2, eps;
};
__global__ void __launch_bounds__(256) k_gate_adam(GateAdamArgs a) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= a.P) return;
    long long q = 0;
    for (int q0 = 0; q0 < a.n_src; q0 += 32) {               // 32 loads in flight per round trip
        long long t[32];
#pragma unroll
        for (int u = 0; u < 32; u++) t[u] = q0 + u < a.n_src ? a.src[(int64_t)(q0 + u) * a.stride + i] : 0;
#pragma unroll
        for (int u = 0; u < 32; u++) q += t[u];
    }
    const float g = (float)((double)q * FIXED_INV);
    a.grad_out[i] = g;
    float m = a.m[i], v = a.v[i], p = a.params[i];
    m = m + a.w1 * (g - m);
    v = v * a.beta2 + (a.omb2 * g) * g;
    const float denom = sqrtf(v) / a.bc2_sqrt + a.eps;
    p = p - a.step_size * (m / denom);
    a.params[i] = p; a.m[i] = m; a.v[i] = v;
}
// data parallel: this rank's total (sum of its workgroups' partial sums) into the tail of its exchange block
__global__ void __launch_bounds__(256) k_gate_rank_total(const long long *partials, int32_t n_part, int32_t P, long long *out) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= P) return;
    long long q = 0;
    for (int w = 0; w < n_part; w++) q += partials[(int64_t)w * P + i];
    out[i] = q;
}

// T = mean of the K+1 layers (the stack + mean of computer(), model.py:221-222): user rows to out_u, item rows to out_i
// (two destinations: with item-item smoothing the item block is an SpMM input and its result lands beside the users)
struct MeanSplitArgs {
    const float *X0; const void *Xl[LGCN_MAX_LAYERS + 1]; int K;
    float *out_u, *out_i; int64_t n4_users, n4;      // 4-element pieces: of the user block, of the whole table
};
template <typename TI>
__global__ void __launch_bounds__(256) k_mean_layers(MeanSplitArgs a) {
    const float div = (float)(a.K + 1);
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < a.n4; i += (int64_t)gridDim.x * 256) {
        f32x4 s = load4(a.X0 + i * 4);
        for (int k = 1; k <= a.K; k++) s += load4((const TI *)a.Xl[k] + i * 4);
        store4((i < a.n4_users ? a.out_u : a.out_i) + i * 4, s / div);
    }
}
// flag the rows [lo, hi) in a row bitmap (item-item smoothing makes every item row of Gs non-zero)
__global__ void __launch_bounds__(256) k_flag_range(uint32_t *bm, int64_t lo, int64_t hi) {
    const int64_t wlo = lo >> 5, whi = (hi + 31) >> 5;
    for (int64_t wd = wlo + (int64_t)blockIdx.x * 256 + threadIdx.x; wd < whi; wd += (int64_t)gridDim.x * 256) {
        uint32_t m = 0xffffffffu;
        if (wd == (lo >> 5)) m &= 0xffffffffu << (lo & 31);
        if (wd == (hi >> 5)) m &= (hi & 31) ? (0xffffffffu >> (32 - (hi & 31))) : 0u;
        if (m == 0xffffffffu) bm[wd] = m; else if (m) atomicOr(bm + wd, m);
    }
}

// slot -> destination row of the global batch (-1: the slot's triplet has a bad id)
__device__ __forceinline__ int64_t slot_row(int c, int b, const int32_t *users, const int32_t *pos,
                                            const int32_t *neg, int32_t n_users, int64_t N) {
    const int u = users[b], p = pos[b], n = neg[b];
    if (u < 0 || u >= n_users || p < 0 || (int64_t)p + n_users >= N || n < 0 || (int64_t)n + n_users >= N) return -1;
    return c == 0 ? (int64_t)u : (int64_t)(c == 1 ? p : n) + n_users;
}

struct SlotArgs {
    const int32_t *users; const int32_t *pos; const int32_t *neg;
    int32_t B; int32_t n_users; int64_t N;
    long long *G64; uint32_t *bitmap;
    float *G32; float div;                                 // k_g32: fp32 copy of the flagged rows, K+1
    const float *gathered; int32_t shard; int32_t world;   // DP scatter
    int32_t skip_rank;                                     // DP scatter: this rank's own block is in G64 already (-1: none)
    const float *terms; float *loss_out; float decay;
    float ent_coeff;
    int64_t blk;                                           // floats per rank in `gathered` (0: default layout)
    uint32_t *item_bitmap;                                 // DP scatter with item-item smoothing: items named by the global batch
    int32_t *cnt;                                          // reg_ego: per-row slot counts (k_scatter / k_flag_rows bump, k_finish clears), or NULL
};

// DP: order-independent scatter of every rank's gradient rows into G64 (+ row flags)
template <int D>
__global__ void __launch_bounds__(256) k_scatter(SlotArgs a) {
    // lane = column (mod 64), like triplet_loss: every 64-bit atomic wave-instruction covers 512 contiguous bytes.
    // (4 columns per lane -- 16 lanes x 8 B at a 32-byte stride per instruction -- ran at 0.24 TB/s of added bytes:
    //  12.9 us per 2048 triplets, 84 us for an 8-rank batch.)
    constexpr int LPT = D < 64 ? D : 64, CPT = D / LPT, SPB = 256 / LPT;
    const int s = blockIdx.x * SPB + threadIdx.x / LPT, l = threadIdx.x % LPT;
    if (s >= 3 * a.B) return;
    const int c = s / a.B, b = s % a.B;
    const int64_t row = slot_row(c, b, a.users, a.pos, a.neg, a.n_users, a.N);
    if (row < 0) return;
    const int r = b / a.shard, i = b % a.shard;
    if (r == a.skip_rank) return;
    const int64_t blk = a.blk ? a.blk : (int64_t)3 * a.shard * D + 2 * a.shard;
    const float *src = a.gathered + r * blk + ((int64_t)c * a.shard + i) * D;
#pragma unroll
    for (int j = 0; j < CPT; j++)
        atomicAdd((unsigned long long *)(a.G64 + row * D + j * LPT + l),
                  (unsigned long long)__double2ll_rn((double)src[j * LPT + l] * FIXED_SCALE));
    if (l == 0) {
        atomicOr(a.bitmap + (row >> 5), 1u << (row & 31));
        if (a.cnt) atomicAdd(a.cnt + row, 1);
        if (a.item_bitmap && c > 0) { const int64_t it = row - a.n_users; atomicOr(a.item_bitmap + (it >> 5), 1u << (it & 31)); }
    }
}

// Gs rows of the batch, once per step: G32[row] = (float)(G64[row] * 2^-50) / (K+1) for every slot's row, after
// ALL contributions are in (k_triplet / k_scatter / the all-reduce).  Slots that share a row write the same
// bytes.  The first backward layer gathers these 4-byte rows (and every Horner epilogue adds them) instead of
// converting the fixed-point rows per gathered copy.
template <int D>
__global__ void __launch_bounds__(256) k_g32(SlotArgs a) {
    constexpr int LPR = D / 4, SPB = 256 / LPR;
    const int s = blockIdx.x * SPB + threadIdx.x / LPR, l = threadIdx.x % LPR;
    if (s >= 3 * a.B) return;
    const int64_t row = slot_row(s / a.B, s % a.B, a.users, a.pos, a.neg, a.n_users, a.N);
    if (row < 0) return;
    store4(a.G32 + row * D + l * 4, loadv_fixed<4>(a.G64 + row * D + l * 4, a.div));
}

// dense data-parallel form: flag the rows of the WHOLE global batch (every rank knows all ids), so the
// row bitmap needs no collective (RCCL has no bitwise-OR reduction)
__global__ void __launch_bounds__(256) k_flag_rows(SlotArgs a) {
    const int s = blockIdx.x * 256 + threadIdx.x;
    if (s >= 3 * a.B) return;
    const int64_t row = slot_row(s / a.B, s % a.B, a.users, a.pos, a.neg, a.n_users, a.N);
    if (row >= 0) {
        atomicOr(a.bitmap + (row >> 5), 1u << (row & 31));
        if (a.cnt) atomicAdd(a.cnt + row, 1);          // dense form: every rank counts the whole global batch itself (part 1 does not)
        if (a.item_bitmap && s >= a.B) { const int64_t it = row - a.n_users; atomicOr(a.item_bitmap + (it >> 5), 1u << (it & 31)); }
    }
}

// End of step: zero what the step touched (G64 rows, bitmap words of the batch rows);
// block 0 also reduces the per-triplet loss terms in a fixed order.
template <int D>
__global__ void __launch_bounds__(256) k_finish(SlotArgs a) {
    constexpr int LPR = D / 4, SPB = 256 / LPR;
    const int s = blockIdx.x * SPB + threadIdx.x / LPR, l = threadIdx.x % LPR;
    if (s < 3 * a.B) {
        const int c = s / a.B, b = s % a.B;
        const int64_t row = slot_row(c, b, a.users, a.pos, a.neg, a.n_users, a.N);
        if (row >= 0) {
                    i64x2 *q = reinterpret_cast<i64x2 *>(a.G64 + row * D + l * 4);
            q[0] = i64x2{0, 0}; q[1] = i64x2{0, 0};
            if (l == 0) { a.bitmap[row >> 5] = 0u; if (a.cnt) a.cnt[row] = 0; }
        }
    }
    // the same single-wave, fixed-order reduction as the fused finish of the last SpMM: identical bits
    if (blockIdx.x == 0 && threadIdx.x < 64)
        reduce_loss_wave(a.terms, a.gathered, a.B, a.shard, D, a.decay, a.loss_out, (int)threadIdx.x, a.ent_coeff, a.blk);
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
// gathered tables of 4 GiB or more need 64-bit offsets (priced at the fp32 row size whatever the table type)
static inline bool big_table(int64_t n_rows, int d) { return n_rows * (int64_t)d * 4 >= (1ll << 32); }

template <int D, typename TI, typename TO, int MODE>
static void launch_spmm_t(const SpmmArgs &a, hipStream_t st) {
    constexpr int RPB = PackGeo<RowGeo<D, TI, (MODE & M_SPARSE) != 0>::RPK>::RPB;
    static_assert(SLICE_PAD % RPB == 0, "slice padding must hold whole workgroups of every variant");
    unsigned grid = 0, widest = 0;
    for (int x = 0; x < XCDS; x++) {
        const unsigned nb = (unsigned)((a.sp.cblk[x + 1] - a.sp.cblk[x]) * (4 / SPMM_WPB) + (a.sp.rows[x + 1] - a.sp.rows[x]) / RPB);
        grid += nb; widest = nb > widest ? nb : widest;
    }
    if (a.remap) grid = widest * XCDS;
    if (grid == 0) return;
    if (big_table(a.n_rows, D)) hipLaunchKernelGGL((k_spmm<D, TI, TO, MODE, true>), dim3(grid), dim3(64 * SPMM_WPB), 0, st, a);
    else hipLaunchKernelGGL((k_spmm<D, TI, TO, MODE, false>), dim3(grid), dim3(64 * SPMM_WPB), 0, st, a);
}

template <int D, int MODE>
static int launch_spmm_d(const SpmmArgs &a, int x_dtype, int y_dtype, hipStream_t st) {
    if (MODE & M_SPARSE) x_dtype = LGCN_F32;           // source is the fixed-point table; TI unused
    if (MODE & M_ADAM) y_dtype = LGCN_F32;
    if (x_dtype == LGCN_FP8 || y_dtype == LGCN_FP8) {
        // fp8 tables: 16 elements per lane, so d >= 64; fp8 <-> bf16 conversions are not instantiated
        if constexpr (D >= 64) {
            if (x_dtype == LGCN_FP8 && y_dtype == LGCN_FP8) { launch_spmm_t<D, fp8_t, fp8_t, MODE>(a, st); return 0; }
            if (x_dtype == LGCN_FP8 && y_dtype == LGCN_F32) { launch_spmm_t<D, fp8_t, float, MODE>(a, st); return 0; }
            if (x_dtype == LGCN_F32 && y_dtype == LGCN_FP8) { launch_spmm_t<D, float, fp8_t, MODE>(a, st); return 0; }
        }
        lgcn_set_error("fp8 tables: embedding dim 64, 128 or 256, and fp8 <-> fp32 only (no fp8 <-> bf16 launch)");
        return 3;
    }
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

// Y = selfX + A X on fp32 tables (item-item smoothing): only the fp32 instance exists
static int launch_spmm_addself(const SpmmArgs &a, int d, hipStream_t st) {
    switch (d) {
    case 32: launch_spmm_t<32, float, float, M_ADDSELF>(a, st); return 0;
    case 64: launch_spmm_t<64, float, float, M_ADDSELF>(a, st); return 0;
    case 128: launch_spmm_t<128, float, float, M_ADDSELF>(a, st); return 0;
    case 256: launch_spmm_t<256, float, float, M_ADDSELF>(a, st); return 0;
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
    if (t != LGCN_F32 && t != LGCN_BF16 && t != LGCN_FP8) { lgcn_set_error("dtype must be LGCN_F32, LGCN_BF16 or LGCN_FP8"); return 3; }
    return 0;
}

extern "C" int lgcn_device_available(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) { (void)hipGetLastError(); return 0; }
    return n > 0;
}

// ---------------------------------------------------------------------------------
// graph object: device CSR (borrowed) + the long-row plan and its scratch (owned)
// ---------------------------------------------------------------------------------
struct lgcn_graph {
    const int32_t *indptr; const int32_t *indices; const float *vals; const int4 *rowinfo;
    const int2 *pk;       // the planned rows' (col,val) pairs, packed, in plan order
    int64_t n_rows, nnz;
    int32_t d_max;
    LongPlan lp; SlicePlan sp;
    void *owned;          // one device allocation holding plan arrays, partials and counters
    // The long-row scratch (partials, tickets) is shared by every launch on this graph: launches
    // are ordered.  A launch on another stream than the previous one first waits for it.
    mutable hipStream_t last_stream; mutable bool used; mutable hipEvent_t order_ev;
};

// a launch on `st` is about to use the graph's scratch: order it behind the previous user
static int graph_acquire(const lgcn_graph *g, hipStream_t st) {
    if (g->used && g->last_stream != st) {
        HIP_OK(hipEventRecord(g->order_ev, g->last_stream));
        HIP_OK(hipStreamWaitEvent(st, g->order_ev, 0));
    }
    g->used = true; g->last_stream = st;
    return 0;
}

// column indices must address rows of X: flag any index outside [0, n_rows)
__global__ void __launch_bounds__(256) k_index_range(const int32_t *indices, int64_t nnz, int32_t n_rows, int32_t *bad) {
    bool b = false;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < nnz; i += (int64_t)gridDim.x * 256) {
        const int32_t c = indices[i];
        b |= (c < 0 || c >= n_rows);
    }
    if (__ballot(b) && (threadIdx.x & 63) == 0) atomicOr(bad, 1);
}

// copy the planned rows' CSR segments into the packed stream: items = (first CSR entry, first stream entry, count, 0)
__global__ void __launch_bounds__(256) k_pack_stream(const int4 *items, int64_t n_items, const int32_t *indices,
                                                     const float *vals, int2 *pk) {
    const int lane = threadIdx.x & 63;
    const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = (int64_t)gridDim.x * 4;
    for (int64_t i = wave; i < n_items; i += nwaves) {
        const int4 it = items[i];
        for (int e = lane; e < it.z; e += 64)
            pk[(int64_t)it.y + e] = make_int2(indices[(int64_t)it.x + e], __float_as_int(vals[(int64_t)it.x + e]));
    }
}

static int graph_create_impl(const int32_t *indptr, const int32_t *indices, const float *vals,
                             int64_t n_rows, int64_t nnz, int32_t d_max, const int32_t *row_order,
                             int64_t n_order, const int64_t *xcd_start, int32_t long_ch, lgcn_graph **out);
extern "C" int lgcn_graph_create(const int32_t *indptr, const int32_t *indices, const float *vals,
                                 int64_t n_rows, int64_t nnz, int32_t d_max, const int32_t *row_order,
                                 int64_t n_order, const int64_t *xcd_start, lgcn_graph **out) {
    return graph_create_impl(indptr, indices, vals, n_rows, nnz, d_max, row_order, n_order, xcd_start, LONG_CH, out);
}
// long_ch: non-zeros per chunk of a split row (LONG_CH for the propagation plans; the hub plan of k_triplet, whose rows
// have 10^4 .. 10^6 non-zeros, takes longer chunks so that a row's last arriver has hundreds, not thousands, of partials to add)
static int graph_create_impl(const int32_t *indptr, const int32_t *indices, const float *vals,
                             int64_t n_rows, int64_t nnz, int32_t d_max, const int32_t *row_order,
                             int64_t n_order, const int64_t *xcd_start, int32_t long_ch, lgcn_graph **out) {
    if (!indptr || !indices || !vals || !out || n_rows <= 0 || nnz < 0 || nnz > 0x7fffffffLL ||
        n_rows >= 0x7fffffffLL) { lgcn_set_error("lgcn_graph_create: invalid argument"); return 3; }
    if (d_max != 32 && d_max != 64 && d_max != 128 && d_max != 256) { lgcn_set_error("lgcn_graph_create: d_max must be 32, 64, 128 or 256"); return 3; }
    std::vector<int32_t> ip((size_t)n_rows + 1);
    HIP_OK(hipMemcpy(ip.data(), indptr, sizeof(int32_t) * ip.size(), hipMemcpyDeviceToHost));
    if (ip[0] != 0 || (int64_t)ip[(size_t)n_rows] != nnz) { lgcn_set_error("lgcn_graph_create: indptr does not match nnz"); return 3; }
    if (nnz > 0) {               // a bad column index would become an out-of-bounds row gather
        int32_t *bad = nullptr, hbad = 0;
        HIP_OK(hipMalloc((void **)&bad, sizeof(int32_t)));
        hipError_t e0 = hipMemset(bad, 0, sizeof(int32_t));
        if (e0 == hipSuccess) {
            const int64_t blocks = (nnz + 255) / 256;
            hipLaunchKernelGGL(k_index_range, dim3((unsigned)(blocks < 4096 ? blocks : 4096)), dim3(256), 0, 0,
                               indices, nnz, (int32_t)n_rows, bad);
            e0 = hipMemcpy(&hbad, bad, sizeof(int32_t), hipMemcpyDeviceToHost);
        }
        (void)hipFree(bad);
        if (e0 != hipSuccess) { lgcn_set_error("lgcn_graph_create: index check failed to run"); return 10; }
        if (hbad) { lgcn_set_error("lgcn_graph_create: column index out of range [0, n_rows)"); return 3; }
    }
    std::vector<int32_t> ord;
    if (!row_order) n_order = n_rows;
    if (n_order < 0 || n_order > n_rows) { lgcn_set_error("lgcn_graph_create: n_order out of range"); return 3; }
    if (row_order) {             // a permutation of 0..n_rows-1, or a subset of the rows without repetition
        ord.resize((size_t)n_order);
        if (n_order) HIP_OK(hipMemcpy(ord.data(), row_order, sizeof(int32_t) * ord.size(), hipMemcpyDeviceToHost));
        std::vector<char> seen((size_t)n_rows, 0);
        for (int64_t i = 0; i < n_order; i++) {
            const int32_t r = ord[(size_t)i];
            if (r < 0 || r >= n_rows || seen[(size_t)r]) { lgcn_set_error("lgcn_graph_create: row_order is not a permutation / repeats a row"); return 3; }
            seen[(size_t)r] = 1;
        }
    }
    for (int64_t r = 0; r < n_rows; r++)
        if (ip[(size_t)r + 1] < ip[(size_t)r]) { lgcn_set_error("lgcn_graph_create: indptr not monotone"); return 3; }
    // ---- slices of the processing order, one per XCD
    int64_t xs[XCDS + 1];
    if (xcd_start) {
        for (int x = 0; x <= XCDS; x++) xs[x] = xcd_start[x];
        bool ok = xs[0] == 0 && xs[XCDS] == n_order;
        for (int x = 0; x < XCDS; x++) ok = ok && xs[x] <= xs[x + 1];
        if (!ok) { lgcn_set_error("lgcn_graph_create: xcd_start must be 9 non-decreasing positions from 0 to n_order"); return 3; }
    } else {                     // balance the work: non-zeros plus a per-row constant (see reorder.py _row_cost)
        const double ROW_COST = 4.0;           // small per-row term (measured: 0..16 equal on Gowalla, larger is slower)
        double total = ROW_COST * (double)n_order;
        for (int64_t p = 0; p < n_order; p++) {
            const int32_t r = row_order ? ord[(size_t)p] : (int32_t)p;
            total += (double)(ip[(size_t)r + 1] - ip[(size_t)r]);
        }
        double acc = 0.0; int x = 1;
        xs[0] = 0;
        for (int64_t p = 0; p < n_order && x < XCDS; p++) {
            const int32_t r = row_order ? ord[(size_t)p] : (int32_t)p;
            acc += (double)(ip[(size_t)r + 1] - ip[(size_t)r]) + ROW_COST;
            while (x < XCDS && acc >= total * x / XCDS) xs[x++] = p + 1;
        }
        while (x <= XCDS) xs[x++] = n_order;
        xs[XCDS] = n_order;
    }
    // ---- per slice: chunks of its long rows (padded to whole workgroups), then its short rows; the
    //      planned rows' entries go into the packed stream in exactly that order
    std::vector<int32_t> long_row, long_nch, chunks, rowinfo, copyplan;
    SlicePlan sp{};
    rowinfo.reserve((size_t)n_order * 4 + 4 * SLICE_PAD * XCDS);
    copyplan.reserve((size_t)n_order * 4);
    int64_t n_pk = 0;
    for (int x = 0; x < XCDS; x++) {
        sp.cblk[x] = (int32_t)(chunks.size() / 16); sp.rows[x] = (int32_t)(rowinfo.size() / 4);
        const size_t slice_begin = rowinfo.size();
        for (int64_t p = xs[x]; p < xs[x + 1]; p++) {
            const int32_t r = row_order ? ord[(size_t)p] : (int32_t)p;
            const int32_t s0 = ip[(size_t)r], deg = ip[(size_t)r + 1] - s0;
            if (deg > LONG_T) {
                const int nch = (deg + long_ch - 1) / long_ch;
                const int32_t lb = (int32_t)n_pk;
                for (int k = 0; k < nch; k++) {
                    const int32_t c0 = k * long_ch, c1 = c0 + long_ch < deg ? c0 + long_ch : deg;
                    const int32_t e[4] = {(int32_t)long_row.size(), lb + c0, lb + c1, k};
                    chunks.insert(chunks.end(), e, e + 4);
                }
                const int32_t cp[4] = {s0, lb, deg, 0};
                copyplan.insert(copyplan.end(), cp, cp + 4);
                n_pk += deg;
                long_row.push_back(r); long_nch.push_back(nch);
            } else {
                const int32_t e[4] = {r, s0, deg, 0};
                rowinfo.insert(rowinfo.end(), e, e + 4);
            }
        }
        // rows of a pack run in lock step: sort every window of SHORT_WIN short rows by length (stable)
        for (size_t w0 = slice_begin; w0 < rowinfo.size(); w0 += 4 * SHORT_WIN) {
            const size_t w1 = std::min(rowinfo.size(), w0 + 4 * (size_t)SHORT_WIN), n = (w1 - w0) / 4;
            std::vector<std::array<int32_t, 4>> tmp(n);
            for (size_t i = 0; i < n; i++) for (int k = 0; k < 4; k++) tmp[i][k] = rowinfo[w0 + 4 * i + k];
            std::stable_sort(tmp.begin(), tmp.end(), [](const std::array<int32_t, 4> &p, const std::array<int32_t, 4> &q) { return p[2] > q[2]; });
            for (size_t i = 0; i < n; i++) for (int k = 0; k < 4; k++) rowinfo[w0 + 4 * i + k] = tmp[i][k];
        }
        // stream positions of the short rows, in their final order
        for (size_t i = slice_begin; i < rowinfo.size(); i += 4) {
            const int32_t cp[4] = {rowinfo[i + 1], (int32_t)n_pk, rowinfo[i + 2], 0};
            if (cp[2] > 0) copyplan.insert(copyplan.end(), cp, cp + 4);
            rowinfo[i + 1] = (int32_t)n_pk;
            n_pk += rowinfo[i + 2];
        }
        const int32_t pad[4] = {-1, 0, 0, 0};
        while ((chunks.size() / 4) % 4) chunks.insert(chunks.end(), pad, pad + 4);
        while ((rowinfo.size() / 4) % SLICE_PAD) rowinfo.insert(rowinfo.end(), pad, pad + 4);
    }
    sp.cblk[XCDS] = (int32_t)(chunks.size() / 16); sp.rows[XCDS] = (int32_t)(rowinfo.size() / 4);
    lgcn_graph *g = new (std::nothrow) lgcn_graph;
    if (!g) { lgcn_set_error("out of memory"); return 4; }
    g->indptr = indptr; g->indices = indices; g->vals = vals; g->rowinfo = nullptr; g->pk = nullptr;
    g->n_rows = n_rows; g->nnz = nnz; g->d_max = d_max;
    g->owned = nullptr; g->lp = LongPlan{}; g->sp = sp;
    g->used = false; g->last_stream = nullptr; g->order_ev = nullptr;
    if (hipEventCreateWithFlags(&g->order_ev, hipEventDisableTiming) != hipSuccess) { delete g; lgcn_set_error("lgcn_graph_create: hipEventCreate failed"); return 10; }
    const size_t n_long = long_row.size(), n_slots = chunks.size() / 4, n_info = rowinfo.size() / 4;
    {
        auto up = [](size_t x) { return (x + 255) & ~(size_t)255; };
        const size_t o_info = 0, o_chk = up(o_info + 16 * n_info), o_row = up(o_chk + 16 * n_slots),
                     o_nch = up(o_row + 4 * n_long), o_cnt = up(o_nch + 4 * n_long),
                     o_par = up(o_cnt + 4 * n_long), o_pk = up(o_par + n_slots * (size_t)d_max * 4),
                     total = o_pk + 8 * (size_t)n_pk + 256;
        char *base = nullptr;
        if (hipMalloc((void **)&base, total) != hipSuccess) { (void)hipEventDestroy(g->order_ev); delete g; lgcn_set_error("lgcn_graph_create: hipMalloc failed"); return 4; }
        g->owned = base;
        hipError_t e1 = hipMemset(base, 0, total);
        if (e1 == hipSuccess && n_info) e1 = hipMemcpy(base + o_info, rowinfo.data(), 16 * n_info, hipMemcpyHostToDevice);
        if (e1 == hipSuccess && n_long) {
            e1 = hipMemcpy(base + o_chk, chunks.data(), 16 * n_slots, hipMemcpyHostToDevice);
            if (e1 == hipSuccess) e1 = hipMemcpy(base + o_row, long_row.data(), 4 * n_long, hipMemcpyHostToDevice);
            if (e1 == hipSuccess) e1 = hipMemcpy(base + o_nch, long_nch.data(), 4 * n_long, hipMemcpyHostToDevice);
        }
        if (e1 == hipSuccess && !copyplan.empty()) {        // fill the packed stream on the device
            int4 *items = nullptr;
            e1 = hipMalloc((void **)&items, copyplan.size() * 4);
            if (e1 == hipSuccess) e1 = hipMemcpy(items, copyplan.data(), copyplan.size() * 4, hipMemcpyHostToDevice);
            if (e1 == hipSuccess) {
                const int64_t n_items = (int64_t)(copyplan.size() / 4), blocks = (n_items + 3) / 4;
                hipLaunchKernelGGL(k_pack_stream, dim3((unsigned)(blocks < 65536 ? blocks : 65536)), dim3(256), 0, 0,
                                   items, n_items, indices, vals, (int2 *)(base + o_pk));
                e1 = hipDeviceSynchronize();
            }
            if (items) (void)hipFree(items);
        }
        if (e1 != hipSuccess) { (void)hipFree(base); (void)hipEventDestroy(g->order_ev); delete g; lgcn_set_error("lgcn_graph_create: plan upload failed"); return 10; }
        g->rowinfo = (const int4 *)(base + o_info);
        g->pk = (const int2 *)(base + o_pk);
        if (n_long) {
            g->lp.chunks = (const int4 *)(base + o_chk);
            g->lp.long_row = (const int32_t *)(base + o_row); g->lp.long_nch = (const int32_t *)(base + o_nch);
            g->lp.counters = (int32_t *)(base + o_cnt);
            g->lp.partials = (float *)(base + o_par);
            g->lp.n_long = (int32_t)n_long; g->lp.n_chunk_slots = (int32_t)n_slots;
        }
    }
    *out = g;
    return 0;
}

extern "C" void lgcn_graph_destroy(lgcn_graph *g) {
    if (!g) return;
    if (g->owned) (void)hipFree(g->owned);
    if (g->order_ev) (void)hipEventDestroy(g->order_ev);
    delete g;
}

static SpmmArgs graph_spmm(const lgcn_graph *g) {
    SpmmArgs a{};
    a.pk = g->pk; a.rowinfo = g->rowinfo; a.lp = g->lp; a.sp = g->sp; a.n_rows = g->n_rows;
    return a;
}

extern "C" int lgcn_spmm_csr(const lgcn_graph *g, const void *X, int x_dtype, void *Y, int y_dtype, int d, void *stream) {
    if (!g || !X || !Y) { lgcn_set_error("lgcn_spmm_csr: null/invalid argument"); return 3; }
    if (check_dtype(x_dtype) || check_dtype(y_dtype)) return 3;
    if (d > g->d_max) { lgcn_set_error("lgcn_spmm_csr: d exceeds the graph's d_max"); return 3; }
    SpmmArgs a = graph_spmm(g);
    a.X = X; a.Y = Y; a.remap = 1;
    int rc = graph_acquire(g, (hipStream_t)stream);
    if (rc) return rc;
    rc = launch_spmm<0>(a, d, x_dtype, y_dtype, (hipStream_t)stream);
    if (rc) return rc;
    HIP_OK(hipGetLastError());
    return 0;
}

// bytes of one [N,d] table of a storage type (fp8: rows + fp32 row scales, padded so that the next table stays 256-byte aligned)
static inline size_t table_bytes(int64_t N, int d, int dtype) {
    if (dtype == LGCN_FP8) return (((size_t)N * d + (size_t)N * 4) + 255) & ~(size_t)255;
    return (size_t)N * d * (dtype == LGCN_BF16 ? 2 : 4);
}
extern "C" int64_t lgcn_table_bytes(int64_t n_rows, int32_t d, int32_t dtype) { return n_rows > 0 && d > 0 ? (int64_t)table_bytes(n_rows, d, dtype) : 0; }
extern "C" int lgcn_to_fp8(const float *src, void *dst, int64_t n_rows, int32_t d, void *stream) {
    if (!src || !dst || n_rows <= 0) { lgcn_set_error("lgcn_to_fp8: invalid argument"); return 3; }
    int rc = launch_to_fp8(src, dst, n_rows, d, (hipStream_t)stream);
    if (rc) return rc;
    HIP_OK(hipGetLastError());
    return 0;
}

extern "C" int lgcn_propagate_mean(const lgcn_graph *g, const float *E0, int K, int d, int act_dtype, void *work,
                                   float *out, void *stream) {
    if (!g || !E0 || !out || K < 1 || K > LGCN_MAX_LAYERS) { lgcn_set_error("lgcn_propagate_mean: invalid argument"); return 3; }
    if (K > 1 && !work) { lgcn_set_error("lgcn_propagate_mean: workspace required for K > 1"); return 3; }
    if (check_dtype(act_dtype)) return 3;
    if (d > g->d_max) { lgcn_set_error("lgcn_propagate_mean: d exceeds the graph's d_max"); return 3; }
    hipStream_t st = (hipStream_t)stream;
    { int rc0 = graph_acquire(g, st); if (rc0) return rc0; }
    const int64_t N = g->n_rows;
    MeanArgs m{};
    m.X0 = E0; m.K = K; m.out = out; m.n4 = N * d / 4; m.d = d; m.N = N;
    const size_t stride = table_bytes(N, d, act_dtype);
    const void *prev = E0; int prev_dtype = LGCN_F32;
    if (act_dtype == LGCN_BF16 && K >= 2) {
        // the same rule as the training step: with bf16 activation storage layer 1 reads bf16(E0).  `out` is free until
        // the last layer writes it (K >= 2): its memory holds the 2-byte copy meanwhile
        launch_to_bf16(E0, (bf16_t *)out, N * d, st);
        prev = out; prev_dtype = LGCN_BF16;
    }
    if (act_dtype == LGCN_FP8 && K >= 2) {                 // the same with fp8 storage: rows + scales fit the fp32 `out`
        int rc = launch_to_fp8(E0, out, N, d, st);
        if (rc) return rc;
        prev = out; prev_dtype = LGCN_FP8;
    }
    for (int k = 1; k <= K; k++) {
        const bool last = (k == K);
        void *y = last ? (void *)out : (void *)((char *)work + (size_t)(k - 1) * stride);
        SpmmArgs a = graph_spmm(g);
        a.X = prev; a.Y = y; a.remap = 1;
        int rc = launch_spmm<0>(a, d, prev_dtype, last ? LGCN_F32 : act_dtype, st);
        if (rc) return rc;
        if (!last) { m.Xl[k] = y; prev = y; prev_dtype = act_dtype; }
    }
    const int64_t blocks = (m.n4 + 255) / 256;
    const unsigned grid = (unsigned)(blocks < 2048 ? blocks : 2048);
    if (act_dtype == LGCN_F32) hipLaunchKernelGGL((k_layer_mean<float>), dim3(grid), dim3(256), 0, st, m);
    else if (act_dtype == LGCN_BF16) hipLaunchKernelGGL((k_layer_mean<bf16_t>), dim3(grid), dim3(256), 0, st, m);
    else hipLaunchKernelGGL((k_layer_mean<fp8_t>), dim3(grid), dim3(256), 0, st, m);
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
    int64_t N;
    int64_t bm_words;             // words per bitmap; cfg.bitmap holds two, used alternately
    int flip;
    void *act[LGCN_MAX_LAYERS + 1];   // act[k] = X_k storage for k = 1..K-1 (K with dense_last; also reused for H)
    int fwd_layers;               // dense forward layers per step: K-1, or K with dense_last
    float *g32;                   // [N,d] fp32 copy of the step's flagged gradient rows (library-owned; k_g32)
    bf16_t *e0b;                  // [N,d] bf16 copy of E0 (library-owned; bf16 activation storage with K >= 2 only)
    void *e0q;                    // fp8 copy of E0, rows + scales (library-owned; fp8 activation storage with K >= 2 only); e0b_fresh covers it too
    bool e0b_fresh;               // e0b == bf16(E0) right now (set by the Adam epilogue inside a multi-step call)
    bool in_loop;                 // inside lgcn_train_epoch / lgcn_train_epoch_dp: the Adam epilogue keeps e0b current
    lgcn_graph *hub_graph;        // plan over the rows with more than hub_nnz non-zeros (or NULL): their last-layer rows are
    int32_t hub_nnz;              //   computed by k_spmm into g32 before k_triplet instead of by the one workgroup of a triplet
    int64_t hub_rows;             // rows in that plan
    bool dp_local;                // data parallel (rows): part 1 also adds this rank's own rows into G64, part 2 scatters the others'
    int dp_rank;                  // rank of the last part 1
    // optional branches of the fork (popularity gate / item-item smoothing): the step runs on ONE final table
    bool variant;                 // either branch is on
    float *tvar;                  // [N,d] fp32 final propagated table T (library-owned)
    uint32_t *item_bitmap;        // [ceil(m_items/32)] items named by the batch (item-item backward), library-owned
    long long *gate_partials;     // [n_wg, P] parameter-gradient partial sums, fixed point (library-owned)
    long long *gate_total;        // [P] this rank's sum of them, all-reduced by the dense data-parallel form (behind gate_partials)
    int32_t gate_P, gate_wgs;
    int32_t *cnt;                 // reg_ego: [N] slots of the running step naming each row (library-owned, zero between steps)
    float *colsum;                // [3 * max_batch] partial scores / reg terms of a column-sharded step (library-owned)
};
// a multi-step call: nobody but this library touches E0 between its steps
struct LoopScope {
    lgcn_ctx *x;
    explicit LoopScope(lgcn_ctx *x_) : x(x_) { x->in_loop = true; x->e0b_fresh = false; }
    ~LoopScope() { x->in_loop = false; x->e0b_fresh = false; }
};

extern "C" int lgcn_ctx_create(const lgcn_train_config *cfg, lgcn_ctx **out) {
    if (!cfg || !out) { lgcn_set_error("lgcn_ctx_create: null argument"); return 3; }
    const lgcn_train_config &c = *cfg;
    if (!c.graph || !c.E0 || !c.adam_m || !c.adam_v || !c.G64 ||
        !c.bitmap || !c.terms || !c.err) { lgcn_set_error("lgcn_ctx_create: null buffer"); return 3; }
    if (c.K < 1 || c.K > LGCN_MAX_LAYERS) { lgcn_set_error("lgcn_ctx_create: K out of range"); return 3; }
    if ((c.K > 1 || c.dense_last) && !c.act) { lgcn_set_error("lgcn_ctx_create: activation workspace missing"); return 3; }
    if (c.d != 32 && c.d != 64 && c.d != 128 && c.d != 256) { lgcn_set_error("embedding dim must be 32, 64, 128 or 256"); return 3; }
    if (check_dtype(c.act_dtype)) return 3;
    if (c.act_dtype == LGCN_FP8 && c.d < 64) { lgcn_set_error("lgcn_ctx_create: fp8 activation storage needs an embedding dim of 64, 128 or 256"); return 3; }
    if (c.act_dtype == LGCN_FP8 && (c.item_pop || c.i2i)) { lgcn_set_error("lgcn_ctx_create: fp8 activation storage is implemented for the default model, not with the optional branches"); return 3; }
    if (c.n_users <= 0 || c.n_users >= c.graph->n_rows || c.max_batch <= 0) { lgcn_set_error("lgcn_ctx_create: bad sizes"); return 3; }
    if (c.d > c.graph->d_max) { lgcn_set_error("lgcn_ctx_create: d exceeds the graph's d_max"); return 3; }
    const bool gate = c.item_pop != nullptr, smooth = c.i2i != nullptr;
    if (gate || smooth) {
        if (!c.dense_last) { lgcn_set_error("lgcn_ctx_create: the optional branches need dense_last = 1 (every layer propagated densely)"); return 3; }
        const int64_t m_items = c.graph->n_rows - c.n_users;
        if (smooth && (!c.i2i_t || c.i2i->n_rows != m_items || c.i2i_t->n_rows != m_items || c.i2i->d_max < c.d || c.i2i_t->d_max < c.d)) {
            lgcn_set_error("lgcn_ctx_create: i2i / i2i_t must be [m_items, m_items] graphs with d_max >= d"); return 3; }
        if (gate && (!c.gate_params || !c.gate_adam_m || !c.gate_adam_v || !c.gate_grad || c.pop_hidden < 1 || c.pop_hidden > GATE_HMAX ||
                     c.gate_hidden < 1 || c.gate_hidden > GATE_HMAX || c.d > 128 || !(c.pop_gate_temp > 0.f))) {
            lgcn_set_error("lgcn_ctx_create: popularity gate needs its parameter / Adam buffers, hidden sizes in 1..64, d <= 128, temperature > 0"); return 3; }
    }
    if (c.reg_ego && (gate || smooth)) { lgcn_set_error("lgcn_ctx_create: reg_ego (upstream's L2 term) is defined for the default model only, not with the optional branches"); return 3; }
    lgcn_ctx *x = new (std::nothrow) lgcn_ctx;
    if (!x) { lgcn_set_error("out of memory"); return 4; }
    x->c = c; x->step = 0; x->N = c.graph->n_rows; x->cnt = nullptr; x->colsum = nullptr;
    x->bm_words = (x->N + 31) / 32; x->flip = 0;
    const size_t stride = table_bytes(x->N, c.d, c.act_dtype);
    for (int k = 0; k <= LGCN_MAX_LAYERS; k++) x->act[k] = nullptr;
    x->fwd_layers = c.dense_last ? c.K : c.K - 1;
    for (int k = 1; k <= x->fwd_layers; k++) x->act[k] = (char *)c.act + (size_t)(k - 1) * stride;
    // rows that are never flagged are never read; zero-filled so that the zero-weight padding reads of row 0 stay finite
    x->g32 = nullptr; x->e0b = nullptr; x->e0q = nullptr; x->e0b_fresh = false; x->in_loop = false; x->dp_local = false; x->dp_rank = -1;
    x->hub_graph = nullptr; x->hub_nnz = 0; x->hub_rows = 0;
    x->variant = gate || smooth; x->tvar = nullptr; x->item_bitmap = nullptr; x->gate_partials = nullptr; x->gate_total = nullptr; x->gate_P = 0; x->gate_wgs = 0;
    const size_t gbytes = (size_t)x->N * c.d * sizeof(float);
    bool ok = hipMalloc((void **)&x->g32, gbytes) == hipSuccess && hipMemset(x->g32, 0, gbytes) == hipSuccess;
    if (ok) ok = hipMalloc((void **)&x->colsum, sizeof(float) * 3 * (size_t)c.max_batch) == hipSuccess;
    if (ok && c.reg_ego) ok = hipMalloc((void **)&x->cnt, sizeof(int32_t) * (size_t)x->N) == hipSuccess && hipMemset(x->cnt, 0, sizeof(int32_t) * (size_t)x->N) == hipSuccess;
    if (ok && x->variant) ok = hipMalloc((void **)&x->tvar, gbytes) == hipSuccess;
    if (ok && smooth) {
        const size_t wb = sizeof(uint32_t) * (size_t)((x->N - c.n_users + 31) / 32);
        ok = hipMalloc((void **)&x->item_bitmap, wb) == hipSuccess && hipMemset(x->item_bitmap, 0, wb) == hipSuccess;
    }
    if (ok && gate) {
        x->gate_P = 2 * c.pop_hidden + c.d * c.pop_hidden + c.d + 2 * c.d * c.gate_hidden + 2 * c.gate_hidden + 1;
        x->gate_wgs = (c.max_batch + GATE_TPB - 1) / GATE_TPB;
        ok = hipMalloc((void **)&x->gate_partials, sizeof(long long) * ((size_t)x->gate_wgs + 1) * x->gate_P) == hipSuccess;
        if (ok) x->gate_total = x->gate_partials + (size_t)x->gate_wgs * x->gate_P;
    }
    if (ok && c.act_dtype == LGCN_BF16 && c.K >= 2) ok = hipMalloc((void **)&x->e0b, gbytes / 2) == hipSuccess;
    if (ok && c.act_dtype == LGCN_FP8 && c.K >= 2) ok = hipMalloc(&x->e0q, table_bytes(x->N, c.d, LGCN_FP8)) == hipSuccess;
    if (ok && !c.dense_last) {
        // Rows too long for one workgroup (a 800 000-neighbour item of the 10M x 1M graph kept ONE k_triplet workgroup busy
        // for 37 ms, several times per batch): a plan over just those rows; k_spmm computes their last-layer rows for
        // the whole chip to share, once per step, before k_triplet.
        const int32_t thr = c.hub_nnz == 0 ? TRIPLET_HUB_NNZ : c.hub_nnz;
        if (thr > 0) {
            std::vector<int32_t> ip((size_t)x->N + 1), hubs;
            ok = hipMemcpy(ip.data(), c.graph->indptr, sizeof(int32_t) * ip.size(), hipMemcpyDeviceToHost) == hipSuccess;
            for (int64_t r = 0; ok && r < x->N; r++) if (ip[(size_t)r + 1] - ip[(size_t)r] > thr) hubs.push_back((int32_t)r);
            if (ok && !hubs.empty()) {
                int32_t *dh = nullptr;
                ok = hipMalloc((void **)&dh, sizeof(int32_t) * hubs.size()) == hipSuccess &&
                     hipMemcpy(dh, hubs.data(), sizeof(int32_t) * hubs.size(), hipMemcpyHostToDevice) == hipSuccess;
                const int32_t hub_chunk = c.hub_chunk > 0 ? c.hub_chunk : TRIPLET_HUB_CHUNK;
                if (ok) ok = graph_create_impl(c.graph->indptr, c.graph->indices, c.graph->vals, x->N, c.graph->nnz, c.d, dh,
                                               (int64_t)hubs.size(), nullptr, hub_chunk, &x->hub_graph) == 0;
                if (dh) (void)hipFree(dh);
                x->hub_nnz = thr; x->hub_rows = (int64_t)hubs.size();
            }
        }
    }
    if (!ok) {
        if (x->g32) (void)hipFree(x->g32);
        if (x->colsum) (void)hipFree(x->colsum);
        if (x->cnt) (void)hipFree(x->cnt);
        if (x->tvar) (void)hipFree(x->tvar);
        if (x->item_bitmap) (void)hipFree(x->item_bitmap);
        if (x->gate_partials) (void)hipFree(x->gate_partials);
        if (x->e0b) (void)hipFree(x->e0b);
        if (x->e0q) (void)hipFree(x->e0q);
        if (x->hub_graph) lgcn_graph_destroy(x->hub_graph);
        delete x;
        lgcn_set_error("lgcn_ctx_create: cannot allocate the library-owned tables (N*d*4 bytes fp32 gradient rows, N*d*2 bf16 parameters)");
        return 4;
    }
    *out = x;
    return 0;
}
extern "C" void lgcn_ctx_destroy(lgcn_ctx *ctx) {
    if (!ctx) return;
    if (ctx->g32) (void)hipFree(ctx->g32);
    if (ctx->colsum) (void)hipFree(ctx->colsum);
    if (ctx->cnt) (void)hipFree(ctx->cnt);
    if (ctx->tvar) (void)hipFree(ctx->tvar);
    if (ctx->item_bitmap) (void)hipFree(ctx->item_bitmap);
    if (ctx->gate_partials) (void)hipFree(ctx->gate_partials);
    if (ctx->e0b) (void)hipFree(ctx->e0b);
    if (ctx->e0q) (void)hipFree(ctx->e0q);
    if (ctx->hub_graph) lgcn_graph_destroy(ctx->hub_graph);
    delete ctx;
}
extern "C" int lgcn_ctx_set_dp_local(lgcn_ctx *ctx, int on) {
    if (!ctx) { lgcn_set_error("lgcn_ctx_set_dp_local: null context"); return 3; }
    ctx->dp_local = on != 0; ctx->dp_rank = -1;
    return 0;
}
extern "C" int64_t lgcn_ctx_get_step(const lgcn_ctx *ctx) { return ctx ? ctx->step : -1; }
extern "C" int64_t lgcn_ctx_hub_rows(const lgcn_ctx *ctx) { return (ctx && ctx->hub_graph) ? ctx->hub_rows : 0; }
extern "C" void lgcn_ctx_set_step(lgcn_ctx *ctx, int64_t s) { if (ctx) ctx->step = s; }
extern "C" void lgcn_ctx_set_lr(lgcn_ctx *ctx, double lr) { if (ctx) ctx->c.lr = lr; }

static SpmmArgs base_spmm(const lgcn_ctx *x) {
    SpmmArgs a = graph_spmm(x->c.graph);
    a.G64 = (long long *)x->c.G64; a.G32 = x->g32; a.bitmap = x->c.bitmap + x->flip * x->bm_words;
    a.div = (float)(x->c.K + 1); a.remap = x->c.xcd_remap;
    return a;
}

// forward layers X_1..X_{K-1}
static int run_forward(lgcn_ctx *x, hipStream_t st) {
    const lgcn_train_config &c = x->c;
    { int rc0 = graph_acquire(c.graph, st); if (rc0) return rc0; }
    const void *prev = c.E0; int prev_dt = LGCN_F32;
    if (x->e0b && x->fwd_layers >= 1) {                 // bf16 activation storage: layer 1 gathers bf16(E0)
        if (!x->in_loop) x->e0b_fresh = false;          // a single-step call: E0 may have been written by anybody since
        if (!x->e0b_fresh) launch_to_bf16(c.E0, x->e0b, x->N * c.d, st);
        x->e0b_fresh = true;
        prev = x->e0b; prev_dt = LGCN_BF16;
    }
    if (x->e0q && x->fwd_layers >= 1) {                 // fp8 activation storage: layer 1 gathers fp8(E0)
        if (!x->in_loop) x->e0b_fresh = false;
        if (!x->e0b_fresh) { int rc = launch_to_fp8(c.E0, x->e0q, x->N, c.d, st); if (rc) return rc; }
        x->e0b_fresh = true;
        prev = x->e0q; prev_dt = LGCN_FP8;
    }
    for (int k = 1; k <= x->fwd_layers; k++) {
        SpmmArgs a = base_spmm(x);
        a.X = prev; a.Y = x->act[k];
        int rc = launch_spmm<0>(a, c.d, prev_dt, c.act_dtype, st);
        if (rc) return rc;
        prev = x->act[k]; prev_dt = c.act_dtype;
    }
    return 0;
}

static int run_bpr(lgcn_ctx *x, const int32_t *users, const int32_t *pos, const int32_t *neg,
                   int32_t B_global, int32_t b_off, int32_t B_local, int32_t shard, bool atomics, bool exchange, hipStream_t st,
                   bool count_slots = true, int cols_phase = 0) {
    const lgcn_train_config &c = x->c;
    BprArgs a{};
    a.reg_ego = c.reg_ego; a.cnt = (atomics && count_slots) ? x->cnt : nullptr;
    a.cols_phase = cols_phase; a.erows = c.contrib; a.colsum = x->colsum;
    a.indptr = c.graph->indptr; a.indices = c.graph->indices; a.vals = c.graph->vals; a.X0 = c.E0; a.K = c.K;
    for (int k = 1; k <= x->fwd_layers; k++) a.Xl[k] = x->act[k];
    a.dense_last = c.dense_last;
    a.n_users = c.n_users; a.N = x->N;
    a.users = users + b_off; a.pos = pos + b_off; a.neg = neg + b_off;
    a.B_local = B_local; a.shard = shard;
    a.inv_B = 1.0f / (float)B_global; a.lam = c.decay / (float)B_global;
    a.G64 = atomics ? (long long *)c.G64 : nullptr; a.bitmap = c.bitmap + x->flip * x->bm_words;
    a.stale_bitmap = c.bitmap + (x->flip ^ 1) * x->bm_words; a.bitmap_words = x->bm_words;
    a.contrib = c.contrib; a.terms = c.terms; a.err = c.err; a.exchange = exchange ? 1 : 0;
    a.terms_off = exchange ? 0 : b_off; a.terms_stride = B_global;
    if (cols_phase == 2) {      // column shard, after the all-reduce: loss terms + gradient scatter from the stored slot rows
        if (B_local <= 0) return 0;
        DISPATCH_D(c.d, {
            const int tpb = 256 / (D < 64 ? D : 64);
            hipLaunchKernelGGL((k_cols_finish<D>), dim3((unsigned)((B_local + tpb - 1) / tpb)), dim3(256), 0, st, a);
        });
        return 0;
    }
    if (x->hub_graph && !c.dense_last && B_local > 0) {
        // last layer of the hub rows, X_K[hub] = (A_hat X_{K-1})[hub] in fp32, into the library's [N,d] table (free until k_g32)
        { int rc0 = graph_acquire(x->hub_graph, st); if (rc0) return rc0; }
        SpmmArgs h = graph_spmm(x->hub_graph);
        h.X = c.K == 1 ? (const void *)c.E0 : x->act[c.K - 1]; h.Y = x->g32; h.remap = c.xcd_remap;
        int rc = launch_spmm<0>(h, c.d, c.K == 1 ? LGCN_F32 : c.act_dtype, LGCN_F32, st);
        if (rc) return rc;
        a.Xhub = x->g32; a.hub_nnz = x->hub_nnz;
    }
    if (B_local <= 0) {
        // a rank whose shard of a short last batch is empty launches nothing, but the row bitmap of
        // two steps ago still has to be cleared (k_triplet / k_triplet_dense does it on the other ranks)
        HIP_OK(hipMemsetAsync(a.stale_bitmap, 0, sizeof(uint32_t) * (size_t)x->bm_words, st));
        return 0;
    }
    DISPATCH_D(c.d, {
        if (c.act_dtype == LGCN_FP8) {
            if constexpr (D >= 64) {
                if (c.dense_last) {
                    const int tpb = 256 / 64;
                    hipLaunchKernelGGL((k_triplet_dense<D, fp8_t>), dim3((unsigned)((B_local + tpb - 1) / tpb)), dim3(256), 0, st, a);
                } else if (big_table(x->N, D)) hipLaunchKernelGGL((k_triplet<D, fp8_t, true>), dim3(B_local), dim3(64 * TripletGeo<fp8_t>::NW), 0, st, a);
                else hipLaunchKernelGGL((k_triplet<D, fp8_t, false>), dim3(B_local), dim3(64 * TripletGeo<fp8_t>::NW), 0, st, a);
            }
        } else if (c.dense_last) {
            const int tpb = 256 / (D < 64 ? D : 64);
            const unsigned gd = (unsigned)((B_local + tpb - 1) / tpb);
            if (c.act_dtype == LGCN_F32) hipLaunchKernelGGL((k_triplet_dense<D, float>), dim3(gd), dim3(256), 0, st, a);
            else hipLaunchKernelGGL((k_triplet_dense<D, bf16_t>), dim3(gd), dim3(256), 0, st, a);
        } else if (big_table(x->N, D)) {
            if (c.act_dtype == LGCN_F32) hipLaunchKernelGGL((k_triplet<D, float, true>), dim3(B_local), dim3(64 * TripletGeo<float>::NW), 0, st, a);
            else hipLaunchKernelGGL((k_triplet<D, bf16_t, true>), dim3(B_local), dim3(64 * TripletGeo<bf16_t>::NW), 0, st, a);
        } else if (c.act_dtype == LGCN_F32) hipLaunchKernelGGL((k_triplet<D, float, false>), dim3(B_local), dim3(64 * TripletGeo<float>::NW), 0, st, a);
        else hipLaunchKernelGGL((k_triplet<D, bf16_t, false>), dim3(B_local), dim3(64 * TripletGeo<bf16_t>::NW), 0, st, a);
    });
    return 0;
}

// floats per rank of the data-parallel exchange block for a shard of S triplets: [3*S*d gradient rows | S loss | S reg terms], with
// the popularity gate also [S entropy terms | pad to an even count | 2*P floats = P int64: the rank's fixed-point MLP gradient sums]
static int64_t dp_block_floats(const lgcn_ctx *x, int64_t S) {
    int64_t n = 3 * S * x->c.d + 2 * S;
    if (x->variant && x->c.item_pop) { n += S; n += n & 1; n += 2 * (int64_t)x->gate_P; }
    return n;
}
static int64_t dp_block_tail(const lgcn_ctx *x, int64_t S) {          // float offset of the int64 tail inside a block
    int64_t n = 3 * S * x->c.d + 3 * S;
    return n + (n & 1);
}
extern "C" int64_t lgcn_dp_block_floats(const lgcn_ctx *x, int32_t B_global, int32_t world) {
    if (!x || B_global <= 0 || world < 1) return 0;
    return dp_block_floats(x, (B_global + world - 1) / world);
}

// The optional branches' forward tail and loss: T = mean of the K+1 dense layers, item-item smoothing, then the loss on the
// ONE final table (popularity gate inside k_triplet_gate).  The batch slice [b_off, b_off + B_local) of a global batch of
// B_global triplets (single GPU: the whole batch); atomics: into G64 / the row flags; exchange: into the exchange block.
static int run_variant_loss(lgcn_ctx *x, const int32_t *users, const int32_t *pos, const int32_t *neg, int32_t B_global,
                            int32_t b_off, int32_t B_local, int32_t shard, bool atomics, bool exchange, hipStream_t st) {
    const lgcn_train_config &c = x->c;
    const int64_t m_items = x->N - c.n_users;
    float *items_mean = c.i2i ? x->g32 : x->tvar;           // with smoothing the item block of T is an SpMM input first
    MeanSplitArgs m{};
    m.X0 = c.E0; m.K = c.K; m.out_u = x->tvar; m.out_i = items_mean;
    for (int k = 1; k <= c.K; k++) m.Xl[k] = x->act[k];
    m.n4_users = (int64_t)c.n_users * c.d / 4; m.n4 = x->N * c.d / 4;
    {
        const int64_t blocks = (m.n4 + 255) / 256;
        const unsigned grid = (unsigned)(blocks < 2048 ? blocks : 2048);
        if (c.act_dtype == LGCN_F32) hipLaunchKernelGGL((k_mean_layers<float>), dim3(grid), dim3(256), 0, st, m);
        else hipLaunchKernelGGL((k_mean_layers<bf16_t>), dim3(grid), dim3(256), 0, st, m);
    }
    if (c.i2i) {            // items = T_items + (alpha I2I) T_items   (model.py:228-229)
        { int rc0 = graph_acquire(c.i2i, st); if (rc0) return rc0; }
        SpmmArgs h = graph_spmm(c.i2i);
        h.X = x->g32 + (int64_t)c.n_users * c.d; h.selfX = (const float *)h.X; h.Y = x->tvar + (int64_t)c.n_users * c.d; h.remap = c.xcd_remap;
        int rc = launch_spmm_addself(h, c.d, st);
        if (rc) return rc;
        HIP_OK(hipMemsetAsync(x->item_bitmap, 0, sizeof(uint32_t) * (size_t)((m_items + 31) / 32), st));
    }
    uint32_t *bm = c.bitmap + x->flip * x->bm_words, *stale = c.bitmap + (x->flip ^ 1) * x->bm_words;
    const int64_t tail = dp_block_tail(x, shard);
    if (B_local <= 0) {     // an empty shard: nothing to launch, but the stale bitmap must go and the block's MLP sums must read zero
        HIP_OK(hipMemsetAsync(stale, 0, sizeof(uint32_t) * (size_t)x->bm_words, st));
        if (exchange && c.item_pop) HIP_OK(hipMemsetAsync(c.contrib + tail, 0, sizeof(long long) * (size_t)x->gate_P, st));
        return 0;
    }
    if (c.item_pop) {
        GateArgs g{};
        g.E = x->tvar; g.item_pop = c.item_pop; g.params = c.gate_params; g.partials = x->gate_partials;
        g.Hp = c.pop_hidden; g.Hg = c.gate_hidden; g.P = x->gate_P;
        g.inv_temp = 1.0f / c.pop_gate_temp; g.ent_scale = c.gate_entropy_coeff / (float)(2 * B_global);
        g.n_users = c.n_users; g.N = x->N; g.users = users + b_off; g.pos = pos + b_off; g.neg = neg + b_off; g.B = B_local;
        g.inv_B = 1.0f / (float)B_global; g.lam = c.decay / (float)B_global;
        g.G64 = atomics ? (long long *)c.G64 : nullptr; g.bitmap = bm; g.item_bitmap = c.i2i ? x->item_bitmap : nullptr;
        g.stale_bitmap = stale; g.bitmap_words = x->bm_words; g.terms = c.terms + (exchange ? 0 : b_off); g.terms_stride = B_global; g.err = c.err;
        g.contrib = c.contrib; g.shard = shard; g.exchange = exchange ? 1 : 0;
        const unsigned grid = (unsigned)((B_local + GATE_TPB - 1) / GATE_TPB);
        const size_t lds = gate_lds_bytes(c.d);
        switch (c.d) {
        case 32: HIP_OK(hipFuncSetAttribute((const void *)k_triplet_gate<32>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
                 hipLaunchKernelGGL((k_triplet_gate<32>), dim3(grid), dim3(256), lds, st, g); break;
        case 64: HIP_OK(hipFuncSetAttribute((const void *)k_triplet_gate<64>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
                 hipLaunchKernelGGL((k_triplet_gate<64>), dim3(grid), dim3(256), lds, st, g); break;
        case 128: HIP_OK(hipFuncSetAttribute((const void *)k_triplet_gate<128>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
                 hipLaunchKernelGGL((k_triplet_gate<128>), dim3(grid), dim3(256), lds, st, g); break;
        default: lgcn_set_error("popularity gate: embedding dim must be 32, 64 or 128"); return 3;
        }
        if (exchange)       // this rank's MLP gradient sums into the tail of its exchange block
            hipLaunchKernelGGL(k_gate_rank_total, dim3((unsigned)((x->gate_P + 255) / 256)), dim3(256), 0, st,
                               (const long long *)x->gate_partials, (int32_t)grid, x->gate_P, (long long *)(c.contrib + tail));
    } else {
        BprArgs a{};
        a.X0 = x->tvar; a.K = 0; a.dense_last = 1;            // e = the row of the final table
        a.n_users = c.n_users; a.N = x->N; a.users = users + b_off; a.pos = pos + b_off; a.neg = neg + b_off; a.B_local = B_local; a.shard = shard;
        a.inv_B = 1.0f / (float)B_global; a.lam = c.decay / (float)B_global;
        a.G64 = atomics ? (long long *)c.G64 : nullptr; a.bitmap = bm; a.stale_bitmap = stale; a.bitmap_words = x->bm_words;
        a.item_bitmap = x->item_bitmap; a.terms = c.terms; a.err = c.err; a.terms_off = exchange ? 0 : b_off; a.terms_stride = B_global;
        a.contrib = c.contrib; a.exchange = exchange ? 1 : 0;
        DISPATCH_D(c.d, {
            const int tpb = 256 / (D < 64 ? D : 64);
            hipLaunchKernelGGL((k_triplet_dense<D, float>), dim3((unsigned)((B_local + tpb - 1) / tpb)), dim3(256), 0, st, a);
        });
    }
    return 0;
}

static SlotArgs slot_args(const lgcn_ctx *x, const int32_t *users, const int32_t *pos, const int32_t *neg, int32_t B,
                          const float *gathered, int32_t shard, int32_t world, float *loss_out) {
    const lgcn_train_config &c = x->c;
    SlotArgs s{};
    s.users = users; s.pos = pos; s.neg = neg; s.B = B; s.n_users = c.n_users; s.N = x->N;
    s.G64 = (long long *)c.G64; s.G32 = x->g32; s.div = (float)(c.K + 1);
    s.bitmap = c.bitmap + x->flip * x->bm_words; s.gathered = gathered; s.shard = shard; s.world = world;
    s.terms = c.terms; s.loss_out = loss_out; s.decay = c.decay; s.skip_rank = -1;
    s.ent_coeff = c.item_pop ? c.gate_entropy_coeff : 0.f;
    s.blk = gathered ? dp_block_floats(x, shard) : 0;
    s.item_bitmap = (x->variant && c.i2i) ? x->item_bitmap : nullptr;
    s.cnt = x->cnt;
    return s;
}
static unsigned scatter_grid(const lgcn_ctx *x, int32_t B) {      // k_scatter: one lane group of min(d, 64) lanes per slot
    const int spb = 256 / (x->c.d < 64 ? x->c.d : 64);
    return (unsigned)((3 * (int64_t)B + spb - 1) / spb);
}
static unsigned slot_grid(const lgcn_ctx *x, int32_t B) {
    const int spb = 256 / (x->c.d / 4);
    return (unsigned)((3 * (int64_t)B + spb - 1) / spb);
}

// buffer the backward layer k writes (k > 1): ping-pong inside the activation workspace (forward
// activations are dead by then)
static void *bwd_buffer(const lgcn_ctx *x, int k) {
    return (x->c.K == 2) ? x->act[1] : x->act[1 + ((x->c.K - k) & 1)];
}

// One layer of the Horner chain h_{k-1} = Gs + A h_k (k = K: sparse input Gs; k = 1: feeds Adam).
// fused_finish: the Adam launch also zeroes the G64 rows it consumes and reduces the loss (single-GPU
// and batch-sharded steps, where every rank runs every row); the row-sharded step finishes separately.
static int backward_layer(lgcn_ctx *x, int k, const int32_t *users, const int32_t *pos, const int32_t *neg, int32_t B,
                          const float *gathered, int32_t shard, float *loss_out, bool fused_finish, hipStream_t st) {
    const lgcn_train_config &c = x->c;
    const bool first = (k == c.K), last = (k == 1);
    if (first) {        // every contribution is in G64 by now: convert the batch rows once
        SlotArgs s = slot_args(x, users, pos, neg, B, gathered, shard, 1, loss_out);
        DISPATCH_D(c.d, hipLaunchKernelGGL((k_g32<D>), dim3(slot_grid(x, B)), dim3(256), 0, st, s));
        if (x->variant && c.i2i) {
            // backward of the smoothing: Gs_items <- Gs_items + (alpha I2I)^T Gs_items, a sparse-input SpMM on the item block; the
            // result is dense, so every item row of Gs counts as non-zero from here on
            { int rc0 = graph_acquire(c.i2i_t, st); if (rc0) return rc0; }
            const int64_t ioff = (int64_t)c.n_users * c.d, m_items = x->N - c.n_users;
            SpmmArgs h = graph_spmm(c.i2i_t);
            h.G32 = x->g32 + ioff; h.bitmap = x->item_bitmap; h.Y = x->tvar + ioff; h.div = (float)(c.K + 1); h.remap = c.xcd_remap;
            int rc = launch_spmm<M_SPARSE | M_ADDG>(h, c.d, LGCN_F32, LGCN_F32, st);
            if (rc) return rc;
            HIP_OK(hipMemcpyAsync(x->g32 + ioff, x->tvar + ioff, sizeof(float) * (size_t)m_items * c.d, hipMemcpyDeviceToDevice, st));
            const int64_t words = (x->N + 31) / 32;
            hipLaunchKernelGGL(k_flag_range, dim3((unsigned)((words + 255) / 256 < 64 ? (words + 255) / 256 : 64)), dim3(256), 0, st,
                               c.bitmap + x->flip * x->bm_words, (int64_t)c.n_users, x->N);
        }
    }
    SpmmArgs a = base_spmm(x);
    a.X = first ? nullptr : bwd_buffer(x, k + 1);
    if (!last) a.Y = bwd_buffer(x, k);
    else {
        const double bc1 = 1.0 - pow(c.beta1, (double)x->step);
        const double bc2 = 1.0 - pow(c.beta2, (double)x->step);
        a.P = c.E0; a.M = c.adam_m; a.V = c.adam_v;
        a.cnt = x->cnt; a.lam = c.decay / (float)B;
        // inside a multi-step call on replicated tables the epilogue keeps the bf16 copy of E0 current (row-sharded steps
        // update only the owned rows and convert again after the exchange)
        a.Pb = (x->e0b && x->in_loop && fused_finish) ? x->e0b : nullptr;
        a.Pq = (x->e0q && x->in_loop && fused_finish) ? x->e0q : nullptr;
        x->e0b_fresh = a.Pb != nullptr || a.Pq != nullptr;
        a.step_size = (float)(c.lr / bc1); a.bc2_sqrt = (float)sqrt(bc2);
        a.w1 = (float)(1.0 - c.beta1); a.beta2 = (float)c.beta2; a.omb2 = (float)(1.0 - c.beta2); a.eps = (float)c.eps;
        if (!first && fused_finish) {       // K >= 2: this launch also cleans the workspace and reduces the loss
            a.clear = 1; a.terms = c.terms; a.gathered = gathered; a.loss_out = loss_out;
            a.B = B; a.shard = shard; a.decay = c.decay; a.ent_coeff = c.item_pop ? c.gate_entropy_coeff : 0.f;
            a.blk = gathered ? dp_block_floats(x, shard) : 0;
        }
    }
    const int prev_dt = first ? LGCN_F32 : c.act_dtype;
    if (first && last) return launch_spmm<M_SPARSE | M_ADDG | M_ADAM>(a, c.d, prev_dt, LGCN_F32, st);
    if (first) return launch_spmm<M_SPARSE | M_ADDG>(a, c.d, prev_dt, c.act_dtype, st);
    if (last) return launch_spmm<M_ADDG | M_ADAM>(a, c.d, prev_dt, LGCN_F32, st);
    return launch_spmm<M_ADDG>(a, c.d, prev_dt, c.act_dtype, st);
}

// [DP scatter] + backward chain + Adam + finish
// gate_reduced: the MLP gradient of the global batch is the all-reduced gate_total (dense data-parallel form)
static int run_backward(lgcn_ctx *x, const int32_t *users, const int32_t *pos, const int32_t *neg, int32_t B,
                        const float *gathered, int32_t shard, int32_t world, float *loss_out, hipStream_t st, bool gate_reduced = false) {
    const lgcn_train_config &c = x->c;
    { int rc0 = graph_acquire(c.graph, st); if (rc0) return rc0; }
    SlotArgs s = slot_args(x, users, pos, neg, B, gathered, shard, world, loss_out);
    const unsigned sgrid = slot_grid(x, B);
    if (gathered) {
        s.skip_rank = x->dp_local ? x->dp_rank : -1;       // part 1 already added this rank's own rows
        DISPATCH_D(c.d, hipLaunchKernelGGL((k_scatter<D>), dim3(scatter_grid(x, B)), dim3(256), 0, st, s));
    }
    x->step += 1;
    // Horner: h_{K-1} = Gs + A Gs (sparse input); h_{k-1} = Gs + A h_k; last one feeds Adam
    for (int k = c.K; k >= 1; k--) {
        int rc = backward_layer(x, k, users, pos, neg, B, gathered, shard, loss_out, true, st);
        if (rc) return rc;
    }
    if (x->variant && c.item_pop) {          // torch.optim.Adam on the gate's MLP parameters, same step count as the tables
        GateAdamArgs ga{};
        if (gathered) {      // data parallel: the ranks' totals sit in the tails of their exchange blocks
            ga.src = (const long long *)(gathered + dp_block_tail(x, shard)); ga.n_src = world; ga.stride = dp_block_floats(x, shard) / 2;
        } else if (gate_reduced) {             // dense form: the all-reduced total
            ga.src = x->gate_total; ga.n_src = 1; ga.stride = x->gate_P;
        } else { ga.src = x->gate_partials; ga.n_src = (B + GATE_TPB - 1) / GATE_TPB; ga.stride = x->gate_P; }
        ga.P = x->gate_P;
        ga.params = c.gate_params; ga.m = c.gate_adam_m; ga.v = c.gate_adam_v; ga.grad_out = c.gate_grad;
        const double bc1 = 1.0 - pow(c.beta1, (double)x->step), bc2 = 1.0 - pow(c.beta2, (double)x->step);
        ga.step_size = (float)(c.lr / bc1); ga.bc2_sqrt = (float)sqrt(bc2);
        ga.w1 = (float)(1.0 - c.beta1); ga.beta2 = (float)c.beta2; ga.omb2 = (float)(1.0 - c.beta2); ga.eps = (float)c.eps;
        hipLaunchKernelGGL(k_gate_adam, dim3((unsigned)((x->gate_P + 255) / 256)), dim3(256), 0, st, ga);
    }
    if (c.K >= 2) { x->flip ^= 1; return 0; }       // next step flags rows in the other bitmap
    DISPATCH_D(c.d, hipLaunchKernelGGL((k_finish<D>), dim3(sgrid), dim3(256), 0, st, s));
    if (x->variant && c.i2i) {
        // K = 1 keeps ONE bitmap (k_finish clears the batch rows' words); the smoothing's backward flagged EVERY item row in it
        // and zeroed nothing of G64 beyond the batch rows, which k_finish has just done: clear the whole bitmap
        HIP_OK(hipMemsetAsync(c.bitmap + x->flip * x->bm_words, 0, sizeof(uint32_t) * (size_t)x->bm_words, st));
    }
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
    if (x->variant) rc = run_variant_loss(x, users, pos, neg, B, 0, B, B, true, false, st);
    else rc = run_bpr(x, users, pos, neg, B, 0, B, B, true, false, st);
    if (rc) return rc;
    if ((rc = run_backward(x, users, pos, neg, B, nullptr, B, 1, loss_out, st))) return rc;
    HIP_OK(hipGetLastError());
    return 0;
}

extern "C" int lgcn_train_epoch(lgcn_ctx *x, const int32_t *users, const int32_t *pos, const int32_t *neg,
                                int64_t T, int32_t B, float *loss_out, void *stream) {
    if (T <= 0) return 0;
    if (!x) { lgcn_set_error("train epoch: null context"); return 3; }
    LoopScope scope(x);
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
    // with dp_local the rank's own rows go into G64 here (atomics) AND into the exchange block; part 2 scatters the others'
    if (x->variant) rc = run_variant_loss(x, users, pos, neg, B_global, b_off, B_local, shard, x->dp_local, true, st);
    else rc = run_bpr(x, users, pos, neg, B_global, b_off, B_local, shard, x->dp_local, true, st);
    if (rc) return rc;
    x->dp_rank = rank;
    HIP_OK(hipGetLastError());
    return 0;
}

extern "C" int lgcn_train_step_dp_dense_part1(lgcn_ctx *x, const int32_t *users, const int32_t *pos, const int32_t *neg,
                                              int32_t B_global, int32_t world, int32_t rank, void *stream) {
    int rc = check_batch(x, users, pos, neg, B_global);
    if (rc) return rc;
    if (world < 1 || rank < 0 || rank >= world) { lgcn_set_error("dp step: bad world/rank"); return 3; }
    const int32_t shard = (B_global + world - 1) / world;
    const int32_t b_off = rank * shard;
    int32_t B_local = B_global - b_off;
    if (B_local > shard) B_local = shard;
    if (B_local < 0) B_local = 0;
    hipStream_t st = (hipStream_t)stream;
    const bool gate = x->variant && x->c.item_pop;
    // this rank owns positions [b_off, b_off+B_local) of the global loss-term arrays (loss | reg | with the gate: entropy);
    // the rest must be zero
    HIP_OK(hipMemsetAsync(x->c.terms, 0, sizeof(float) * (gate ? 3 : 2) * (size_t)B_global, st));
    if ((rc = run_forward(x, st))) return rc;
    if (x->variant) {
        // optional branches: the shard's rows go into this rank's G64 by atomics like the default model's; the MLP
        // parameter gradients of the shard are summed (fixed point: any order) into gate_total, one more all-reduce
        if ((rc = run_variant_loss(x, users, pos, neg, B_global, b_off, B_local, shard, true, false, st))) return rc;
        if (gate) {
            if (B_local > 0)
                hipLaunchKernelGGL(k_gate_rank_total, dim3((unsigned)((x->gate_P + 255) / 256)), dim3(256), 0, st,
                                   (const long long *)x->gate_partials, (int32_t)((B_local + GATE_TPB - 1) / GATE_TPB), x->gate_P, x->gate_total);
            else HIP_OK(hipMemsetAsync(x->gate_total, 0, sizeof(long long) * (size_t)x->gate_P, st));
        }
    } else if ((rc = run_bpr(x, users, pos, neg, B_global, b_off, B_local, shard, true, false, st, false))) return rc;
    SlotArgs s{};
    s.users = users; s.pos = pos; s.neg = neg; s.B = B_global; s.n_users = x->c.n_users; s.N = x->N;
    s.bitmap = x->c.bitmap + x->flip * x->bm_words; s.cnt = x->cnt;
    s.item_bitmap = (x->variant && x->c.i2i) ? x->item_bitmap : nullptr;      // the smoothing's backward reads the items of the GLOBAL batch
    hipLaunchKernelGGL(k_flag_rows, dim3((3 * B_global + 255) / 256), dim3(256), 0, st, s);
    HIP_OK(hipGetLastError());
    return 0;
}

extern "C" int lgcn_ctx_gate_total(const lgcn_ctx *x, void **buf, int32_t *count) {
    if (!x || !buf || !count) { lgcn_set_error("lgcn_ctx_gate_total: null argument"); return 3; }
    *buf = x->gate_total; *count = x->gate_total ? x->gate_P : 0;
    return 0;
}

extern "C" int lgcn_train_step_dp_part2(lgcn_ctx *x, const int32_t *users, const int32_t *pos, const int32_t *neg,
                                        int32_t B_global, int32_t world, const float *gathered, float *loss_out,
                                        void *stream) {
    int rc = check_batch(x, users, pos, neg, B_global);
    if (rc) return rc;
    if (!loss_out || world < 1) { lgcn_set_error("dp step part 2: invalid argument"); return 3; }
    const int32_t shard = (B_global + world - 1) / world;
    // gathered == NULL is the dense form: G64, the terms and (popularity gate) gate_total hold the reduced sums of the global batch
    if ((rc = run_backward(x, users, pos, neg, B_global, gathered, shard, world, loss_out, (hipStream_t)stream, gathered == nullptr))) return rc;
    HIP_OK(hipGetLastError());
    return 0;
}

// ---------------------------------------------------------------------------------
// Column-sharded data parallelism: rank r holds columns [r d/W, (r+1) d/W) of E0, of the Adam state and of every activation --
// its context is an ordinary context of width d / W over the same graph.  Propagation, the gradient scatter, the backward
// chain and Adam are independent per column; the one thing that is not is the score of a triplet, a dot product over ALL
// columns.  So the step is: part 1 (forward + slot rows + PARTIAL scores / reg terms of this rank's columns), ONE all-reduce
// of 3 B floats, part 2 (loss terms, gradient rows of this rank's columns, backward, Adam).  Per-rank SpMM work falls with
// the world size (rows of d / W elements), which batch sharding cannot offer.  Not bitwise equal to the single-GPU step: the
// dot products are summed in another order (W partial sums); everything else is.
// ---------------------------------------------------------------------------------
extern "C" int lgcn_train_step_cols_part1(lgcn_ctx *x, const int32_t *users, const int32_t *pos, const int32_t *neg,
                                          int32_t B, float **partials, void *stream) {
    int rc = check_batch(x, users, pos, neg, B);
    if (rc) return rc;
    if (x->variant) { lgcn_set_error("column-sharded step: the popularity gate / item-item smoothing mix columns (MLPs over the row): not supported"); return 3; }
    if (!x->c.contrib) { lgcn_set_error("column-sharded step: cfg.contrib (3*max_batch*d floats: the batch's slot rows) missing"); return 3; }
    hipStream_t st = (hipStream_t)stream;
    if ((rc = run_forward(x, st))) return rc;
    if ((rc = run_bpr(x, users, pos, neg, B, 0, B, B, false, false, st, false, 1))) return rc;
    if (partials) *partials = x->colsum;
    HIP_OK(hipGetLastError());
    return 0;
}
extern "C" int lgcn_train_step_cols_part2(lgcn_ctx *x, const int32_t *users, const int32_t *pos, const int32_t *neg,
                                          int32_t B, float *loss_out, void *stream) {
    int rc = check_batch(x, users, pos, neg, B);
    if (rc) return rc;
    if (!loss_out) { lgcn_set_error("column-sharded step part 2: loss_out is null"); return 3; }
    hipStream_t st = (hipStream_t)stream;
    if ((rc = run_bpr(x, users, pos, neg, B, 0, B, B, true, false, st, true, 2))) return rc;
    if ((rc = run_backward(x, users, pos, neg, B, nullptr, B, 1, loss_out, st))) return rc;
    HIP_OK(hipGetLastError());
    return 0;
}

// ---------------------------------------------------------------------------------
// Row-sharded propagation (SURVEY 8e "beyond the contract"; the analogue of the reference's A_split
// row folds, dataloader.py:192-201): the context's graph plan holds only the rows this rank owns; a
// phase computes those rows of one layer, and the owners' rows are exchanged before the next phase.
// ---------------------------------------------------------------------------------
extern "C" int lgcn_rs_phase(lgcn_ctx *x, int32_t phase, int32_t k, const int32_t *users, const int32_t *pos, const int32_t *neg,
                             int32_t B_global, int32_t world, int32_t rank, const float *gathered, float *loss_out, void *stream) {
    int rc = check_batch(x, users, pos, neg, B_global);
    if (rc) return rc;
    if (x->variant) { lgcn_set_error("lgcn_rs_phase: the popularity gate / item-item smoothing run on one GPU only"); return 3; }
    const lgcn_train_config &c = x->c;
    hipStream_t st = (hipStream_t)stream;
    if (world < 1 || rank < 0 || rank >= world) { lgcn_set_error("lgcn_rs_phase: bad world/rank"); return 3; }
    const int32_t shard = (B_global + world - 1) / world;
    if ((rc = graph_acquire(c.graph, st))) return rc;
    switch (phase) {
    case LGCN_RS_FWD: {                               // X_k[owned] = (A X_{k-1})[owned], k = 1..K-1
        if (k < 1 || k > x->fwd_layers) { lgcn_set_error("lgcn_rs_phase: forward layer out of range"); return 3; }
        SpmmArgs a = base_spmm(x);
        // bf16 / fp8 activation storage: layer 1 gathers bf16(E0) / fp8(E0), converted after every exchange of the owners' Adam rows
        const bool shadow = k == 1 && (x->e0b != nullptr || x->e0q != nullptr);
        if (shadow) {
            if (x->e0b) launch_to_bf16(c.E0, x->e0b, x->N * c.d, st);
            else if ((rc = launch_to_fp8(c.E0, x->e0q, x->N, c.d, st))) return rc;
            x->e0b_fresh = false;
        }
        a.X = k == 1 ? (shadow ? (x->e0b ? (const void *)x->e0b : (const void *)x->e0q) : (const void *)c.E0) : x->act[k - 1]; a.Y = x->act[k];
        rc = launch_spmm<0>(a, c.d, k == 1 && !shadow ? LGCN_F32 : c.act_dtype, c.act_dtype, st);
        break;
    }
    case LGCN_RS_BPR: {                               // this rank's batch shard -> cfg.contrib
        if (!c.contrib) { lgcn_set_error("lgcn_rs_phase: cfg.contrib exchange buffer missing"); return 3; }
        const int32_t b_off = rank * shard;
        int32_t B_local = B_global - b_off;
        if (B_local > shard) B_local = shard;
        if (B_local < 0) B_local = 0;
        rc = run_bpr(x, users, pos, neg, B_global, b_off, B_local, shard, false, true, st);
        break;
    }
    case LGCN_RS_SCATTER: {                           // all ranks' gradient rows -> G64 + row flags
        if (!gathered) { lgcn_set_error("lgcn_rs_phase: gathered blocks missing"); return 3; }
        SlotArgs s = slot_args(x, users, pos, neg, B_global, gathered, shard, world, loss_out);
        DISPATCH_D(c.d, hipLaunchKernelGGL((k_scatter<D>), dim3(scatter_grid(x, B_global)), dim3(256), 0, st, s));
        break;
    }
    case LGCN_RS_BWD:                                 // h_{k-1}[owned] = Gs + (A h_k)[owned], k = K..1; k = 1: Adam on the owned rows
        if (k < 1 || k > c.K) { lgcn_set_error("lgcn_rs_phase: backward layer out of range"); return 3; }
        if (k == c.K) x->step += 1;
        rc = backward_layer(x, k, users, pos, neg, B_global, gathered, shard, loss_out, false, st);
        break;
    case LGCN_RS_FINISH: {                            // zero the batch rows of G64 + their flags, reduce the loss
        if (!loss_out) { lgcn_set_error("lgcn_rs_phase: loss_out is null"); return 3; }
        SlotArgs s = slot_args(x, users, pos, neg, B_global, gathered, shard, world, loss_out);
        DISPATCH_D(c.d, hipLaunchKernelGGL((k_finish<D>), dim3(slot_grid(x, B_global)), dim3(256), 0, st, s));
        break;
    }
    default: lgcn_set_error("lgcn_rs_phase: unknown phase"); return 3;
    }
    if (rc) return rc;
    HIP_OK(hipGetLastError());
    return 0;
}

// what a phase wrote and the owners must exchange: buffer + element type
extern "C" int lgcn_rs_buffer(const lgcn_ctx *x, int32_t phase, int32_t k, void **buf, int32_t *dtype) {
    if (!x || !buf || !dtype) { lgcn_set_error("lgcn_rs_buffer: null argument"); return 3; }
    const lgcn_train_config &c = x->c;
    if (phase == LGCN_RS_FWD && k >= 1 && k <= x->fwd_layers) { *buf = x->act[k]; *dtype = c.act_dtype; return 0; }
    if (phase == LGCN_RS_BWD && k > 1 && k <= c.K) { *buf = bwd_buffer(x, k); *dtype = c.act_dtype; return 0; }
    if (phase == LGCN_RS_BWD && k == 1) { *buf = c.E0; *dtype = LGCN_F32; return 0; }
    lgcn_set_error("lgcn_rs_buffer: this phase exchanges nothing");
    return 3;
}

// every owner broadcasts its two row ranges (users, items) of `buf` in place; one RCCL group
// (an fp8 table: the rows are d bytes each and the owners' row scales, fp32 behind the rows, travel with them)
static int rs_exchange(const RcclApi *api, lgcn_dp *dp, void *buf, int dtype, int d, int64_t n_rows, const int64_t *ranges, hipStream_t st) {
    const size_t es = dtype == LGCN_BF16 ? 2 : dtype == LGCN_FP8 ? 1 : 4;
    const ncclDataType_t nt = dtype == LGCN_BF16 ? ncclBfloat16 : dtype == LGCN_FP8 ? ncclUint8 : ncclFloat32;
    float *scales = dtype == LGCN_FP8 ? (float *)((char *)buf + (size_t)n_rows * d) : nullptr;
    ncclResult_t r = api->GroupStart();
    for (int q = 0; q < dp->world && r == ncclSuccess; q++)
        for (int part = 0; part < 2 && r == ncclSuccess; part++) {
            const int64_t lo = ranges[4 * q + 2 * part], hi = ranges[4 * q + 2 * part + 1];
            if (hi <= lo) continue;
            char *p = (char *)buf + (size_t)lo * d * es;
            r = api->Broadcast(p, p, (size_t)(hi - lo) * d, nt, q, dp->comm, st);
            if (scales && r == ncclSuccess) r = api->Broadcast(scales + lo, scales + lo, (size_t)(hi - lo), ncclFloat32, q, dp->comm, st);
        }
    if (r == ncclSuccess) r = api->GroupEnd(); else (void)api->GroupEnd();
    if (r != ncclSuccess) { lgcn_set_error("RCCL broadcast of owned rows failed"); return 11; }
    return 0;
}

// A whole data-parallel epoch from ONE host call: per global batch, part 1 -> RCCL collective on the
// SAME stream (no host synchronisation, no Python between the kernels and the collective) -> part 2.
static int train_epoch_dp_impl(lgcn_ctx *x, lgcn_dp *dp, const int32_t *users, const int32_t *pos, const int32_t *neg,
                               int64_t T, int32_t B_global, int32_t reduce, const int64_t *row_ranges, float *gathered,
                               float *loss_out, void *stream) {
    if (!x || !dp || !users || !pos || !neg || !loss_out) { lgcn_set_error("lgcn_train_epoch_dp: null argument"); return 3; }
    if (reduce != LGCN_DP_ROWS && reduce != LGCN_DP_DENSE && reduce != LGCN_DP_ROW_SHARDED && reduce != LGCN_DP_COLS) { lgcn_set_error("lgcn_train_epoch_dp: unknown reduce mode"); return 3; }
    if (reduce == LGCN_DP_ROW_SHARDED && !row_ranges) { lgcn_set_error("lgcn_train_epoch_dp: row_ranges missing"); return 3; }
    if (reduce != LGCN_DP_DENSE && reduce != LGCN_DP_COLS && !gathered) { lgcn_set_error("lgcn_train_epoch_dp: gathered workspace missing"); return 3; }
    if (B_global <= 0 || B_global > x->c.max_batch) { lgcn_set_error("lgcn_train_epoch_dp: batch size out of range"); return 3; }
    const RcclApi *api = dp->api;             // RCCL, or the in-process loopback of the tests
    if (!api) { lgcn_set_error("lgcn_train_epoch_dp: communicator without collectives"); return 12; }
    hipStream_t st = (hipStream_t)stream;
    const int world = dp->world, rank = dp->rank;
    LoopScope scope(x);
    struct LocalScope {         // batch-sharded rows mode: every rank adds its own rows itself, k_scatter the other ranks'
        lgcn_ctx *x; bool was;
        LocalScope(lgcn_ctx *x_, bool on) : x(x_), was(x_->dp_local) { x->dp_local = on; x->dp_rank = -1; }
        ~LocalScope() { x->dp_local = was; x->dp_rank = -1; }
    } local_scope(x, reduce == LGCN_DP_ROWS);
    int64_t i = 0;
    for (int64_t t = 0; t < T; t += B_global, i++) {
        const int32_t b = (int32_t)((T - t) < B_global ? (T - t) : B_global);
        int rc;
        ncclResult_t r;
        if (reduce == LGCN_DP_COLS) {
            if ((rc = lgcn_train_step_cols_part1(x, users + t, pos + t, neg + t, b, nullptr, stream))) return rc;
            r = api->AllReduce(x->colsum, x->colsum, (size_t)3 * b, ncclFloat32, ncclSum, dp->comm, st);
            if (r != ncclSuccess) { lgcn_set_error("ncclAllReduce failed"); return 11; }
            if ((rc = lgcn_train_step_cols_part2(x, users + t, pos + t, neg + t, b, loss_out + 3 * i, stream))) return rc;
        } else if (reduce == LGCN_DP_ROW_SHARDED) {
            const int32_t K = x->c.K, d = x->c.d;
            const int32_t *u = users + t, *p = pos + t, *n = neg + t;
            void *buf; int32_t dt;
            for (int k = 1; k <= x->fwd_layers; k++) {
                if ((rc = lgcn_rs_phase(x, LGCN_RS_FWD, k, u, p, n, b, world, rank, nullptr, nullptr, stream))) return rc;
                if ((rc = lgcn_rs_buffer(x, LGCN_RS_FWD, k, &buf, &dt))) return rc;
                if ((rc = rs_exchange(api, dp, buf, dt, d, x->N, row_ranges, st))) return rc;
            }
            if ((rc = lgcn_rs_phase(x, LGCN_RS_BPR, 0, u, p, n, b, world, rank, nullptr, nullptr, stream))) return rc;
            const int64_t S = (b + world - 1) / world, blk = 3 * S * d + 2 * S;
            r = api->AllGather(x->c.contrib, gathered, (size_t)blk, ncclFloat32, dp->comm, st);
            if (r != ncclSuccess) { lgcn_set_error("ncclAllGather failed"); return 11; }
            if ((rc = lgcn_rs_phase(x, LGCN_RS_SCATTER, 0, u, p, n, b, world, rank, gathered, nullptr, stream))) return rc;
            for (int k = K; k >= 1; k--) {
                if ((rc = lgcn_rs_phase(x, LGCN_RS_BWD, k, u, p, n, b, world, rank, gathered, nullptr, stream))) return rc;
                if ((rc = lgcn_rs_buffer(x, LGCN_RS_BWD, k, &buf, &dt))) return rc;
                if ((rc = rs_exchange(api, dp, buf, dt, d, x->N, row_ranges, st))) return rc;
            }
            if ((rc = lgcn_rs_phase(x, LGCN_RS_FINISH, 0, u, p, n, b, world, rank, gathered, loss_out + 3 * i, stream))) return rc;
        } else if (reduce == LGCN_DP_ROWS) {
            if ((rc = lgcn_train_step_dp_part1(x, users + t, pos + t, neg + t, b, world, rank, stream))) return rc;
            const int64_t S = (b + world - 1) / world, blk = dp_block_floats(x, S);
            r = api->AllGather(x->c.contrib, gathered, (size_t)blk, ncclFloat32, dp->comm, st);
            if (r != ncclSuccess) { lgcn_set_error("ncclAllGather failed"); return 11; }
            if ((rc = lgcn_train_step_dp_part2(x, users + t, pos + t, neg + t, b, world, gathered, loss_out + 3 * i, stream))) return rc;
        } else {
            if ((rc = lgcn_train_step_dp_dense_part1(x, users + t, pos + t, neg + t, b, world, rank, stream))) return rc;
            r = api->AllReduce(x->c.G64, x->c.G64, (size_t)x->N * x->c.d, ncclInt64, ncclSum, dp->comm, st);
            const bool gate = x->variant && x->c.item_pop;
            if (r == ncclSuccess) r = api->AllReduce(x->c.terms, x->c.terms, (size_t)(gate ? 3 : 2) * b, ncclFloat32, ncclSum, dp->comm, st);
            if (r == ncclSuccess && gate) r = api->AllReduce(x->gate_total, x->gate_total, (size_t)x->gate_P, ncclInt64, ncclSum, dp->comm, st);
            if (r != ncclSuccess) { lgcn_set_error("ncclAllReduce failed"); return 11; }
            if ((rc = lgcn_train_step_dp_part2(x, users + t, pos + t, neg + t, b, world, nullptr, loss_out + 3 * i, stream))) return rc;
        }
    }
    return 0;
}

extern "C" int lgcn_train_epoch_dp(lgcn_ctx *x, lgcn_dp *dp, const int32_t *users, const int32_t *pos, const int32_t *neg,
                                   int64_t T, int32_t B_global, int32_t reduce, const int64_t *row_ranges, float *gathered,
                                   float *loss_out, void *stream) {
    const int rc = train_epoch_dp_impl(x, dp, users, pos, neg, T, B_global, reduce, row_ranges, gathered, loss_out, stream);
    // a rank that leaves the loop early never reaches its next collective: release the loopback ranks waiting for it
    // (RCCL has its own abort / timeout machinery)
    if (rc && dp && dp->loopback) lgcn_dp_loopback_abort(dp);
    return rc;
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
