// Experiment (not product code): upper bound of the neighbour-row gather for a CSR graph.
// Each wave takes 64 consecutive non-zeros (one coalesced index tile), gathers the 64 rows
// (LPR lanes x 16/8 bytes per row, NPW rows per instruction, U in flight) and sums them all,
// ignoring row boundaries: no per-row reduce / store / indptr logic.  Tells how far the real
// SpMM (k_spmm) is from what the memory system gives for THIS access pattern.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;

template <int D, typename T, int U>
__global__ void __launch_bounds__(256) k_tile_gather(const int32_t *indices, const float *vals, int64_t nnz,
                                                     const T *X, float *out) {
    constexpr int LPR = D / 4, NPW = 64 / LPR;
    __shared__ int2 stage[4][64];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int g = lane / LPR, l = lane % LPR;
    const int64_t tile = (int64_t)blockIdx.x * 4 + wid;
    const int64_t base = tile * 64;
    if (base >= nnz) return;
    const int n = (int)min((int64_t)64, nnz - base);
    if (lane < n) stage[wid][lane] = make_int2(indices[base + lane], __float_as_int(vals[base + lane]));
    __builtin_amdgcn_wave_barrier();
    f32x4 acc = {0, 0, 0, 0};
    for (int j = g; j < n; j += NPW * U) {
        int2 cv[U];
        typedef typename std::conditional<sizeof(T) == 4, f32x4, bf16x4>::type raw_t;
        raw_t x[U];
#pragma unroll
        for (int u = 0; u < U; u++) cv[u] = stage[wid][min(j + u * NPW, 63)];
#pragma unroll
        for (int u = 0; u < U; u++) {
            if (j + u * NPW < n) x[u] = *reinterpret_cast<const raw_t *>(X + (int64_t)cv[u].x * D + l * 4);
            else { x[u] = 0; cv[u].y = 0; }
        }
#pragma unroll
        for (int u = 0; u < U; u++) acc += __int_as_float(cv[u].y) * __builtin_convertvector(x[u], f32x4);
    }
#pragma unroll
    for (int off = LPR; off < 64; off <<= 1) {
        acc.x += __shfl_xor(acc.x, off); acc.y += __shfl_xor(acc.y, off);
        acc.z += __shfl_xor(acc.z, off); acc.w += __shfl_xor(acc.w, off);
    }
    if (lane < LPR) *reinterpret_cast<f32x4 *>(out + tile * D + l * 4) = acc;
}

extern "C" int exp_tile_gather(const int32_t *indices, const float *vals, int64_t nnz, const void *X, int bf16,
                               int U, float *out, void *stream) {
    const unsigned grid = (unsigned)(((nnz + 63) / 64 + 3) / 4);
    hipStream_t st = (hipStream_t)stream;
#define GO(T, UU) hipLaunchKernelGGL((k_tile_gather<64, T, UU>), dim3(grid), dim3(256), 0, st, indices, vals, nnz, (const T *)X, out)
    if (!bf16) { if (U == 4) GO(float, 4); else if (U == 8) GO(float, 8); else GO(float, 16); }
    else { if (U == 4) GO(__bf16, 4); else if (U == 8) GO(__bf16, 8); else GO(__bf16, 16); }
    return (int)hipGetLastError();
}
