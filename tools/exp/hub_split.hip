// EXPERIMENT (not part of the product library; built and loaded only by tools/hub_split.py).
// Y_hub = A_hub * X_hub: the most-gathered columns of the propagation matrix taken out of the main SpMM.  One persistent
// workgroup per CU (16 waves) copies the H hub rows of X into LDS once and walks its share of the rows of A_hub; a lane group
// of d/4 lanes owns a row, its entries (LDS slot, weight) come from a CSR stream of their own.
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef __attribute__((ext_vector_type(4))) float f32x4;

template <int D>
__global__ void __launch_bounds__(1024) k_hub_spmm(const float *X, const int32_t *hub_rows, int H,
                                                   const int32_t *indptr, const int2 *ent /* (slot, weight bits) */, int64_t n_rows, float *Y) {
    extern __shared__ __attribute__((aligned(16))) float xh[];          // [H][D]
    constexpr int LPR = D / 4, RPW = 64 / LPR;                            // lanes per row, rows per wave
    for (int p = threadIdx.x; p < H * LPR; p += 1024) {
        const int r = p / LPR, c = p % LPR;
        *reinterpret_cast<f32x4 *>(&xh[r * D + c * 4]) = *reinterpret_cast<const f32x4 *>(X + (int64_t)hub_rows[r] * D + c * 4);
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, g = lane / LPR, l = lane % LPR;
    const int64_t stride = (int64_t)gridDim.x * 16 * RPW;
    for (int64_t row = ((int64_t)blockIdx.x * 16 + wave) * RPW + g; row < n_rows; row += stride) {
        const int s = indptr[row], e = indptr[row + 1];
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        int i = s;
        for (; i + 4 <= e; i += 4) {                                       // four entries in flight
            const int2 a0 = ent[i], a1 = ent[i + 1], a2 = ent[i + 2], a3 = ent[i + 3];
            const f32x4 x0 = *reinterpret_cast<const f32x4 *>(&xh[a0.x * D + l * 4]), x1 = *reinterpret_cast<const f32x4 *>(&xh[a1.x * D + l * 4]);
            const f32x4 x2 = *reinterpret_cast<const f32x4 *>(&xh[a2.x * D + l * 4]), x3 = *reinterpret_cast<const f32x4 *>(&xh[a3.x * D + l * 4]);
            acc += __int_as_float(a0.y) * x0; acc += __int_as_float(a1.y) * x1; acc += __int_as_float(a2.y) * x2; acc += __int_as_float(a3.y) * x3;
        }
        for (; i < e; i++) {
            const int2 a0 = ent[i];
            acc += __int_as_float(a0.y) * *reinterpret_cast<const f32x4 *>(&xh[a0.x * D + l * 4]);
        }
        *reinterpret_cast<f32x4 *>(Y + row * D + l * 4) = acc;
    }
}

extern "C" int hub_spmm(const float *X, const int32_t *hub_rows, int H, const int32_t *indptr, const void *ent, int64_t n_rows, int d,
                        float *Y, int blocks, void *stream) {
    const size_t lds = (size_t)H * d * 4;
    hipStream_t st = (hipStream_t)stream;
#define GO(DD) do { if (hipFuncSetAttribute((const void *)k_hub_spmm<DD>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return 2; \
                    hipLaunchKernelGGL((k_hub_spmm<DD>), dim3(blocks), dim3(1024), lds, st, X, hub_rows, H, indptr, (const int2 *)ent, n_rows, Y); } while (0)
    if (d == 64) GO(64); else if (d == 128) GO(128); else return 3;
#undef GO
    return hipGetLastError() == hipSuccess ? 0 : 1;
}
