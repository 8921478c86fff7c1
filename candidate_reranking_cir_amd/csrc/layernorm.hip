// cir_layernorm / cir_embed_layernorm: row LayerNorm with fp32 statistics, one wave per row.
//
// Bound: HBM.  Algorithmic bytes per row: cols * (4 [x] + 4 [residual, if any] + 4 [y32] + 2 [y16]).
// A row of 768 fp32 is 3 KiB = three 16-byte vectors per lane, fully coalesced; the reduction is a
// 6-step wave shuffle (no LDS, no barrier).  Mean and variance use the two-pass form on register
// data (the row is read once).
#include "common.hpp"

namespace cir {

template <typename T, int NCH>
__device__ __forceinline__ void ln_row(float4 (&v)[NCH], int lane, int cols, const float* gamma, const float* beta,
                                       float eps, float* y32, T* y16) {
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < NCH; ++c)
        if ((lane + c * 64) * 4 < cols) s += (v[c].x + v[c].y) + (v[c].z + v[c].w);
    const float mean = wave_sum(s) / (float)cols;
    float q = 0.f;
#pragma unroll
    for (int c = 0; c < NCH; ++c)
        if ((lane + c * 64) * 4 < cols) {
            const float dx = v[c].x - mean, dy = v[c].y - mean, dz = v[c].z - mean, dw = v[c].w - mean;
            q += (dx * dx + dy * dy) + (dz * dz + dw * dw);
        }
    const float rstd = rsqrtf(wave_sum(q) / (float)cols + eps);
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        const int col = (lane + c * 64) * 4;
        if (col < cols) {
            const float4 g4 = *reinterpret_cast<const float4*>(gamma + col);
            const float4 b4 = *reinterpret_cast<const float4*>(beta + col);
            float4 o;
            o.x = (v[c].x - mean) * rstd * g4.x + b4.x;
            o.y = (v[c].y - mean) * rstd * g4.y + b4.y;
            o.z = (v[c].z - mean) * rstd * g4.z + b4.z;
            o.w = (v[c].w - mean) * rstd * g4.w + b4.w;
            if (y32) *reinterpret_cast<float4*>(y32 + col) = o;
            if (y16) {
                u32x2 p;
                p.x = pack2<T>(o.x, o.y);
                p.y = pack2<T>(o.z, o.w);
                *reinterpret_cast<u32x2*>(y16 + col) = p;
            }
        }
    }
}

template <typename T, int NCH>
__global__ __launch_bounds__(256) void layernorm_kernel(const float* x, int64_t sX, const float* res, int64_t sR,
                                                        const float* gamma, const float* beta, int64_t sG, float* y32,
                                                        T* y16, int64_t sY, int64_t rows, int cols, float eps) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int b = blockIdx.y;
    const float* xr = x + b * sX + row * cols;
    const float* rr = res ? res + b * sR + row * cols : nullptr;
    float4 v[NCH];
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        const int col = (lane + c * 64) * 4;
        if (col < cols) {
            v[c] = *reinterpret_cast<const float4*>(xr + col);
            if (rr) {
                const float4 r4 = *reinterpret_cast<const float4*>(rr + col);
                v[c].x += r4.x; v[c].y += r4.y; v[c].z += r4.z; v[c].w += r4.w;
            }
        } else {
            v[c] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
    }
    ln_row<T, NCH>(v, lane, cols, gamma + b * sG, beta + b * sG, eps, y32 ? y32 + b * sY + row * cols : nullptr,
                   y16 ? y16 + b * sY + row * cols : nullptr);
}

template <typename T, int NCH>
__global__ __launch_bounds__(256) void embed_ln_kernel(const int64_t* ids, const float* word, const float* pos,
                                                       const float* gamma, const float* beta, float* y32, T* y16,
                                                       int64_t rows, int L, int cols, int vocab, float eps) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    int64_t id = ids[row];
    id = id < 0 ? 0 : (id >= vocab ? vocab - 1 : id);  // never read outside the table
    const float* wr = word + id * cols;
    const float* pr = pos + (row % L) * cols;
    float4 v[NCH];
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        const int col = (lane + c * 64) * 4;
        if (col < cols) {
            const float4 w4 = *reinterpret_cast<const float4*>(wr + col);
            const float4 p4 = *reinterpret_cast<const float4*>(pr + col);
            v[c] = make_float4(w4.x + p4.x, w4.y + p4.y, w4.z + p4.z, w4.w + p4.w);
        } else {
            v[c] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
    }
    ln_row<T, NCH>(v, lane, cols, gamma, beta, eps, y32 ? y32 + row * cols : nullptr, y16 ? y16 + row * cols : nullptr);
}

template <typename T>
static int launch_ln(const float* x, int64_t sX, const float* res, int64_t sR, const float* gamma, const float* beta,
                     int64_t sG, float* y32, void* y16, int64_t sY, int64_t rows, int cols, int batch, float eps,
                     hipStream_t s) {
    dim3 grid((unsigned)((rows + 3) / 4), (unsigned)batch), block(256);
    const int nch = (cols + 255) / 256;
    T* y = reinterpret_cast<T*>(y16);
    switch (nch) {
        case 1: hipLaunchKernelGGL((layernorm_kernel<T, 1>), grid, block, 0, s, x, sX, res, sR, gamma, beta, sG, y32, y, sY, rows, cols, eps); break;
        case 2: hipLaunchKernelGGL((layernorm_kernel<T, 2>), grid, block, 0, s, x, sX, res, sR, gamma, beta, sG, y32, y, sY, rows, cols, eps); break;
        case 3: hipLaunchKernelGGL((layernorm_kernel<T, 3>), grid, block, 0, s, x, sX, res, sR, gamma, beta, sG, y32, y, sY, rows, cols, eps); break;
        default: hipLaunchKernelGGL((layernorm_kernel<T, 4>), grid, block, 0, s, x, sX, res, sR, gamma, beta, sG, y32, y, sY, rows, cols, eps); break;
    }
    CIR_LAUNCH_RESULT();
}

template <typename T>
static int launch_embed(const int64_t* ids, const float* word, const float* pos, const float* gamma, const float* beta,
                        float* y32, void* y16, int64_t rows, int L, int cols, int vocab, float eps, hipStream_t s) {
    dim3 grid((unsigned)((rows + 3) / 4)), block(256);
    const int nch = (cols + 255) / 256;
    T* y = reinterpret_cast<T*>(y16);
    switch (nch) {
        case 1: hipLaunchKernelGGL((embed_ln_kernel<T, 1>), grid, block, 0, s, ids, word, pos, gamma, beta, y32, y, rows, L, cols, vocab, eps); break;
        case 2: hipLaunchKernelGGL((embed_ln_kernel<T, 2>), grid, block, 0, s, ids, word, pos, gamma, beta, y32, y, rows, L, cols, vocab, eps); break;
        case 3: hipLaunchKernelGGL((embed_ln_kernel<T, 3>), grid, block, 0, s, ids, word, pos, gamma, beta, y32, y, rows, L, cols, vocab, eps); break;
        default: hipLaunchKernelGGL((embed_ln_kernel<T, 4>), grid, block, 0, s, ids, word, pos, gamma, beta, y32, y, rows, L, cols, vocab, eps); break;
    }
    CIR_LAUNCH_RESULT();
}

}  // namespace cir

extern "C" int cir_layernorm(const float* x, int64_t strideX, const float* residual, int64_t strideR, const float* gamma,
                             const float* beta, int64_t strideG, float* y32, void* y16, int64_t strideY, int64_t rows,
                             int cols, int batch, float eps, int dtype16, void* stream) {
    CIR_CHECK_PTR(x); CIR_CHECK_PTR(gamma); CIR_CHECK_PTR(beta);
    if (!y32 && !y16) return CIR_EINVAL;
    if (rows <= 0 || cols <= 0 || batch <= 0) return CIR_EINVAL;
    if (cols % 4 != 0 || cols > 1024) return CIR_ESHAPE;
    if (dtype16 != CIR_BF16 && dtype16 != CIR_F16) return CIR_EDTYPE;
    if (!cir_aligned16(x) || !cir_aligned16(gamma) || !cir_aligned16(beta) || (residual && !cir_aligned16(residual)) ||
        (y32 && !cir_aligned16(y32)) || (y16 && (reinterpret_cast<uintptr_t>(y16) & 7)) || strideX % 4 || strideR % 4 ||
        strideG % 4 || strideY % 4)
        return CIR_EALIGN;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (dtype16 == CIR_BF16)
        return cir::launch_ln<__bf16>(x, strideX, residual, strideR, gamma, beta, strideG, y32, y16, strideY, rows, cols, batch, eps, s);
    return cir::launch_ln<_Float16>(x, strideX, residual, strideR, gamma, beta, strideG, y32, y16, strideY, rows, cols, batch, eps, s);
}

extern "C" int cir_embed_layernorm(const int64_t* ids, const float* word, const float* pos, const float* gamma,
                                   const float* beta, float* y32, void* y16, int64_t rows, int L, int cols, int vocab,
                                   float eps, int dtype16, void* stream) {
    CIR_CHECK_PTR(ids); CIR_CHECK_PTR(word); CIR_CHECK_PTR(pos); CIR_CHECK_PTR(gamma); CIR_CHECK_PTR(beta);
    if (!y32 && !y16) return CIR_EINVAL;
    if (rows <= 0 || L <= 0 || cols <= 0 || vocab <= 0) return CIR_EINVAL;
    if (cols % 4 != 0 || cols > 1024) return CIR_ESHAPE;
    if (dtype16 != CIR_BF16 && dtype16 != CIR_F16) return CIR_EDTYPE;
    if (!cir_aligned16(word) || !cir_aligned16(pos) || !cir_aligned16(gamma) || !cir_aligned16(beta) ||
        (y32 && !cir_aligned16(y32)) || (y16 && (reinterpret_cast<uintptr_t>(y16) & 7)))
        return CIR_EALIGN;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (dtype16 == CIR_BF16) return cir::launch_embed<__bf16>(ids, word, pos, gamma, beta, y32, y16, rows, L, cols, vocab, eps, s);
    return cir::launch_embed<_Float16>(ids, word, pos, gamma, beta, y32, y16, rows, L, cols, vocab, eps, s);
}
