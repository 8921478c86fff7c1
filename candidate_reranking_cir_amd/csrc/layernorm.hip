// cir_layernorm / cir_embed_layernorm: row LayerNorm with fp32 statistics, one wave per row.
//
// Bound: HBM.  Algorithmic bytes per row: cols * (sx [x] + sx [residual, if any] + so [stream copy of y] + 2 [operand copy]),
// sx / so = 4 (fp32 residual stream) or 2 (fp16 residual stream); statistics and the affine map are fp32 either way.
// A row of 768 fp32 is 3 KiB = three 16-byte vectors per lane, fully coalesced; the reduction is a
// 6-step wave shuffle (no LDS, no barrier).  Mean and variance use the two-pass form on register
// data (the row is read once).
#include "common.hpp"

namespace cir {

// four consecutive elements of a stream row (fp32 or fp16) <-> float4
template <typename S> __device__ __forceinline__ float4 load4(const S* p);
template <> __device__ __forceinline__ float4 load4<float>(const float* p) { return *reinterpret_cast<const float4*>(p); }
template <> __device__ __forceinline__ float4 load4<_Float16>(const _Float16* p) {
    const f16x4 h = *reinterpret_cast<const f16x4*>(p);
    return make_float4((float)h[0], (float)h[1], (float)h[2], (float)h[3]);
}
template <typename S> __device__ __forceinline__ void store4(S* p, const float4& o);
template <> __device__ __forceinline__ void store4<float>(float* p, const float4& o) { *reinterpret_cast<float4*>(p) = o; }
template <> __device__ __forceinline__ void store4<_Float16>(_Float16* p, const float4& o) {
    u32x2 pk;
    pk.x = pack2<_Float16>(o.x, o.y);
    pk.y = pack2<_Float16>(o.z, o.w);
    *reinterpret_cast<u32x2*>(p) = pk;
}

template <typename T, typename OS, int NCH>
__device__ __forceinline__ void ln_row(float4 (&v)[NCH], int lane, int cols, const float* gamma, const float* beta,
                                       float eps, OS* y32, T* y16) {
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
            if (y32) store4<OS>(y32 + col, o);
            if (y16) {
                u32x2 p;
                p.x = pack2<T>(o.x, o.y);
                p.y = pack2<T>(o.z, o.w);
                *reinterpret_cast<u32x2*>(y16 + col) = p;
            }
        }
    }
}

template <typename T, typename XS, typename OS, int NCH>
__global__ __launch_bounds__(256) void layernorm_kernel(const XS* x, int64_t sX, const XS* res, int64_t sR,
                                                        const float* gamma, const float* beta, int64_t sG, OS* y32,
                                                        T* y16, int64_t sY, int64_t rows, int cols, float eps) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int b = blockIdx.y;
    const XS* xr = x + b * sX + row * cols;
    const XS* rr = res ? res + b * sR + row * cols : nullptr;
    float4 v[NCH];
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        const int col = (lane + c * 64) * 4;
        if (col < cols) {
            v[c] = load4<XS>(xr + col);
            if (rr) {
                const float4 r4 = load4<XS>(rr + col);
                v[c].x += r4.x; v[c].y += r4.y; v[c].z += r4.z; v[c].w += r4.w;
            }
        } else {
            v[c] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
    }
    ln_row<T, OS, NCH>(v, lane, cols, gamma + b * sG, beta + b * sG, eps, y32 ? y32 + b * sY + row * cols : nullptr,
                       y16 ? y16 + b * sY + row * cols : nullptr);
}

// fp16 residual stream: a 768-wide row is only 1.5 KiB, so with four elements per lane the loads are 8 bytes wide (half the
// rate of 16-byte accesses).  Here a HALF-wave owns a row (two rows per wave): 8 halves = 16 bytes per lane and chunk,
// reductions over 32 lanes; same arithmetic (fp32 statistics, two-pass on register data).
template <typename T, typename OS, int NCH>
__global__ __launch_bounds__(256) void layernorm_h16_kernel(const _Float16* x, int64_t sX, const _Float16* res, int64_t sR,
                                                            const float* gamma, const float* beta, int64_t sG, OS* ys,
                                                            T* y16, int64_t sY, int64_t rows, int cols, float eps) {
    const int l32 = threadIdx.x & 31;
    const int64_t row = (int64_t)blockIdx.x * 8 + (threadIdx.x >> 5);
    if (row >= rows) return;
    const int b = blockIdx.y;
    const _Float16* xr = x + b * sX + row * cols;
    const _Float16* rr = res ? res + b * sR + row * cols : nullptr;
    float v[NCH][8];
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        const int col = (l32 + c * 32) * 8;
        if (col < cols) {
            const f16x8 h = *reinterpret_cast<const f16x8*>(xr + col);
#pragma unroll
            for (int e = 0; e < 8; ++e) v[c][e] = (float)h[e];
            if (rr) {
                const f16x8 r8 = *reinterpret_cast<const f16x8*>(rr + col);
#pragma unroll
                for (int e = 0; e < 8; ++e) v[c][e] += (float)r8[e];
            }
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[c][e] = 0.f;
        }
    }
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < NCH; ++c)
#pragma unroll
        for (int e = 0; e < 8; e += 2) s += v[c][e] + v[c][e + 1];
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    const float mean = s / (float)cols;
    float q = 0.f;
#pragma unroll
    for (int c = 0; c < NCH; ++c)
        if ((l32 + c * 32) * 8 < cols) {
#pragma unroll
            for (int e = 0; e < 8; ++e) { const float d = v[c][e] - mean; q += d * d; }
        }
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) q += __shfl_xor(q, o, 64);
    const float rstd = rsqrtf(q / (float)cols + eps);
    const float* gp = gamma + b * sG;
    const float* bp = beta + b * sG;
    OS* ysr = ys ? ys + b * sY + row * cols : nullptr;
    T* y16r = y16 ? y16 + b * sY + row * cols : nullptr;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        const int col = (l32 + c * 32) * 8;
        if (col < cols) {
            float o[8];
#pragma unroll
            for (int h4 = 0; h4 < 2; ++h4) {
                const float4 g4 = *reinterpret_cast<const float4*>(gp + col + 4 * h4);
                const float4 b4 = *reinterpret_cast<const float4*>(bp + col + 4 * h4);
                o[4 * h4 + 0] = (v[c][4 * h4 + 0] - mean) * rstd * g4.x + b4.x;
                o[4 * h4 + 1] = (v[c][4 * h4 + 1] - mean) * rstd * g4.y + b4.y;
                o[4 * h4 + 2] = (v[c][4 * h4 + 2] - mean) * rstd * g4.z + b4.z;
                o[4 * h4 + 3] = (v[c][4 * h4 + 3] - mean) * rstd * g4.w + b4.w;
            }
            if (ysr) {
                if constexpr (__is_same(OS, float)) {
                    *reinterpret_cast<float4*>(ysr + col) = make_float4(o[0], o[1], o[2], o[3]);
                    *reinterpret_cast<float4*>(ysr + col + 4) = make_float4(o[4], o[5], o[6], o[7]);
                } else {
                    u32x4 p;
                    p.x = pack2<_Float16>(o[0], o[1]); p.y = pack2<_Float16>(o[2], o[3]);
                    p.z = pack2<_Float16>(o[4], o[5]); p.w = pack2<_Float16>(o[6], o[7]);
                    *reinterpret_cast<u32x4*>(ysr + col) = p;
                }
            }
            if (y16r) {
                u32x4 p;
                p.x = pack2<T>(o[0], o[1]); p.y = pack2<T>(o[2], o[3]);
                p.z = pack2<T>(o[4], o[5]); p.w = pack2<T>(o[6], o[7]);
                *reinterpret_cast<u32x4*>(y16r + col) = p;
            }
        }
    }
}

// LayerNorm of fp32 stream rows with the GEMM operand written as "split8" rows (common.hpp) next to the fp32 stream copy: the text32 mode's
// LayerNorms feed a split8 GEMM (qkv, fc1), and a separate split pass would read the stream copy back (4 bytes per element more, and a launch).
// One wave per row; per 4-element chunk a lane stores 16 B of stream, 8 B of fp16 terms and 2 x 4 B of e4m3 terms.
template <int NCH>
__global__ __launch_bounds__(256) void layernorm_split8_kernel(const float* x, int64_t sX, const float* res, int64_t sR, const float* gamma,
                                                               const float* beta, int64_t sG, float* ys, int64_t sY, char* sp, int64_t ld_sp,
                                                               int64_t s_sp, int64_t rows, int cols, float eps) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int b = blockIdx.y;
    const float* xr = x + b * sX + row * cols;
    const float* rr = res ? res + b * sR + row * cols : nullptr;
    float4 v[NCH];
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        const int col = (lane + c * 64) * 4;
        if (col < cols) {
            v[c] = *reinterpret_cast<const float4*>(xr + col);
            if (rr) {
                const float4 r4 = *reinterpret_cast<const float4*>(rr + col);
                v[c].x += r4.x; v[c].y += r4.y; v[c].z += r4.z; v[c].w += r4.w;
            }
            s += (v[c].x + v[c].y) + (v[c].z + v[c].w);
        } else {
            v[c] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
    }
    const float mean = wave_sum(s) / (float)cols;        // the same two-pass statistics, in the same order, as ln_row: the stream copies agree bit for bit
    float q = 0.f;
#pragma unroll
    for (int c = 0; c < NCH; ++c)
        if ((lane + c * 64) * 4 < cols) {
            const float dx = v[c].x - mean, dy = v[c].y - mean, dz = v[c].z - mean, dw = v[c].w - mean;
            q += (dx * dx + dy * dy) + (dz * dz + dw * dw);
        }
    const float rstd = rsqrtf(wave_sum(q) / (float)cols + eps);
    const float* gp = gamma + b * sG;
    const float* bp = beta + b * sG;
    float* yr = ys ? ys + b * sY + row * cols : nullptr;
    char* spr = sp + b * s_sp + row * ld_sp;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        const int col = (lane + c * 64) * 4;
        if (col < cols) {
            const float4 g4 = *reinterpret_cast<const float4*>(gp + col);
            const float4 b4 = *reinterpret_cast<const float4*>(bp + col);
            float o[4];
            o[0] = (v[c].x - mean) * rstd * g4.x + b4.x;
            o[1] = (v[c].y - mean) * rstd * g4.y + b4.y;
            o[2] = (v[c].z - mean) * rstd * g4.z + b4.z;
            o[3] = (v[c].w - mean) * rstd * g4.w + b4.w;
            if (yr) *reinterpret_cast<float4*>(yr + col) = make_float4(o[0], o[1], o[2], o[3]);
            const Split4 t = split8_x4(o);
            *reinterpret_cast<u32x2*>(spr + 2 * col) = u32x2{t.h01, t.h23};
            *reinterpret_cast<unsigned*>(spr + 2 * cols + col) = t.lo8;
            *reinterpret_cast<unsigned*>(spr + 3 * cols + col) = t.hi8;
        }
    }
}

template <typename T, typename OS, int NCH>
__global__ __launch_bounds__(256) void embed_ln_kernel(const int64_t* ids, const float* word, const float* pos,
                                                       const float* gamma, const float* beta, OS* y32, T* y16,
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
    ln_row<T, OS, NCH>(v, lane, cols, gamma, beta, eps, y32 ? y32 + row * cols : nullptr, y16 ? y16 + row * cols : nullptr);
}

template <typename T, typename XS, typename OS>
static int launch_ln(const void* x_, int64_t sX, const void* res_, int64_t sR, const float* gamma, const float* beta,
                     int64_t sG, void* ys_, void* y16, int64_t sY, int64_t rows, int cols, int batch, float eps,
                     hipStream_t s) {
    dim3 grid((unsigned)((rows + 3) / 4), (unsigned)batch), block(256);
    const int nch = (cols + 255) / 256;
    const XS* x = reinterpret_cast<const XS*>(x_);
    const XS* res = reinterpret_cast<const XS*>(res_);
    OS* y32 = reinterpret_cast<OS*>(ys_);
    T* y = reinterpret_cast<T*>(y16);
    if constexpr (__is_same(XS, _Float16)) {
        const bool al16 = !((reinterpret_cast<uintptr_t>(x_) | reinterpret_cast<uintptr_t>(res_) | reinterpret_cast<uintptr_t>(ys_) | reinterpret_cast<uintptr_t>(y16)) & 15u);
        if (cols % 8 == 0 && al16 && sX % 8 == 0 && sR % 8 == 0 && sY % 8 == 0) {      // 16-byte accesses: two rows per wave
            dim3 grid8((unsigned)((rows + 7) / 8), (unsigned)batch);
            switch ((cols + 255) / 256) {
                case 1: hipLaunchKernelGGL((layernorm_h16_kernel<T, OS, 1>), grid8, block, 0, s, x, sX, res, sR, gamma, beta, sG, y32, y, sY, rows, cols, eps); break;
                case 2: hipLaunchKernelGGL((layernorm_h16_kernel<T, OS, 2>), grid8, block, 0, s, x, sX, res, sR, gamma, beta, sG, y32, y, sY, rows, cols, eps); break;
                case 3: hipLaunchKernelGGL((layernorm_h16_kernel<T, OS, 3>), grid8, block, 0, s, x, sX, res, sR, gamma, beta, sG, y32, y, sY, rows, cols, eps); break;
                default: hipLaunchKernelGGL((layernorm_h16_kernel<T, OS, 4>), grid8, block, 0, s, x, sX, res, sR, gamma, beta, sG, y32, y, sY, rows, cols, eps); break;
            }
            CIR_LAUNCH_RESULT();
        }
    }
    switch (nch) {
        case 1: hipLaunchKernelGGL((layernorm_kernel<T, XS, OS, 1>), grid, block, 0, s, x, sX, res, sR, gamma, beta, sG, y32, y, sY, rows, cols, eps); break;
        case 2: hipLaunchKernelGGL((layernorm_kernel<T, XS, OS, 2>), grid, block, 0, s, x, sX, res, sR, gamma, beta, sG, y32, y, sY, rows, cols, eps); break;
        case 3: hipLaunchKernelGGL((layernorm_kernel<T, XS, OS, 3>), grid, block, 0, s, x, sX, res, sR, gamma, beta, sG, y32, y, sY, rows, cols, eps); break;
        default: hipLaunchKernelGGL((layernorm_kernel<T, XS, OS, 4>), grid, block, 0, s, x, sX, res, sR, gamma, beta, sG, y32, y, sY, rows, cols, eps); break;
    }
    CIR_LAUNCH_RESULT();
}

template <typename T, typename OS>
static int launch_embed(const int64_t* ids, const float* word, const float* pos, const float* gamma, const float* beta,
                        void* ys_, void* y16, int64_t rows, int L, int cols, int vocab, float eps, hipStream_t s) {
    dim3 grid((unsigned)((rows + 3) / 4)), block(256);
    const int nch = (cols + 255) / 256;
    OS* y32 = reinterpret_cast<OS*>(ys_);
    T* y = reinterpret_cast<T*>(y16);
    switch (nch) {
        case 1: hipLaunchKernelGGL((embed_ln_kernel<T, OS, 1>), grid, block, 0, s, ids, word, pos, gamma, beta, y32, y, rows, L, cols, vocab, eps); break;
        case 2: hipLaunchKernelGGL((embed_ln_kernel<T, OS, 2>), grid, block, 0, s, ids, word, pos, gamma, beta, y32, y, rows, L, cols, vocab, eps); break;
        case 3: hipLaunchKernelGGL((embed_ln_kernel<T, OS, 3>), grid, block, 0, s, ids, word, pos, gamma, beta, y32, y, rows, L, cols, vocab, eps); break;
        default: hipLaunchKernelGGL((embed_ln_kernel<T, OS, 4>), grid, block, 0, s, ids, word, pos, gamma, beta, y32, y, rows, L, cols, vocab, eps); break;
    }
    CIR_LAUNCH_RESULT();
}

}  // namespace cir

extern "C" int cir_layernorm(const void* x, int x_dtype, int64_t strideX, const void* residual, int64_t strideR, const float* gamma,
                             const float* beta, int64_t strideG, void* y_stream, int y_stream_dtype, void* y16, int64_t strideY,
                             int64_t rows, int cols, int batch, float eps, int dtype16, void* stream) {
    CIR_CHECK_PTR(x); CIR_CHECK_PTR(gamma); CIR_CHECK_PTR(beta);
    if (!y_stream && !y16) return CIR_EINVAL;
    if (rows <= 0 || cols <= 0 || batch <= 0) return CIR_EINVAL;
    if (cols % 4 != 0 || cols > 1024) return CIR_ESHAPE;
    if (dtype16 != CIR_BF16 && dtype16 != CIR_F16) return CIR_EDTYPE;
    if ((x_dtype != CIR_F32 && x_dtype != CIR_F16) || (y_stream && y_stream_dtype != CIR_F32 && y_stream_dtype != CIR_F16)) return CIR_EDTYPE;
    const uintptr_t xal = x_dtype == CIR_F32 ? 15u : 7u, yal = y_stream_dtype == CIR_F32 ? 15u : 7u;
    if ((reinterpret_cast<uintptr_t>(x) & xal) || !cir_aligned16(gamma) || !cir_aligned16(beta) ||
        (residual && (reinterpret_cast<uintptr_t>(residual) & xal)) || (y_stream && (reinterpret_cast<uintptr_t>(y_stream) & yal)) ||
        (y16 && (reinterpret_cast<uintptr_t>(y16) & 7)) || strideX % 4 || strideR % 4 || strideG % 4 || strideY % 4)
        return CIR_EALIGN;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const bool xf = x_dtype == CIR_F32, yf = !y_stream || y_stream_dtype == CIR_F32;
#define CIR_LN(TT)                                                                                                                   \
    do {                                                                                                                             \
        if (xf && yf) return cir::launch_ln<TT, float, float>(x, strideX, residual, strideR, gamma, beta, strideG, y_stream, y16, strideY, rows, cols, batch, eps, s);         \
        if (xf) return cir::launch_ln<TT, float, _Float16>(x, strideX, residual, strideR, gamma, beta, strideG, y_stream, y16, strideY, rows, cols, batch, eps, s);            \
        if (yf) return cir::launch_ln<TT, _Float16, float>(x, strideX, residual, strideR, gamma, beta, strideG, y_stream, y16, strideY, rows, cols, batch, eps, s);            \
        return cir::launch_ln<TT, _Float16, _Float16>(x, strideX, residual, strideR, gamma, beta, strideG, y_stream, y16, strideY, rows, cols, batch, eps, s);                 \
    } while (0)
    if (dtype16 == CIR_BF16) CIR_LN(__bf16);
    CIR_LN(_Float16);
#undef CIR_LN
}

extern "C" int cir_layernorm_split8(const float* x, int64_t strideX, const float* residual, int64_t strideR, const float* gamma, const float* beta,
                                    int64_t strideG, float* y_stream, int64_t strideY, void* y_split, int64_t ld_split_bytes,
                                    int64_t stride_split_bytes, int64_t rows, int cols, int batch, float eps, void* stream) {
    CIR_CHECK_PTR(x); CIR_CHECK_PTR(gamma); CIR_CHECK_PTR(beta); CIR_CHECK_PTR(y_split);
    if (rows <= 0 || cols <= 0 || batch <= 0) return CIR_EINVAL;
    if (cols % 16 != 0 || cols > 1024 || ld_split_bytes < 4 * (int64_t)cols) return CIR_ESHAPE;
    if (!cir_aligned16(x) || !cir_aligned16(gamma) || !cir_aligned16(beta) || (residual && !cir_aligned16(residual)) ||
        (y_stream && !cir_aligned16(y_stream)) || !cir_aligned16(y_split) || strideX % 4 || strideR % 4 || strideG % 4 || strideY % 4 ||
        ld_split_bytes % 16 || stride_split_bytes % 16)
        return CIR_EALIGN;
    dim3 grid((unsigned)((rows + 3) / 4), (unsigned)batch), block(256);
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    char* sp = reinterpret_cast<char*>(y_split);
    switch ((cols + 255) / 256) {
        case 1: hipLaunchKernelGGL((cir::layernorm_split8_kernel<1>), grid, block, 0, s, x, strideX, residual, strideR, gamma, beta, strideG, y_stream, strideY, sp, ld_split_bytes, stride_split_bytes, rows, cols, eps); break;
        case 2: hipLaunchKernelGGL((cir::layernorm_split8_kernel<2>), grid, block, 0, s, x, strideX, residual, strideR, gamma, beta, strideG, y_stream, strideY, sp, ld_split_bytes, stride_split_bytes, rows, cols, eps); break;
        case 3: hipLaunchKernelGGL((cir::layernorm_split8_kernel<3>), grid, block, 0, s, x, strideX, residual, strideR, gamma, beta, strideG, y_stream, strideY, sp, ld_split_bytes, stride_split_bytes, rows, cols, eps); break;
        default: hipLaunchKernelGGL((cir::layernorm_split8_kernel<4>), grid, block, 0, s, x, strideX, residual, strideR, gamma, beta, strideG, y_stream, strideY, sp, ld_split_bytes, stride_split_bytes, rows, cols, eps); break;
    }
    CIR_LAUNCH_RESULT();
}

extern "C" int cir_embed_layernorm(const int64_t* ids, const float* word, const float* pos, const float* gamma,
                                   const float* beta, void* y_stream, int y_stream_dtype, void* y16, int64_t rows, int L, int cols,
                                   int vocab, float eps, int dtype16, void* stream) {
    CIR_CHECK_PTR(ids); CIR_CHECK_PTR(word); CIR_CHECK_PTR(pos); CIR_CHECK_PTR(gamma); CIR_CHECK_PTR(beta);
    if (!y_stream && !y16) return CIR_EINVAL;
    if (rows <= 0 || L <= 0 || cols <= 0 || vocab <= 0) return CIR_EINVAL;
    if (cols % 4 != 0 || cols > 1024) return CIR_ESHAPE;
    if (dtype16 != CIR_BF16 && dtype16 != CIR_F16) return CIR_EDTYPE;
    if (y_stream && y_stream_dtype != CIR_F32 && y_stream_dtype != CIR_F16) return CIR_EDTYPE;
    const uintptr_t yal = y_stream_dtype == CIR_F32 ? 15u : 7u;
    if (!cir_aligned16(word) || !cir_aligned16(pos) || !cir_aligned16(gamma) || !cir_aligned16(beta) ||
        (y_stream && (reinterpret_cast<uintptr_t>(y_stream) & yal)) || (y16 && (reinterpret_cast<uintptr_t>(y16) & 7)))
        return CIR_EALIGN;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const bool yf = !y_stream || y_stream_dtype == CIR_F32;
    if (dtype16 == CIR_BF16) {
        if (yf) return cir::launch_embed<__bf16, float>(ids, word, pos, gamma, beta, y_stream, y16, rows, L, cols, vocab, eps, s);
        return cir::launch_embed<__bf16, _Float16>(ids, word, pos, gamma, beta, y_stream, y16, rows, L, cols, vocab, eps, s);
    }
    if (yf) return cir::launch_embed<_Float16, float>(ids, word, pos, gamma, beta, y_stream, y16, rows, L, cols, vocab, eps, s);
    return cir::launch_embed<_Float16, _Float16>(ids, word, pos, gamma, beta, y_stream, y16, rows, L, cols, vocab, eps, s);
}
