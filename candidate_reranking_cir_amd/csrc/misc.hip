// Small HBM-bound kernels around the GEMMs: patch im2col, ViT token assembly, the 768->2 head,
// per-row descending argsort, plus cir_version / cir_strerror.
#include "common.hpp"

namespace cir {

// ---- patchify: one thread per (patch, c, ky) moves a 16-pixel row segment (coalesced 32/64-byte reads) ----
template <typename TI, typename TO>
__global__ __launch_bounds__(256) void patchify_kernel(const TI* img, TO* out, int B, int C, int H, int Wd, int p, int vec) {
    const int gw = Wd / p, gh = H / p;
    const int64_t total = (int64_t)B * gh * gw * C * p;  // row segments
    const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= total) return;
    // consecutive threads -> consecutive px for fixed (b, c, y): reads stream along image rows
    int64_t t = gid;
    const int px = (int)(t % gw); t /= gw;
    const int ky = (int)(t % p); t /= p;
    const int py = (int)(t % gh); t /= gh;
    const int c = (int)(t % C);
    const int64_t b = t / C;
    const TI* src = img + ((b * C + c) * H + (py * p + ky)) * (int64_t)Wd + px * p;
    TO* dst = out + ((b * gh + py) * gw + px) * (int64_t)(C * p * p) + (c * p + ky) * p;
    if (vec) {   // the ViT-B/16 case (p = 16, aligned rows): a 16-pixel segment = 32 (or 64) contiguous, aligned bytes
        if constexpr (sizeof(TO) == 4) {   // fp32 pixels -> fp32 patches ("exact" mode): four 16-byte moves
            const float4* s4 = reinterpret_cast<const float4*>(src);
            float4* d4 = reinterpret_cast<float4*>(dst);
            const float4 a0 = s4[0], a1 = s4[1], a2 = s4[2], a3 = s4[3];
            d4[0] = a0; d4[1] = a1; d4[2] = a2; d4[3] = a3;
        } else if constexpr (sizeof(TI) == 2) {   // same 16-bit type in and out: two 16-byte moves
            const u32x4* s4 = reinterpret_cast<const u32x4*>(src);
            u32x4* d4 = reinterpret_cast<u32x4*>(dst);
            const u32x4 v0 = s4[0], v1 = s4[1];
            d4[0] = v0;
            d4[1] = v1;
        } else {                            // fp32 pixels: four 16-byte loads, packed to two 16-byte stores
            const float4* s4 = reinterpret_cast<const float4*>(src);
            const float4 a0 = s4[0], a1 = s4[1], a2 = s4[2], a3 = s4[3];
            u32x4 o0, o1;
            o0.x = pack2<TO>(a0.x, a0.y); o0.y = pack2<TO>(a0.z, a0.w); o0.z = pack2<TO>(a1.x, a1.y); o0.w = pack2<TO>(a1.z, a1.w);
            o1.x = pack2<TO>(a2.x, a2.y); o1.y = pack2<TO>(a2.z, a2.w); o1.z = pack2<TO>(a3.x, a3.y); o1.w = pack2<TO>(a3.z, a3.w);
            u32x4* d4 = reinterpret_cast<u32x4*>(dst);
            d4[0] = o0;
            d4[1] = o1;
        }
        return;
    }
    for (int kx = 0; kx < p; ++kx) dst[kx] = static_cast<TO>(static_cast<float>(src[kx]));
}

// ---- x[b][0] = cls + pos[0]; x[b][1+i] = proj[b*P+i] + pos[1+i] (float4 per thread) ----------------------
// S = element type of proj and x: float (fp32 residual stream) or _Float16 (16-bit stream); cls / pos are fp32 parameters
template <typename S>
__global__ __launch_bounds__(256) void vit_assemble_kernel(const S* proj, const float* cls, const float* pos, S* x,
                                                           int B, int P, int D) {
    const int d4 = D / 4;
    const int64_t total = (int64_t)B * (P + 1) * d4;
    const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= total) return;
    const int c = (int)(gid % d4);
    const int64_t row = gid / d4;
    const int tok = (int)(row % (P + 1));
    const int64_t b = row / (P + 1);
    const float4 pe = reinterpret_cast<const float4*>(pos + (int64_t)tok * D)[c];
    float4 v;
    if (tok == 0) {
        v = reinterpret_cast<const float4*>(cls)[c];
    } else if constexpr (__is_same(S, float)) {
        v = reinterpret_cast<const float4*>(proj + (b * P + tok - 1) * D)[c];
    } else {
        const f16x4 h = reinterpret_cast<const f16x4*>(proj + (b * P + tok - 1) * D)[c];
        v = make_float4((float)h[0], (float)h[1], (float)h[2], (float)h[3]);
    }
    const float4 o = make_float4(v.x + pe.x, v.y + pe.y, v.z + pe.z, v.w + pe.w);
    if constexpr (__is_same(S, float)) {
        reinterpret_cast<float4*>(x + row * D)[c] = o;
    } else {
        u32x2 pk;
        pk.x = pack2<_Float16>(o.x, o.y);
        pk.y = pack2<_Float16>(o.z, o.w);
        reinterpret_cast<u32x2*>(x + row * D)[c] = pk;
    }
}

// ---- y[m][n] = x[m] . W[n] + bias[n], N <= 8, one wave per row ---------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void small_linear_kernel(const T* x, int64_t ldx, const T* W, const float* bias, float* y,
                                                           int64_t M, int N, int K) {
    using X8 = typename Elem<T>::x8;
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    float acc[8];
#pragma unroll
    for (int n = 0; n < 8; ++n) acc[n] = 0.f;
    for (int k = lane * 8; k < K; k += 512) {
        const X8 xv = *reinterpret_cast<const X8*>(x + row * ldx + k);
#pragma unroll
        for (int n = 0; n < 8; ++n) {
            if (n < N) {
                const X8 wv = *reinterpret_cast<const X8*>(W + (int64_t)n * K + k);
#pragma unroll
                for (int j = 0; j < 8; ++j) acc[n] = fmaf(static_cast<float>(xv[j]), static_cast<float>(wv[j]), acc[n]);
            }
        }
    }
#pragma unroll
    for (int n = 0; n < 8; ++n) {
        if (n < N) {
            const float s = wave_sum(acc[n]);
            if (lane == 0) y[row * N + n] = s + (bias ? bias[n] : 0.f);
        }
    }
}


// ---- dst[i] = convert(src[index[i]]) for whole rows: candidate gather from the index-feature bank (the
// reference's torch.stack(itemgetter(*names)(name_to_feat)), validate_stage2.py:115,251), per-query ->
// per-candidate expansion of hidden states, and dtype conversion at the API boundary. 8 elements per thread. ----
template <typename TI, typename TO>
__global__ __launch_bounds__(256) void gather_rows_kernel(const TI* src, const int64_t* index, TO* dst, int64_t n_rows,
                                                          int64_t row_elems, int64_t src_rows) {
    const int64_t vec_per_row = row_elems / 8;
    const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= n_rows * vec_per_row) return;
    const int64_t row = gid / vec_per_row, c = (gid - row * vec_per_row) * 8;
    int64_t sr = index ? index[row] : row;
    sr = sr < 0 ? 0 : (sr >= src_rows ? src_rows - 1 : sr);
    const TI* s = src + sr * row_elems + c;
    TO* d = dst + row * row_elems + c;
    float v[8];
    if constexpr (sizeof(TI) == 4) {
        const float4 a = reinterpret_cast<const float4*>(s)[0], b = reinterpret_cast<const float4*>(s)[1];
        v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
    } else {
        typedef __attribute__((ext_vector_type(8))) TI ti8;
        const ti8 a = *reinterpret_cast<const ti8*>(s);
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = static_cast<float>(a[j]);
    }
    if constexpr (sizeof(TO) == 4) {
        reinterpret_cast<float4*>(d)[0] = make_float4(v[0], v[1], v[2], v[3]);
        reinterpret_cast<float4*>(d)[1] = make_float4(v[4], v[5], v[6], v[7]);
    } else {
        u32x4 o;
        o.x = pack2<TO>(v[0], v[1]); o.y = pack2<TO>(v[2], v[3]); o.z = pack2<TO>(v[4], v[5]); o.w = pack2<TO>(v[6], v[7]);
        *reinterpret_cast<u32x4*>(d) = o;
    }
}


// ---- y = x W^T + bias (mode 0), 1 - x W^T (mode 1) or x W^T - 1 (mode 2 = the exact negative of mode 1), all fp32: the 256-d retrieval heads and the cosine-distance
// matrix of stage I (validate.py:57, 202; blip_stage1.py:83).  A few GFLOP per split: a plain LDS-tiled FMA kernel. ----
__global__ __launch_bounds__(256) void linear_f32_kernel(const float* x, int64_t ldx, const float* W, const float* bias, float* y,
                                                        int64_t M, int N, int K, int mode) {
    __shared__ float As[16][65], Ws[16][65];
    const int tid = threadIdx.x, tx = tid & 15, ty = tid >> 4;
    const int64_t row0 = (int64_t)blockIdx.y * 64;
    const int col0 = blockIdx.x * 64;
    float acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = 0.f;
    for (int k0 = 0; k0 < K; k0 += 16) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int e = tid + i * 256, r = e >> 4, kk = e & 15;
            As[kk][r] = (row0 + r < M && k0 + kk < K) ? x[(row0 + r) * ldx + k0 + kk] : 0.f;
            Ws[kk][r] = (col0 + r < N && k0 + kk < K) ? W[(int64_t)(col0 + r) * K + k0 + kk] : 0.f;
        }
        __syncthreads();
#pragma unroll
        for (int kk = 0; kk < 16; ++kk) {
            float av[4], wv[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) { av[i] = As[kk][ty * 4 + i]; wv[i] = Ws[kk][tx * 4 + i]; }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = fmaf(av[i], wv[j], acc[i][j]);
        }
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int64_t m = row0 + ty * 4 + i;
        if (m >= M) continue;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int n = col0 + tx * 4 + j;
            if (n >= N) continue;
            const float v = acc[i][j] + (bias ? bias[n] : 0.f);
            y[m * N + n] = mode == 1 ? 1.0f - v : (mode == 2 ? v - 1.0f : v);
        }
    }
}

// ---- y = x / max(||x||_2, 1e-12) per row (F.normalize, blip_stage1.py:58, 83), one wave per row ----
__global__ __launch_bounds__(256) void l2_normalize_kernel(const float* x, float* y, int64_t rows, int cols) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    float s = 0.f;
    for (int c = lane; c < cols; c += 64) { const float v = x[row * cols + c]; s = fmaf(v, v, s); }
    const float inv = 1.0f / fmaxf(sqrtf(wave_sum(s)), 1e-12f);
    for (int c = lane; c < cols; c += 64) y[row * cols + c] = x[row * cols + c] * inv;
}

// ---- descending argsort of one row per workgroup: bitonic network on (value, index) in LDS ------------------
__global__ __launch_bounds__(256) void topk_desc_kernel(const float* logits, int64_t* idx, int K, int n_pow2) {
    extern __shared__ __attribute__((aligned(16))) char dyn[];
    float* val = reinterpret_cast<float*>(dyn);
    int* ind = reinterpret_cast<int*>(dyn + (size_t)n_pow2 * 4);
    const float* row = logits + (int64_t)blockIdx.x * K;
    for (int i = threadIdx.x; i < n_pow2; i += blockDim.x) {
        float v = i < K ? row[i] : -INFINITY;
        if (v != v) v = -INFINITY;  // NaN sorts last
        val[i] = v;
        ind[i] = i < K ? i : 0x7fffffff;
    }
    __syncthreads();
    // "a before b" <=> a.val > b.val, or equal values and a.idx < b.idx (padding has idx = INT_MAX)
    for (int size = 2; size <= n_pow2; size <<= 1) {
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            for (int t = threadIdx.x; t < n_pow2 / 2; t += blockDim.x) {
                const int lo = 2 * t - (t & (stride - 1));
                const int hi = lo + stride;
                const bool asc_block = ((lo & size) != 0);  // alternate direction to build bitonic runs
                const float va = val[lo], vb = val[hi];
                const int ia = ind[lo], ib = ind[hi];
                const bool a_first = (va > vb) || (va == vb && ia < ib);
                const bool swap = asc_block ? a_first : !a_first;
                if (swap) { val[lo] = vb; val[hi] = va; ind[lo] = ib; ind[hi] = ia; }
            }
            __syncthreads();
        }
    }
    for (int i = threadIdx.x; i < K; i += blockDim.x) idx[(int64_t)blockIdx.x * K + i] = ind[i];
}

}  // namespace cir

extern "C" int cir_version(void) { return CIR_ABI_VERSION; }

namespace cir { int g_tune[3] = {0, 0, 0}; }

extern "C" int cir_set_tuning(int knob, int value) {
    switch (knob) {
        case CIR_TUNE_GEMM_TILE: if (value != 0 && value != 128 && value != 256) return CIR_EINVAL; break;
        case CIR_TUNE_GEMM_GROUP_W: if (value < 0 || value > 64) return CIR_EINVAL; break;
        case CIR_TUNE_ATTN_SHARED_MAX: if (value != 0 && value != -1 && value != -2 && (value < 32 || value > 608)) return CIR_EINVAL; break;
        default: return CIR_EINVAL;
    }
    cir::g_tune[knob] = value;
    return CIR_OK;
}

extern "C" const char* cir_strerror(int code) {
    switch (code) {
        case CIR_OK: return "ok";
        case CIR_EINVAL: return "invalid argument (null pointer or non-positive extent)";
        case CIR_ESHAPE: return "extent not supported by the gfx950 kernels";
        case CIR_EALIGN: return "pointer or stride not aligned for 16-byte vector access";
        case CIR_EDTYPE: return "dtype code not supported by this entry point";
        default: return code > 0 ? hipGetErrorString((hipError_t)code) : "unknown cirrank error";
    }
}

extern "C" int cir_patchify(const void* image, int img_dtype, void* patches, int dtype16, int B, int C, int H, int Wd,
                            int patch, void* stream) {
    using namespace cir;
    CIR_CHECK_PTR(image); CIR_CHECK_PTR(patches);
    if (B <= 0 || C <= 0 || H <= 0 || Wd <= 0 || patch <= 0) return CIR_EINVAL;
    if (H % patch || Wd % patch) return CIR_ESHAPE;
    if (dtype16 != CIR_BF16 && dtype16 != CIR_F16 && dtype16 != CIR_F32) return CIR_EDTYPE;
    if (img_dtype != CIR_F32 && img_dtype != dtype16) return CIR_EDTYPE;
    const int64_t total = (int64_t)B * C * H * (Wd / patch);
    dim3 grid((unsigned)((total + 255) / 256)), block(256);
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const int vec = patch == 16 && Wd % 16 == 0 && cir_aligned16(image) && cir_aligned16(patches);
    if (dtype16 == CIR_F32) {   // "exact" mode: the patches stay fp32 (ABI v11)
        hipLaunchKernelGGL((patchify_kernel<float, float>), grid, block, 0, s, (const float*)image, (float*)patches, B, C, H, Wd, patch, vec);
        CIR_LAUNCH_RESULT();
    }
    if (dtype16 == CIR_BF16) {
        if (img_dtype == CIR_F32) hipLaunchKernelGGL((patchify_kernel<float, __bf16>), grid, block, 0, s, (const float*)image, (__bf16*)patches, B, C, H, Wd, patch, vec);
        else hipLaunchKernelGGL((patchify_kernel<__bf16, __bf16>), grid, block, 0, s, (const __bf16*)image, (__bf16*)patches, B, C, H, Wd, patch, vec);
    } else {
        if (img_dtype == CIR_F32) hipLaunchKernelGGL((patchify_kernel<float, _Float16>), grid, block, 0, s, (const float*)image, (_Float16*)patches, B, C, H, Wd, patch, vec);
        else hipLaunchKernelGGL((patchify_kernel<_Float16, _Float16>), grid, block, 0, s, (const _Float16*)image, (_Float16*)patches, B, C, H, Wd, patch, vec);
    }
    CIR_LAUNCH_RESULT();
}

extern "C" int cir_vit_assemble(const void* proj, const float* cls, const float* pos, void* x, int stream_dtype, int B, int P, int D,
                                void* stream) {
    CIR_CHECK_PTR(proj); CIR_CHECK_PTR(cls); CIR_CHECK_PTR(pos); CIR_CHECK_PTR(x);
    if (B <= 0 || P <= 0 || D <= 0) return CIR_EINVAL;
    if (D % 8) return CIR_ESHAPE;
    if (stream_dtype != CIR_F32 && stream_dtype != CIR_F16) return CIR_EDTYPE;
    if (!cir_aligned16(proj) || !cir_aligned16(cls) || !cir_aligned16(pos) || !cir_aligned16(x)) return CIR_EALIGN;
    const int64_t total = (int64_t)B * (P + 1) * (D / 4);
    dim3 grid((unsigned)((total + 255) / 256)), block(256);
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (stream_dtype == CIR_F32)
        hipLaunchKernelGGL(cir::vit_assemble_kernel<float>, grid, block, 0, s, reinterpret_cast<const float*>(proj), cls, pos, reinterpret_cast<float*>(x), B, P, D);
    else
        hipLaunchKernelGGL(cir::vit_assemble_kernel<_Float16>, grid, block, 0, s, reinterpret_cast<const _Float16*>(proj), cls, pos, reinterpret_cast<_Float16*>(x), B, P, D);
    CIR_LAUNCH_RESULT();
}

extern "C" int cir_small_linear(const void* x, int64_t ldx, const void* W, const float* bias, float* y, int64_t M, int N,
                                int K, int dtype, void* stream) {
    using namespace cir;
    CIR_CHECK_PTR(x); CIR_CHECK_PTR(W); CIR_CHECK_PTR(y);
    if (M <= 0 || N <= 0 || K <= 0) return CIR_EINVAL;
    if (N > 8 || K % 8) return CIR_ESHAPE;
    if (dtype != CIR_BF16 && dtype != CIR_F16) return CIR_EDTYPE;
    if (!cir_aligned16(x) || !cir_aligned16(W) || ldx % 8) return CIR_EALIGN;
    dim3 grid((unsigned)((M + 3) / 4)), block(256);
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (dtype == CIR_BF16) hipLaunchKernelGGL((small_linear_kernel<__bf16>), grid, block, 0, s, (const __bf16*)x, ldx, (const __bf16*)W, bias, y, M, N, K);
    else hipLaunchKernelGGL((small_linear_kernel<_Float16>), grid, block, 0, s, (const _Float16*)x, ldx, (const _Float16*)W, bias, y, M, N, K);
    CIR_LAUNCH_RESULT();
}

extern "C" int cir_topk_desc(const float* logits, int64_t* idx, int Q, int K, void* stream) {
    CIR_CHECK_PTR(logits); CIR_CHECK_PTR(idx);
    if (Q <= 0 || K <= 0) return CIR_EINVAL;
    if (K > 8192) return CIR_ESHAPE;   // 8192 (value, index) pairs = 64 KiB of LDS: every index of the reference's datasets fits
    int n = 2;
    while (n < K) n <<= 1;
    dim3 grid((unsigned)Q), block(256);
    hipLaunchKernelGGL(cir::topk_desc_kernel, grid, block, (size_t)n * 8, reinterpret_cast<hipStream_t>(stream), logits, idx, K, n);
    CIR_LAUNCH_RESULT();
}

namespace cir {
template <typename TI, typename TO>
static void launch_gather(const void* src, const int64_t* index, void* dst, int64_t n_rows, int64_t row_elems, int64_t src_rows, hipStream_t s) {
    const int64_t total = n_rows * (row_elems / 8);
    dim3 grid((unsigned)((total + 255) / 256)), block(256);
    hipLaunchKernelGGL((gather_rows_kernel<TI, TO>), grid, block, 0, s, (const TI*)src, index, (TO*)dst, n_rows, row_elems, src_rows);
}
}  // namespace cir

extern "C" int cir_gather_rows(const void* src, int src_dtype, const int64_t* index, void* dst, int dst_dtype, int64_t n_rows,
                               int64_t row_elems, int64_t src_rows, void* stream) {
    using namespace cir;
    CIR_CHECK_PTR(src); CIR_CHECK_PTR(dst);
    if (n_rows <= 0 || row_elems <= 0 || src_rows <= 0) return CIR_EINVAL;
    if (row_elems % 8) return CIR_ESHAPE;
    if (!cir_aligned16(src) || !cir_aligned16(dst)) return CIR_EALIGN;
    if (n_rows * (row_elems / 8) > 0x7fffffffLL * 256) return CIR_ESHAPE;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
#define CIR_G(SD, DD, TI, TO) if (src_dtype == SD && dst_dtype == DD) { launch_gather<TI, TO>(src, index, dst, n_rows, row_elems, src_rows, s); CIR_LAUNCH_RESULT(); }
    CIR_G(CIR_F32, CIR_F32, float, float)
    CIR_G(CIR_F32, CIR_BF16, float, __bf16)
    CIR_G(CIR_F32, CIR_F16, float, _Float16)
    CIR_G(CIR_BF16, CIR_BF16, __bf16, __bf16)
    CIR_G(CIR_F16, CIR_F16, _Float16, _Float16)
    CIR_G(CIR_BF16, CIR_F32, __bf16, float)
    CIR_G(CIR_F16, CIR_F32, _Float16, float)
    CIR_G(CIR_BF16, CIR_F16, __bf16, _Float16)
    CIR_G(CIR_F16, CIR_BF16, _Float16, __bf16)
#undef CIR_G
    return CIR_EDTYPE;
}

namespace cir {
// fp32 -> two fp16 terms hi + lo (hi = fp16(y), lo = fp16(y - hi)), y = act(x): the operand split of the 3-product text-side GEMMs
// (text32 mode).  fp16 subnormals are kept by the MFMA, so lo needs no scaling: |lo| <= 2^-12 |y| is held to 2^-25 absolute.
// 8 elements per thread; rows of `cols` elements with leading dimension ldx (elements); outputs with leading dimension ldo, `hi2`
// (optional) receives a second copy of hi: with hi = buf, lo = buf + cols, hi2 = buf + 2 cols, ldo = 3 cols one row of buf is
// [hi | lo | hi] - against a weight row [W_hi | W_hi | W_lo] ONE fp16 GEMM of depth 3 cols forms all three products in one accumulator.
template <int ACT>
__global__ __launch_bounds__(256) void split16_kernel(const float* __restrict__ x, int64_t ldx, _Float16* __restrict__ hi, _Float16* __restrict__ lo,
                                                      _Float16* __restrict__ hi2, int64_t ldo, int64_t rows, int cols) {
    const int64_t per_row = cols >> 3;
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= rows * per_row) return;
    const int64_t r = i / per_row;
    const int c = (int)(i - r * per_row) * 8;
    const float4 v0 = *reinterpret_cast<const float4*>(x + r * ldx + c), v1 = *reinterpret_cast<const float4*>(x + r * ldx + c + 4);
    float v[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
    f16x8 h, l;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        float y = v[j];
        if (ACT == CIR_ACT_GELU) y = 0.5f * y * (1.0f + erff(y * 0.70710678118654752f));   // as written in the reference (exact mode's GELU)
        else if (ACT == CIR_ACT_RELU) y = fmaxf(y, 0.f);
        asm volatile("" : "+v"(y));          // ONE value of y feeds both terms: re-evaluated with another contraction it can land on the
        const _Float16 hj = (_Float16)y;     // other side of a rounding tie, and hi + lo then misses y by an fp16 ulp (seen: 159 of 3.8 M)
        float hf = (float)hj;
        asm volatile("" : "+v"(hf));
        h[j] = hj;
        l[j] = (_Float16)(y - hf);
    }
    *reinterpret_cast<f16x8*>(hi + r * ldo + c) = h;
    *reinterpret_cast<f16x8*>(lo + r * ldo + c) = l;
    if (hi2 != nullptr) *reinterpret_cast<f16x8*>(hi2 + r * ldo + c) = h;
}
// fp32 rows -> "split8" operand rows [hi fp16 | lo8 | hi8] (common.hpp), optionally through an activation: the stand-alone producer
// (the LayerNorm, the fp32 attention and the GEMM epilogue emit the same rows themselves).  8 elements per thread.
template <int ACT>
__global__ __launch_bounds__(256) void split8_kernel(const float* __restrict__ x, int64_t ldx, char* __restrict__ out, int64_t ldo, int64_t rows, int cols) {
    const int64_t per_row = cols >> 3;
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= rows * per_row) return;
    const int64_t r = i / per_row;
    const int c = (int)(i - r * per_row) * 8;
    const float4 v0 = *reinterpret_cast<const float4*>(x + r * ldx + c), v1 = *reinterpret_cast<const float4*>(x + r * ldx + c + 4);
    float v[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
    if (ACT == CIR_ACT_GELU) gelu_erf_as8(v);
    else if (ACT == CIR_ACT_RELU) {
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = fmaxf(v[j], 0.f);
    }
    const float a[4] = {v[0], v[1], v[2], v[3]}, b[4] = {v[4], v[5], v[6], v[7]};
    const Split4 sa = split8_x4(a), sb = split8_x4(b);
    const u32x4 h = {sa.h01, sa.h23, sb.h01, sb.h23};
    const u32x2 l8 = {sa.lo8, sb.lo8}, h8 = {sa.hi8, sb.hi8};
    char* row = out + r * ldo;
    *reinterpret_cast<u32x4*>(row + 2 * c) = h;
    *reinterpret_cast<u32x2*>(row + 2 * cols + c) = l8;
    *reinterpret_cast<u32x2*>(row + 3 * cols + c) = h8;
}
}  // namespace cir


extern "C" int cir_split16(const float* x, int64_t ldx, void* hi, void* lo, void* hi2, int64_t ldo, int64_t rows, int cols, int act, void* stream) {
    using namespace cir;
    CIR_CHECK_PTR(x); CIR_CHECK_PTR(hi); CIR_CHECK_PTR(lo);
    if (rows <= 0 || cols <= 0) return CIR_EINVAL;
    if (cols % 8 || ldx % 4 || ldo % 8 || ldo < cols) return CIR_ESHAPE;
    if (act < CIR_ACT_NONE || act > CIR_ACT_RELU) return CIR_EINVAL;
    if (!cir_aligned16(x) || !cir_aligned16(hi) || !cir_aligned16(lo) || (hi2 && !cir_aligned16(hi2))) return CIR_EALIGN;
    const int64_t n = rows * (cols / 8);
    if (n > 0x7fffffffLL * 256) return CIR_ESHAPE;
    dim3 grid((unsigned)((n + 255) / 256)), block(256);
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    _Float16* h = reinterpret_cast<_Float16*>(hi);
    _Float16* l = reinterpret_cast<_Float16*>(lo);
    _Float16* h2 = reinterpret_cast<_Float16*>(hi2);
    if (act == CIR_ACT_GELU) hipLaunchKernelGGL((split16_kernel<CIR_ACT_GELU>), grid, block, 0, s, x, ldx, h, l, h2, ldo, rows, cols);
    else if (act == CIR_ACT_RELU) hipLaunchKernelGGL((split16_kernel<CIR_ACT_RELU>), grid, block, 0, s, x, ldx, h, l, h2, ldo, rows, cols);
    else hipLaunchKernelGGL((split16_kernel<CIR_ACT_NONE>), grid, block, 0, s, x, ldx, h, l, h2, ldo, rows, cols);
    CIR_LAUNCH_RESULT();
}

extern "C" int cir_split8(const float* x, int64_t ldx, void* out, int64_t ldo_bytes, int64_t rows, int cols, int act, void* stream) {
    using namespace cir;
    CIR_CHECK_PTR(x); CIR_CHECK_PTR(out);
    if (rows <= 0 || cols <= 0) return CIR_EINVAL;
    if (cols % 8 || ldx % 4 || ldo_bytes % 16 || ldo_bytes < 4 * (int64_t)cols) return CIR_ESHAPE;
    if (act < CIR_ACT_NONE || act > CIR_ACT_RELU) return CIR_EINVAL;
    if (!cir_aligned16(x) || !cir_aligned16(out) || cols % 16) return CIR_EALIGN;     // the three segments of a row start 16-byte aligned
    const int64_t n = rows * (cols / 8);
    if (n > 0x7fffffffLL * 256) return CIR_ESHAPE;
    dim3 grid((unsigned)((n + 255) / 256)), block(256);
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    char* o = reinterpret_cast<char*>(out);
    if (act == CIR_ACT_GELU) hipLaunchKernelGGL((split8_kernel<CIR_ACT_GELU>), grid, block, 0, s, x, ldx, o, ldo_bytes, rows, cols);
    else if (act == CIR_ACT_RELU) hipLaunchKernelGGL((split8_kernel<CIR_ACT_RELU>), grid, block, 0, s, x, ldx, o, ldo_bytes, rows, cols);
    else hipLaunchKernelGGL((split8_kernel<CIR_ACT_NONE>), grid, block, 0, s, x, ldx, o, ldo_bytes, rows, cols);
    CIR_LAUNCH_RESULT();
}

extern "C" int cir_linear_f32(const float* x, int64_t ldx, const float* W, const float* bias, float* y, int64_t M, int N, int K,
                              int mode, void* stream) {
    CIR_CHECK_PTR(x); CIR_CHECK_PTR(W); CIR_CHECK_PTR(y);
    if (M <= 0 || N <= 0 || K <= 0) return CIR_EINVAL;
    if (mode < 0 || mode > 2) return CIR_EINVAL;
    dim3 grid((unsigned)((N + 63) / 64), (unsigned)((M + 63) / 64)), block(256);
    if (grid.y > 65535u * 16u) return CIR_ESHAPE;
    hipLaunchKernelGGL(cir::linear_f32_kernel, grid, block, 0, reinterpret_cast<hipStream_t>(stream), x, ldx, W, bias, y, M, N, K, mode);
    CIR_LAUNCH_RESULT();
}

extern "C" int cir_l2_normalize(const float* x, float* y, int64_t rows, int cols, void* stream) {
    CIR_CHECK_PTR(x); CIR_CHECK_PTR(y);
    if (rows <= 0 || cols <= 0) return CIR_EINVAL;
    dim3 grid((unsigned)((rows + 3) / 4)), block(256);
    hipLaunchKernelGGL(cir::l2_normalize_kernel, grid, block, 0, reinterpret_cast<hipStream_t>(stream), x, y, rows, cols);
    CIR_LAUNCH_RESULT();
}
