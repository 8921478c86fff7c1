// cir_gemm_bias_act: C = act(A * W^T + bias) (+ residual) on MFMA, 16-bit inputs / fp32 accumulate.
//
// Bound: MFMA (dense bf16/f16, ~2.5 PFLOP/s peak on MI355X).  Algorithmic work 2*M*N*K flop.
//
// Structure (v1: 128x128x64 tile, 4 waves, LDS-DMA staging, two LDS buffers):
//   * A (activations, M x K) and W (weights, N x K, torch Linear layout) are both K-contiguous, so
//     a 16-byte chunk of a row is directly one MFMA fragment (8 consecutive k for one row).
//   * Tiles are staged with global_load_lds (16 B/lane, 1 KiB per wave-instruction = 8 rows x 128 B).
//     The LDS image is lane-linear, so the bank swizzle lives on the per-lane SOURCE address:
//     LDS row r, 16-B slot s holds global chunk s ^ (r & 7); fragments are read back with the same
//     XOR (ds_read_b128, conflict-free for the 16x16x32 operand pattern).
//   * MFMA orientation is "swapped": weights are the first operand, activations the second, so
//     the accumulator holds 4 consecutive output FEATURES per register quad for one row.  The W
//     tile's rows are permuted on their way into LDS so that a lane ends up with 16 consecutive
//     features of one output row -> 16/32/64-byte vector stores, fp32 bias/residual as float4.
//   * Workgroup -> tile mapping is XCD-aware (blocks that share an XCD's L2 walk neighbouring
//     tiles of one row panel, so the A panel is fetched from HBM once per XCD).

#include <type_traits>

#include "gemm_args.hpp"

namespace cir {

constexpr int BM = 128, BN = 128, BK = 64;
constexpr int kTileBytes = BM * BK * 2;  // 16 KiB per operand tile: 128 rows of 128 bytes = 64 16-bit or 32 fp32 k-values per row

// fp32 operands (the "exact" precision mode, round 5): the same tile image - a 128-byte row holds 32 fp32 k-values, a 16-byte
// chunk 4 of them - on v_mfma_f32_16x16x4_f32 (f32 in, f32 accumulate: a chain of IEEE fmaf's, 1/16 of the 16-bit MFMA rate,
// 157.3 TFLOP/s peak).  Lane group g reads chunk g of a 64-byte k-group as before; the chunk's 4 values feed 4 MFMAs, the
// s-th of which contracts k = {s, 4 + s, 8 + s, 12 + s} of the group (both operands use the same map, and a sum over k does
// not care in which MFMA a k-value rides).  Accumulator layout = the 16x16x32 one, so staging, swizzle, W-row permutation
// and epilogue are shared with the 16-bit instantiations.
template <> struct Elem<float> {
    using x8 = f32x4;     // one 16-byte fragment chunk
};
template <typename T> struct Tile {
    static constexpr int kChunk = 16 / (int)sizeof(T);      // elements per 16-byte chunk
    static constexpr int kBK = 128 / (int)sizeof(T);        // k-values per tile row
};
template <typename T>
__device__ __forceinline__ void mma_frag(f32x4& acc, const typename Elem<T>::x8& w, const typename Elem<T>::x8& x) {
    if constexpr (std::is_same<T, float>::value) {
#pragma unroll
        for (int s = 0; s < 4; ++s) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(w[s], x[s], acc, 0, 0, 0);
    } else {
        acc = Elem<T>::mfma16(w, x, acc);
    }
}


// OUT: 0 = C in the operand type T, 1 = fp32 C, 2 = fp16 C (the 16-bit residual stream).  The residual is fp32 for
// OUT 0 / 1 and fp16 for OUT 2 (element type of the stream it is part of); the sum is formed in fp32 and rounded once.
template <typename T, int OUT>
__global__ __launch_bounds__(256, 2) void gemm_kernel(const GemmArgs a) {
    constexpr bool OUT_F32 = OUT == 1;
    using X8 = typename Elem<T>::x8;
    __shared__ __attribute__((aligned(16))) char smem[4 * kTileBytes];  // [buf][A|W]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave & 1, wn = wave >> 1;
    const int r15 = lane & 15, g = lane >> 4;

    // ---- which tile ---------------------------------------------------------------------------
    const int nblk = gridDim.x;
    int id = xcd_remap(blockIdx.x, nblk);
    const int per_batch = a.tiles_m * a.tiles_n;
    const int z = id / per_batch;
    id -= z * per_batch;
    const int tile_m = id / a.tiles_n, tile_n = id - tile_m * a.tiles_n;
    const int64_t m0 = (int64_t)tile_m * BM;
    const int n0 = tile_n * BN;

    const T* A = reinterpret_cast<const T*>(a.A) + z * a.sA;
    const T* W = reinterpret_cast<const T*>(a.W) + z * a.sW;

    // ---- per-lane staging sources: 4 wave-instructions of 8 rows for each operand ---------------
    const int srow = lane >> 3;                 // row inside the 8-row piece
    const int schunk = (lane & 7) ^ srow;       // global 16-B chunk this lane fetches (swizzle on the source)
    const T* a_src[4];
    const T* w_src[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int lrow = (wave * 4 + j) * 8 + srow;           // LDS row 0..127
        int64_t gm = m0 + lrow;
        gm = gm < a.M ? gm : a.M - 1;                         // clamp the ragged M edge (stores are predicated)
        a_src[j] = A + gm * a.lda + schunk * Tile<T>::kChunk;
        // W rows are permuted so that lane group g of the accumulator owns 16 consecutive features
        const int perm = (lrow & 64) + ((lrow & 15) >> 2) * 16 + ((lrow >> 4) & 3) * 4 + (lrow & 3);
        int gn = n0 + perm;
        gn = gn < a.N ? gn : a.N - 1;
        w_src[j] = W + (int64_t)gn * a.ldw + schunk * Tile<T>::kChunk;
    }

    auto stage = [&](int kt, int buf) {
        char* base = smem + buf * 2 * kTileBytes + wave * 4096;
        const int koff = kt * Tile<T>::kBK;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            __builtin_amdgcn_global_load_lds((gptr_t)(a_src[j] + koff), (lptr_t)(base + j * 1024), 16, 0, 0);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            __builtin_amdgcn_global_load_lds((gptr_t)(w_src[j] + koff), (lptr_t)(base + kTileBytes + j * 1024), 16, 0, 0);
        }
    };

    // Accumulators start at the bias, as in the 256 x 256 kernel: the two kernels then form every output with the SAME sequence
    // of fp32 operations (bias, then the K-steps in ascending order through the same MFMA shape), so which tile size the
    // dispatcher picks - it depends on the batch size - never changes a bit of the result.
    f32x4 acc[4][4];
    {
        const int nb_ = n0 + wn * 64 + g * 16;
        float bias_[16];
        if (a.bias != nullptr && nb_ + 16 <= a.N) {
            const float4* bp = reinterpret_cast<const float4*>(a.bias + z * a.sBias + nb_);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4 b4 = bp[q];
                bias_[q * 4 + 0] = b4.x; bias_[q * 4 + 1] = b4.y; bias_[q * 4 + 2] = b4.z; bias_[q * 4 + 3] = b4.w;
            }
        } else {
#pragma unroll
            for (int q = 0; q < 16; ++q) bias_[q] = 0.f;
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{bias_[j * 4 + 0], bias_[j * 4 + 1], bias_[j * 4 + 2], bias_[j * 4 + 3]};
    }

    // fragment read offsets (bytes) inside a tile: row*128 + ((kc ^ (row&7)) << 4)
    const int a_row_off = (wm * 64 + r15) * 128;
    const int w_row_off = (wn * 64 + r15) * 128;
    const int swz = r15 & 7;

    const int nk = a.K / Tile<T>::kBK;
    stage(0, 0);
    for (int kt = 0; kt < nk; ++kt) {
        // The LDS-DMA pieces of tile kt must have landed before any wave reads them.  The wait is EXPLICIT: hipcc's wait-count
        // pass dropped the vmcnt(0) it used to put in front of this barrier once a VGPR-destination load (the bias) preceded
        // the loop (round 4: every K >= 128 result wrong, K = 64 right) - do not rely on it for DMA-written LDS.
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();  // tile kt landed everywhere; buffer (kt+1)&1 is free again
        if (kt + 1 < nk) stage(kt + 1, (kt + 1) & 1);
        const char* As = smem + (kt & 1) * 2 * kTileBytes;
        const char* Ws = As + kTileBytes;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const int coff = ((ks * 4 + g) ^ swz) << 4;
            X8 af[4], wf[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                af[i] = *reinterpret_cast<const X8*>(As + a_row_off + i * 2048 + coff);
                wf[i] = *reinterpret_cast<const X8*>(Ws + w_row_off + i * 2048 + coff);
            }
#pragma unroll
            for (int mi = 0; mi < 4; ++mi)
#pragma unroll
                for (int ni = 0; ni < 4; ++ni) mma_frag<T>(acc[mi][ni], wf[ni], af[mi]);
        }
    }

    // ---- epilogue: lane (r15, g) owns, per mi, row m and features nb .. nb+15 ----------------------
    const int nb = n0 + wn * 64 + g * 16;
    if (nb + 16 > a.N) return;
    const bool has_res = a.R != nullptr;
#pragma unroll
    for (int mi = 0; mi < 4; ++mi) {
        const int64_t m = m0 + wm * 64 + mi * 16 + r15;
        if (m >= a.M) continue;
        float v[16];
#pragma unroll
        for (int ni = 0; ni < 4; ++ni)
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) v[ni * 4 + jj] = acc[mi][ni][jj];
        if (a.act == CIR_ACT_GELU && std::is_same<T, float>::value) {
#pragma unroll
            for (int q = 0; q < 16; ++q) v[q] = 0.5f * v[q] * (1.0f + erff(v[q] * 0.70710678118654752f));   // ACT2FN['gelu'] / nn.GELU, as written
        } else if (a.act == CIR_ACT_GELU) {
#pragma unroll
            for (int q = 0; q < 16; q += 8) {      // the 256 x 256 kernel's packed evaluation: the same operations, the same bits
                float w8[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) w8[e] = v[q + e];
                gelu_erf8(w8);
#pragma unroll
                for (int e = 0; e < 8; ++e) v[q + e] = w8[e];
            }
        } else if (a.act == CIR_ACT_RELU) {
#pragma unroll
            for (int q = 0; q < 16; ++q) v[q] = fmaxf(v[q], 0.f);
        }
        if (has_res) {
            if constexpr (OUT == 2) {
                typedef __attribute__((ext_vector_type(8))) _Float16 h8;
                const h8* rp = reinterpret_cast<const h8*>(reinterpret_cast<const _Float16*>(a.R) + z * a.sR + m * a.ldr + nb);
#pragma unroll
                for (int q = 0; q < 2; ++q) {       // acc + bias is rounded to the stream type BEFORE the residual joins it (and the
                    const h8 r8 = rp[q];            // sum is rounded again): what the 256 x 256 epilogue does in its 16-bit row layout
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[q * 8 + e] = (float)(_Float16)v[q * 8 + e] + (float)r8[e];
                }
            } else {
                const float4* rp = reinterpret_cast<const float4*>(reinterpret_cast<const float*>(a.R) + z * a.sR + m * a.ldr + nb);
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float4 r4 = rp[q];
                    v[q * 4 + 0] += r4.x; v[q * 4 + 1] += r4.y; v[q * 4 + 2] += r4.z; v[q * 4 + 3] += r4.w;
                }
            }
        }
        if constexpr (OUT_F32) {
            float4* cp = reinterpret_cast<float4*>(reinterpret_cast<float*>(a.C) + z * a.sC + m * a.ldc + nb);
#pragma unroll
            for (int q = 0; q < 4; ++q) cp[q] = make_float4(v[q * 4], v[q * 4 + 1], v[q * 4 + 2], v[q * 4 + 3]);
        } else {
            using CT = typename std::conditional<OUT == 2, _Float16, T>::type;
            u32x4* cp = reinterpret_cast<u32x4*>(reinterpret_cast<CT*>(a.C) + z * a.sC + m * a.ldc + nb);
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                u32x4 o;
                o.x = pack2<CT>(v[q * 8 + 0], v[q * 8 + 1]);
                o.y = pack2<CT>(v[q * 8 + 2], v[q * 8 + 3]);
                o.z = pack2<CT>(v[q * 8 + 4], v[q * 8 + 5]);
                o.w = pack2<CT>(v[q * 8 + 6], v[q * 8 + 7]);
                cp[q] = o;
            }
        }
    }
}

// ---- "split8" operands (round 6; common.hpp): the 128 x 128 tile with the K loop in three segments - fp16 MFMA on the leading terms,
// block-scaled fp8 MFMA (one 16x16x128 per accumulator and 128-byte K-tile: the same two 16-byte chunks per lane the fp16 k-steps read,
// taken together as 32 e4m3 values; both operands use the same chunk order, and a sum over k does not care which lane group carries a
// value) on the two correction terms.  Every accumulator sees the same sequence of operations as in the 256 x 256 kernel: the
// dispatcher's tile choice never changes a bit.  OSPL: the output is written as split8 rows of act(acc) (the next GEMM's operand: fc1 -> fc2).
template <bool OSPL>
__global__ __launch_bounds__(256, 2) void gemm_split8_kernel(const GemmArgs a) {
    using X8 = f16x8;
    __shared__ __attribute__((aligned(16))) char smem[4 * kTileBytes];  // [buf][A|W]
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave & 1, wn = wave >> 1;
    const int r15 = lane & 15, g = lane >> 4;
    const int nblk = gridDim.x;
    int id = xcd_remap(blockIdx.x, nblk);
    const int per_batch = a.tiles_m * a.tiles_n;
    const int z = id / per_batch;
    id -= z * per_batch;
    const int tile_m = id / a.tiles_n, tile_n = id - tile_m * a.tiles_n;
    const int64_t m0 = (int64_t)tile_m * BM;
    const int n0 = tile_n * BN;
    const _Float16* A = reinterpret_cast<const _Float16*>(a.A) + z * a.sA;
    const _Float16* W = reinterpret_cast<const _Float16*>(a.W) + z * a.sW;
    const int srow = lane >> 3;
    const int schunk = (lane & 7) ^ srow;
    const _Float16* a_src[4];
    const _Float16* w_src[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int lrow = (wave * 4 + j) * 8 + srow;
        int64_t gm = m0 + lrow;
        gm = gm < a.M ? gm : a.M - 1;
        a_src[j] = A + gm * a.lda + schunk * 8;
        const int perm = (lrow & 64) + ((lrow & 15) >> 2) * 16 + ((lrow >> 4) & 3) * 4 + (lrow & 3);
        int gn = n0 + perm;
        gn = gn < a.N ? gn : a.N - 1;
        w_src[j] = W + (int64_t)gn * a.ldw + schunk * 8;
    }
    auto stage = [&](int kt, int buf) {
        char* base = smem + buf * 2 * kTileBytes + wave * 4096;
        const int koff = kt * 64;                                   // 128 bytes per K-tile whatever it holds
#pragma unroll
        for (int j = 0; j < 4; ++j) __builtin_amdgcn_global_load_lds((gptr_t)(a_src[j] + koff), (lptr_t)(base + j * 1024), 16, 0, 0);
#pragma unroll
        for (int j = 0; j < 4; ++j) __builtin_amdgcn_global_load_lds((gptr_t)(w_src[j] + koff), (lptr_t)(base + kTileBytes + j * 1024), 16, 0, 0);
    };
    f32x4 acc[4][4];
    {
        const int nb_ = n0 + wn * 64 + g * 16;
        float bias_[16];
        if (a.bias != nullptr && nb_ + 16 <= a.N) {
            const float4* bp = reinterpret_cast<const float4*>(a.bias + z * a.sBias + nb_);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4 b4 = bp[q];
                bias_[q * 4 + 0] = b4.x; bias_[q * 4 + 1] = b4.y; bias_[q * 4 + 2] = b4.z; bias_[q * 4 + 3] = b4.w;
            }
        } else {
#pragma unroll
            for (int q = 0; q < 16; ++q) bias_[q] = 0.f;
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{bias_[j * 4 + 0], bias_[j * 4 + 1], bias_[j * 4 + 2], bias_[j * 4 + 3]};
    }
    const int a_row_off = (wm * 64 + r15) * 128;
    const int w_row_off = (wn * 64 + r15) * 128;
    const int swz = r15 & 7;
    const int c0 = ((0 + g) ^ swz) << 4, c1 = ((4 + g) ^ swz) << 4;
    const int nk = a.K >> 6;
    const int k_seg2 = a.k16 + (a.k16 >> 1);
    stage(0, 0);
    for (int kt = 0; kt < nk; ++kt) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (explicit: see gemm_kernel)
        __syncthreads();
        if (kt + 1 < nk) stage(kt + 1, (kt + 1) & 1);
        const char* As = smem + (kt & 1) * 2 * kTileBytes;
        const char* Ws = As + kTileBytes;
        X8 af[4][2], wf[4][2];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            af[i][0] = *reinterpret_cast<const X8*>(As + a_row_off + i * 2048 + c0);
            af[i][1] = *reinterpret_cast<const X8*>(As + a_row_off + i * 2048 + c1);
            wf[i][0] = *reinterpret_cast<const X8*>(Ws + w_row_off + i * 2048 + c0);
            wf[i][1] = *reinterpret_cast<const X8*>(Ws + w_row_off + i * 2048 + c1);
        }
        if (kt < a.k16) {
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int mi = 0; mi < 4; ++mi)
#pragma unroll
                    for (int ni = 0; ni < 4; ++ni) acc[mi][ni] = Elem<_Float16>::mfma16(wf[ni][ks], af[mi][ks], acc[mi][ni]);
        } else {
            const int sc_a = kt < k_seg2 ? a.sc_a1 : a.sc_a2, sc_w = kt < k_seg2 ? a.sc_w1 : a.sc_w2;
            i32x8 a8[4], w8[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                a8[i] = __builtin_bit_cast(i32x8, __builtin_shufflevector(af[i][0], af[i][1], 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15));
                w8[i] = __builtin_bit_cast(i32x8, __builtin_shufflevector(wf[i][0], wf[i][1], 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15));
            }
#pragma unroll
            for (int mi = 0; mi < 4; ++mi)
#pragma unroll
                for (int ni = 0; ni < 4; ++ni)
                    acc[mi][ni] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(w8[ni], a8[mi], acc[mi][ni], 0, 0, 0, sc_w, 0, sc_a);
        }
    }
    const int nb = n0 + wn * 64 + g * 16;
    if (nb + 16 > a.N) return;
    const bool has_res = a.R != nullptr;
#pragma unroll
    for (int mi = 0; mi < 4; ++mi) {
        const int64_t m = m0 + wm * 64 + mi * 16 + r15;
        if (m >= a.M) continue;
        float v[16];
#pragma unroll
        for (int ni = 0; ni < 4; ++ni)
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) v[ni * 4 + jj] = acc[mi][ni][jj];
        if (a.act == CIR_ACT_GELU) {
#pragma unroll
            for (int q = 0; q < 16; q += 8) {      // the 256 x 256 kernel's packed evaluation: the same operations, the same bits
                float w8[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) w8[e] = v[q + e];
                gelu_erf_as8(w8);
#pragma unroll
                for (int e = 0; e < 8; ++e) v[q + e] = w8[e];
            }
        } else if (a.act == CIR_ACT_RELU) {
#pragma unroll
            for (int q = 0; q < 16; ++q) v[q] = fmaxf(v[q], 0.f);
        }
        if constexpr (!OSPL) {
            if (has_res) {
                const float4* rp = reinterpret_cast<const float4*>(reinterpret_cast<const float*>(a.R) + z * a.sR + m * a.ldr + nb);
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float4 r4 = rp[q];
                    v[q * 4 + 0] += r4.x; v[q * 4 + 1] += r4.y; v[q * 4 + 2] += r4.z; v[q * 4 + 3] += r4.w;
                }
            }
            float4* cp = reinterpret_cast<float4*>(reinterpret_cast<float*>(a.C) + z * a.sC + m * a.ldc + nb);
#pragma unroll
            for (int q = 0; q < 4; ++q) cp[q] = make_float4(v[q * 4], v[q * 4 + 1], v[q * 4 + 2], v[q * 4 + 3]);
        } else {
            char* row = reinterpret_cast<char*>(a.C) + z * a.sC + m * a.ldc;       // bytes
            const float q0[4] = {v[0], v[1], v[2], v[3]}, q1[4] = {v[4], v[5], v[6], v[7]}, q2[4] = {v[8], v[9], v[10], v[11]}, q3[4] = {v[12], v[13], v[14], v[15]};
            const Split4 s0 = split8_x4(q0), s1 = split8_x4(q1), s2 = split8_x4(q2), s3 = split8_x4(q3);
            const u32x4 h0 = {s0.h01, s0.h23, s1.h01, s1.h23}, h1 = {s2.h01, s2.h23, s3.h01, s3.h23};
            const u32x4 l8 = {s0.lo8, s1.lo8, s2.lo8, s3.lo8}, h8 = {s0.hi8, s1.hi8, s2.hi8, s3.hi8};
            *reinterpret_cast<u32x4*>(row + 2 * nb) = h0;
            *reinterpret_cast<u32x4*>(row + 2 * nb + 16) = h1;
            *reinterpret_cast<u32x4*>(row + 2 * (int64_t)a.n_logical + nb) = l8;
            *reinterpret_cast<u32x4*>(row + 3 * (int64_t)a.n_logical + nb) = h8;
        }
    }
}

}  // namespace cir

static inline int e8m0_word(int exp2) { const int b = (127 + exp2) & 0xff; return b * 0x01010101; }

extern "C" int cir_gemm_split8(const void* A, int64_t lda_bytes, int64_t strideA_bytes, const void* W, int64_t ldw_bytes, int64_t strideW_bytes,
                               const float* bias, int64_t strideBias, const float* residual, int64_t ldr, int64_t strideR,
                               void* C, int64_t ldc, int64_t strideC, int64_t M, int N, int K, int batch, int act, int out_split,
                               int w_exp_hi, int w_exp_lo, void* stream) {
    using namespace cir;
    CIR_CHECK_PTR(A); CIR_CHECK_PTR(W); CIR_CHECK_PTR(C);
    if (M <= 0 || N <= 0 || K <= 0 || batch <= 0) return CIR_EINVAL;
    if (K % 256 != 0 || N % 16 != 0) return CIR_ESHAPE;                        // whole K-tile pairs in each of the three segments
    if (out_split != 0 && out_split != 1) return CIR_EINVAL;
    if (act < CIR_ACT_NONE || act > CIR_ACT_RELU) return CIR_EINVAL;
    if (out_split && (residual || N % 64 != 0)) return CIR_EINVAL;            // split8 rows out: a linear-then-activation site (fc1), no residual
    if (residual && act != CIR_ACT_NONE) return CIR_EINVAL;
    if (w_exp_hi < -100 || w_exp_hi > 100 || w_exp_lo < -100 || w_exp_lo > 100) return CIR_EINVAL;
    if (lda_bytes < 4 * (int64_t)K || ldw_bytes < 4 * (int64_t)K) return CIR_ESHAPE;
    if (!cir_aligned16(A) || !cir_aligned16(W) || !cir_aligned16(C) || lda_bytes % 16 || ldw_bytes % 16 || strideA_bytes % 16 || strideW_bytes % 16) return CIR_EALIGN;
    if (out_split ? (ldc % 16 || strideC % 16 || ldc < 4 * (int64_t)N) : (ldc % 4 || strideC % 4)) return CIR_EALIGN;
    if (bias && (!cir_aligned16(bias) || strideBias % 4)) return CIR_EALIGN;
    if (residual && (!cir_aligned16(residual) || ldr % 4 || strideR % 4)) return CIR_EALIGN;
    GemmArgs a;
    a.A = A; a.lda = lda_bytes / 2; a.sA = strideA_bytes / 2;
    a.W = W; a.ldw = ldw_bytes / 2; a.sW = strideW_bytes / 2;
    a.bias = bias; a.sBias = strideBias;
    a.R = residual; a.ldr = ldr; a.sR = strideR;
    a.C = C; a.ldc = ldc; a.sC = strideC;
    a.M = M; a.N = N; a.K = 2 * K; a.batch = batch; a.act = act; a.group_w = 1; a.dbg = 0;
    a.k16 = K / 64;
    a.sc_a1 = e8m0_word(-kSplitLoExp); a.sc_w1 = e8m0_word(-w_exp_hi);
    a.sc_a2 = e8m0_word(0); a.sc_w2 = e8m0_word(-w_exp_lo);
    a.n_logical = N;
    a.tiles_m = (int)((M + BM - 1) / BM);
    a.tiles_n = (N + BN - 1) / BN;
    const int64_t nblk = (int64_t)a.tiles_m * a.tiles_n * batch;
    if (nblk > 0x7fffffff) return CIR_ESHAPE;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const int64_t nblk256 = ((M + 255) / 256) * ((N + 255) / 256) * batch;
    bool use256 = N >= 256 && nblk256 >= 192 && a.lda < (1 << 21) && a.ldw < (1 << 21);
    if (g_tune[CIR_TUNE_GEMM_TILE] == 128) use256 = false;
    else if (g_tune[CIR_TUNE_GEMM_TILE] == 256) use256 = a.lda < (1 << 21) && a.ldw < (1 << 21);
    if (use256) {
        launch_gemm256_split8(a, out_split, s);
        CIR_LAUNCH_RESULT();
    }
    dim3 grid((unsigned)nblk), block(256);
    if (out_split) hipLaunchKernelGGL((gemm_split8_kernel<true>), grid, block, 0, s, a);
    else hipLaunchKernelGGL((gemm_split8_kernel<false>), grid, block, 0, s, a);
    CIR_LAUNCH_RESULT();
}

extern "C" int cir_gemm_bias_act(const void* A, int64_t lda, int64_t strideA, const void* W, int64_t ldw, int64_t strideW,
                                 const float* bias, int64_t strideBias, const void* residual, int res_dtype, int64_t ldr,
                                 int64_t strideR, void* C, int64_t ldc, int64_t strideC, int64_t M, int N, int K, int batch,
                                 int act, int in_dtype, int out_dtype, void* stream) {
    using namespace cir;
    CIR_CHECK_PTR(A); CIR_CHECK_PTR(W); CIR_CHECK_PTR(C);
    if (M <= 0 || N <= 0 || K <= 0 || batch <= 0) return CIR_EINVAL;
    if (in_dtype == CIR_F32) {
        // "exact" mode (ABI v11): fp32 operands, fp32 C, fp32 residual, exact-erf GELU - v_mfma_f32_16x16x4_f32 on the 128 x 128 tile
        if (K % 32 != 0 || N % 16 != 0) return CIR_ESHAPE;
        if (out_dtype != CIR_F32 || (residual && res_dtype != CIR_F32)) return CIR_EDTYPE;
        if (act < CIR_ACT_NONE || act > CIR_ACT_RELU) return CIR_EINVAL;
        if (!cir_aligned16(A) || !cir_aligned16(W) || !cir_aligned16(C) || lda % 4 || ldw % 4 || strideA % 4 || strideW % 4 || ldc % 4 || strideC % 4)
            return CIR_EALIGN;
        if (bias && (!cir_aligned16(bias) || strideBias % 4)) return CIR_EALIGN;
        if (residual && (!cir_aligned16(residual) || ldr % 4 || strideR % 4)) return CIR_EALIGN;
        GemmArgs a;
        a.A = A; a.lda = lda; a.sA = strideA;
        a.W = W; a.ldw = ldw; a.sW = strideW;
        a.bias = bias; a.sBias = strideBias;
        a.R = residual; a.ldr = ldr; a.sR = strideR;
        a.C = C; a.ldc = ldc; a.sC = strideC;
        a.M = M; a.N = N; a.K = K; a.batch = batch; a.act = act; a.group_w = 1; a.dbg = 0;
        a.tiles_m = (int)((M + BM - 1) / BM);
        a.tiles_n = (N + BN - 1) / BN;
        const int64_t nblk = (int64_t)a.tiles_m * a.tiles_n * batch;
        if (nblk > 0x7fffffff) return CIR_ESHAPE;
        hipLaunchKernelGGL((gemm_kernel<float, 1>), dim3((unsigned)nblk), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), a);
        CIR_LAUNCH_RESULT();
    }
    if (K % BK != 0 || N % 16 != 0) return CIR_ESHAPE;
    if (in_dtype != CIR_BF16 && in_dtype != CIR_F16) return CIR_EDTYPE;
    if (out_dtype != in_dtype && out_dtype != CIR_F32 && out_dtype != CIR_F16) return CIR_EDTYPE;
    if (act < CIR_ACT_NONE || act > CIR_ACT_RELU) return CIR_EINVAL;
    // residual: fp32 (with any C) or fp16 (only as part of an fp16 C: the 16-bit residual stream)
    if (residual && res_dtype != CIR_F32 && !(res_dtype == CIR_F16 && out_dtype == CIR_F16)) return CIR_EDTYPE;
    // an fp16 C from bf16 operands is the residual STREAM: its epilogue reads the residual as fp16 (an fp32 residual there
    // would be reinterpreted, not converted)
    if (residual && out_dtype == CIR_F16 && in_dtype != CIR_F16 && res_dtype != CIR_F16) return CIR_EDTYPE;
    const int64_t out_elems_per16 = out_dtype == CIR_F32 ? 4 : 8;
    if (!cir_aligned16(A) || !cir_aligned16(W) || !cir_aligned16(C) || lda % 8 || ldw % 8 || strideA % 8 || strideW % 8 ||
        ldc % out_elems_per16 || strideC % out_elems_per16)
        return CIR_EALIGN;
    if (bias && (!cir_aligned16(bias) || strideBias % 4)) return CIR_EALIGN;
    const int64_t res_elems_per16 = res_dtype == CIR_F32 ? 4 : 8;
    if (residual && (!cir_aligned16(residual) || ldr % res_elems_per16 || strideR % res_elems_per16)) return CIR_EALIGN;

    GemmArgs a;
    a.A = A; a.lda = lda; a.sA = strideA;
    a.W = W; a.ldw = ldw; a.sW = strideW;
    a.bias = bias; a.sBias = strideBias;
    a.R = residual; a.ldr = ldr; a.sR = strideR;
    a.C = C; a.ldc = ldc; a.sC = strideC;
    a.M = M; a.N = N; a.K = K; a.batch = batch; a.act = act; a.group_w = 1;
    a.tiles_m = (int)((M + BM - 1) / BM);
    a.tiles_n = (N + BN - 1) / BN;
    const int64_t nblk = (int64_t)a.tiles_m * a.tiles_n * batch;
    if (nblk > 0x7fffffff) return CIR_ESHAPE;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    // epilogue kind: 1 = fp32 C (fp32 residual), 2 = fp16 stream C with an fp16 residual or from bf16 operands,
    // 0 = C in the operand type (an fp32 residual is then served by the 128 x 128 kernel only)
    const bool stream16 = out_dtype == CIR_F16 && (in_dtype != CIR_F16 || (residual && res_dtype == CIR_F16));
    const int out_kind = out_dtype == CIR_F32 ? 1 : (stream16 ? 2 : 0);
    // Tile choice: the 256x256 8-phase kernel needs about a full wave of workgroups (256 CUs) to pay;
    // small problems keep the 128x128 kernel (more, smaller tiles).  cir_set_tuning(CIR_TUNE_GEMM_TILE, 128|256) forces one.
    const int64_t nblk256 = ((M + 255) / 256) * ((N + 255) / 256) * batch;
    // (the 256 kernel adds the residual in the row layout of its fp32-layout epilogues: linear epilogues only; 32-bit operand offsets)
    bool use256 = N >= 256 && nblk256 >= 192;
    const bool can256 = !(residual && (act != CIR_ACT_NONE || out_kind == 0)) && K % 128 == 0 && lda < (1 << 21) && ldw < (1 << 21);   // K-tile pairs; tile-relative 32-bit offsets
    if (g_tune[CIR_TUNE_GEMM_TILE] == 128) use256 = false;
    else if (g_tune[CIR_TUNE_GEMM_TILE] == 256) use256 = true;
    if (use256 && can256) {
        launch_gemm256(a, in_dtype, out_kind, s);
        CIR_LAUNCH_RESULT();
    }
    dim3 grid((unsigned)nblk), block(256);
#define CIR_LAUNCH128(TT) \
    do { if (out_kind == 1) hipLaunchKernelGGL((gemm_kernel<TT, 1>), grid, block, 0, s, a); \
         else if (out_kind == 2) hipLaunchKernelGGL((gemm_kernel<TT, 2>), grid, block, 0, s, a); \
         else hipLaunchKernelGGL((gemm_kernel<TT, 0>), grid, block, 0, s, a); } while (0)
    if (in_dtype == CIR_BF16) CIR_LAUNCH128(__bf16);
    else CIR_LAUNCH128(_Float16);
#undef CIR_LAUNCH128
    CIR_LAUNCH_RESULT();
}

extern "C" int cir_gemm_ln_bias_act(const void* X, int64_t ldx, const void* Wg, int64_t ldw, const float* colsum, const float* bias,
                                    void* C, int64_t ldc, int64_t M, int N, int K, float eps, int act, int dtype, void* stream) {
    using namespace cir;
    CIR_CHECK_PTR(X); CIR_CHECK_PTR(Wg); CIR_CHECK_PTR(colsum); CIR_CHECK_PTR(bias); CIR_CHECK_PTR(C);
    if (M <= 0 || N <= 0 || K <= 0 || !(eps > 0.f)) return CIR_EINVAL;
    if (dtype != CIR_F16) return CIR_EDTYPE;                 // the rows ARE the fp16 residual stream: no bf16 / fp32 form
    if (act != CIR_ACT_NONE && act != CIR_ACT_GELU) return CIR_EINVAL;
    if (K % 128 != 0 || N % 16 != 0 || N < 64) return CIR_ESHAPE;   // K-tile pairs of the 256 x 256 kernel; whole 64-feature wave columns
    if (ldx >= (1 << 21) || ldw >= (1 << 21)) return CIR_ESHAPE;
    if (!cir_aligned16(X) || !cir_aligned16(Wg) || !cir_aligned16(C) || !cir_aligned16(colsum) || !cir_aligned16(bias) || ldx % 8 || ldw % 8 || ldc % 8)
        return CIR_EALIGN;
    GemmArgs a;
    a.A = X; a.lda = ldx; a.sA = 0;
    a.W = Wg; a.ldw = ldw; a.sW = 0;
    a.bias = bias; a.sBias = 0;
    a.R = nullptr; a.ldr = 0; a.sR = 0;
    a.C = C; a.ldc = ldc; a.sC = 0;
    a.M = M; a.N = N; a.K = K; a.batch = 1; a.act = act; a.group_w = 1; a.dbg = 0;
    a.colsum = colsum; a.ln_eps = eps;
    if (((M + 255) / 256) * (int64_t)((N + 255) / 256) > 0x7fffffff) return CIR_ESHAPE;
    launch_gemm256(a, CIR_F16, 3, reinterpret_cast<hipStream_t>(stream));
    CIR_LAUNCH_RESULT();
}
