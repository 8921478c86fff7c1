// cir_cross_attention_folded for 225 .. 608 image tokens (the reference's 384-px geometry: 577 tokens; validate_stage2.py:327) - the query-side
// fold of xattn_fold.hip with ONE 16-row block per wave: the transposed score tile S^T of a block against 608 keys is 38 accumulator tiles = 152
// registers (three blocks, as in the 224-key kernel, would need 456).  A workgroup (8 waves) covers 4 heads x 32 tokens = 128 stacked query rows;
// three workgroups per (candidate, branch).  Same four products and the same accumulator-as-operand chaining as xattn_fold.hip (G1 Q'^T = W_k^T q^T,
// G2 S^T += X Q'^T, softmax, G3 C'^T = X^T P^T, G4 ctx^T += W_v C'^T); a unit is 32 features, and X is staged per UNIT (608 keys x 32 features =
// 38 KB, two buffers): 64-byte rows with the chunk position XOR-swizzled by f(row) = {0, 2, 3, 1}[(row >> 2) & 3] for phase 1's ds_read_b128
// (the four 16-lane groups of a b128 read then hit 64 distinct banks), 96-byte rows for phase 2's transposing reads (8 rows x 8 dwords at a
// 24-dword stride: distinct multiples of 8).  Weight fragments: the same per-fragment packing (ops.fold_pack_key / fold_pack_value).
// Per (candidate, branch): 1.50 GFLOP instead of 2.83 (K|V GEMM 2.72 + attention 0.11).

#include "xattn_fold.hpp"

namespace cir {

constexpr int kKB16 = 38;                    // 16-key blocks: 608 keys
constexpr int kBuf16 = 4096 * 16;            // one X-unit buffer: phase 2 needs 608 rows x 6 slots = 3648 slots (8 per wave-instruction row of 512)
constexpr int kStrideQ16 = 544;              // q rows in LDS: 4 heads x 64 x 2 B + 32 (136 dwords = 8 mod 64: conflict-free b128 fragment reads)
constexpr int kStrideP2 = 96;                // phase 2's row stride

template <typename T, bool MASKED = false>
__global__ __launch_bounds__(512, 2) void xattn_fold16_kernel(const FoldArgs a) {
    using X8 = typename Elem<T>::x8;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l16 = lane & 15, g = lane >> 4;
    // XCDs 0-3 run branch 0, XCDs 4-7 branch 1 (see xattn_fold.hip); within a branch: (candidate, head group of 4)
    const int b = (blockIdx.x >> 2) & 1;
    const int idx = (blockIdx.x >> 3) * 4 + (blockIdx.x & 3);
    const int t = idx / 3, hg = idx - 3 * t;
    if (t >= a.T) return;

    const T* X = reinterpret_cast<const T*>(a.x) + (int64_t)t * a.x_s1;
    const int head = 4 * hg + (wave >> 1);                       // this wave's head; its 16 token rows: 16 (wave & 1) + l16
    const int tok = 16 * (wave & 1) + l16;

    // q of this workgroup's 4 heads in LDS: 32 token rows x 256 values
    char* const qs = smem + 2 * kBuf16;
    {
        const T* qb_ = reinterpret_cast<const T*>(a.q) + (int64_t)b * a.q_sb + (int64_t)t * a.L * a.q_rs + hg * 256;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int c = tid + i * 512;                          // 1024 16-byte pieces: 32 rows x 32
            const int row = c >> 5, ch = c & 31;
            X8 v;
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = static_cast<T>(0.f);
            if (row < a.L) v = *reinterpret_cast<const X8*>(qb_ + (int64_t)row * a.q_rs + ch * 8);
            *reinterpret_cast<X8*>(qs + row * kStrideQ16 + ch * 16) = v;
        }
    }
    const int qoff = tok * kStrideQ16 + ((wave >> 1) * 64 + 8 * g) * 2;

    const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(X), 0, a.N * kFoldD * 2, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_k = __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(reinterpret_cast<const T*>(a.wkt) + (int64_t)b * a.w_sb), 0, kFoldD * kFoldD * 2, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_v = __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(reinterpret_cast<const T*>(a.wvp) + (int64_t)b * a.w_sb), 0, kFoldD * kFoldD * 2, 0x00020000);
    const int wlane = lane * 16;

    // phase-1 staging: 5 pieces per wave (2560 slots of 16 B: 608 rows x 4 + slack); slot s = row s >> 2, position s & 3 holds chunk (s & 3) ^ f(row)
    int xoff1[5];
#pragma unroll
    for (int j = 0; j < 5; ++j) {
        const int s = (wave * 5 + j) * 64 + lane;
        const int row = s >> 2;
        const int f = (0x1E >> (2 * ((row >> 2) & 3))) & 3;      // {0, 2, 3, 1}[(row >> 2) & 3] packed two bits each: 0b00_01_11_10 -> 0x1E
        xoff1[j] = (min(row, a.N - 1) * kFoldD + (((s & 3) ^ f) * 8)) * 2;
    }

    // ---------------------------------------------------------------- phase 1 ----------------------------------------------------------
    f32x4 S[kKB16];
#pragma unroll
    for (int kb = 0; kb < kKB16; ++kb) S[kb] = f32x4{0.f, 0.f, 0.f, 0.f};
    X8 w[2][2];                                                   // [interleaved 16-feature tile][k-step]
    auto load_wk = [&](int unit) {
#pragma unroll
        for (int fbh = 0; fbh < 2; ++fbh)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) w[fbh][ks] = wload<X8>(rs_k, wlane, (((unit * 2 + fbh) * 12 + head) * 2 + ks) * 1024);
    };
    {
        char* base = smem + wave * 5 * 1024;
#pragma unroll
        for (int j = 0; j < 5; ++j) FOLD_DMA(rs_x, xoff1[j], 0, base + j * 1024);
    }
    load_wk(0);
    const int fr = (0x1E >> (2 * ((l16 >> 2) & 3))) & 3;         // this lane's read swizzle: row = 16 kb + l16 -> f depends on (l16 >> 2) only
    for (int u = 0; u < 24; ++u) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        const char* xs = smem + (u & 1) * kBuf16;
        // G1: Q'^T tiles of features [32 u, 32 u + 32), interleaved (tile t row 4 g' + r = feature 8 g' + 4 t + r)
        f32x4 a1[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const X8 qf = *reinterpret_cast<const X8*>(qs + qoff + 64 * ks);
#pragma unroll
            for (int fbh = 0; fbh < 2; ++fbh) a1[fbh] = Elem<T>::mfma16(w[fbh][ks], qf, a1[fbh]);
        }
        const X8 bq = pack_acc2<T>(a1[0], a1[1]);
        __builtin_amdgcn_sched_barrier(0);
        const bool more = u + 1 < 24;
        auto mem_op = [&](int j) {                                // next unit's 4 weight fragments, then its 5 DMA pieces: one request per key block
            if (!more) return;
            if (j < 4) w[j >> 1][j & 1] = wload<X8>(rs_k, wlane, ((((u + 1) * 2 + (j >> 1)) * 12 + head) * 2 + (j & 1)) * 1024);
            else if (j < 9) FOLD_DMA(rs_x, xoff1[j - 4], (u + 1) * 64, smem + ((u + 1) & 1) * kBuf16 + (wave * 5 + j - 4) * 1024);
        };
        // G2: k-slot (g, j) = feature 32 u + 8 g + j: piece g of the key's 64-byte row, at position g ^ f(row)
        const char* xr = xs + l16 * 64 + ((g ^ fr) << 4);
        auto rd = [&](int kb) { return *reinterpret_cast<const X8*>(xr + kb * 16 * 64); };
        X8 xa[3];
        xa[0] = rd(0);
        xa[1] = rd(1);
#pragma unroll
        for (int kb = 0; kb < kKB16; ++kb) {
            if (kb + 2 < kKB16) xa[(kb + 2) % 3] = rd(kb + 2);
            mem_op(kb);
            __builtin_amdgcn_sched_barrier(0);
            S[kb] = Elem<T>::mfma16(xa[kb % 3], bq, S[kb]);
            __builtin_amdgcn_sched_barrier(0);
        }
    }

    // ---------------------------------------------------------------- softmax ----------------------------------------------------------
    const float sl = a.scale * 1.4426950408889634f;
    float m = -INFINITY;
#pragma unroll
    for (int kb = 0; kb < kKB16; ++kb)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int key = 16 * kb + 4 * g + r;
            float v = key < a.N ? S[kb][r] : -INFINITY;
            if constexpr (MASKED) {   // (as xattn_fold_kernel: log2-domain logits incl. the additive key mask)
                const float mk = a.mask[(int64_t)t * a.m_st + min(key, a.N - 1)];
                v = key < a.N ? fmaf(fmaxf(mk, -2.0e38f), 1.4426950408889634f, S[kb][r] * sl) : -INFINITY;
            }
            S[kb][r] = v;
            m = fmaxf(m, v);
        }
    m = fmaxf(m, __shfl_xor(m, 16, 64));
    m = fmaxf(m, __shfl_xor(m, 32, 64));
    const float ms = MASKED ? m : m * sl;
    float sum = 0.f;
#pragma unroll
    for (int kb = 0; kb < kKB16; ++kb)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float p = MASKED ? __builtin_amdgcn_exp2f(S[kb][r] - ms) : __builtin_amdgcn_exp2f(fmaf(S[kb][r], sl, -ms));
            S[kb][r] = p;
            sum += p;
        }
    sum += __shfl_xor(sum, 16, 64);
    sum += __shfl_xor(sum, 32, 64);
    const float rinv = 1.0f / sum;
    X8 P[kKB16 / 2];
#pragma unroll
    for (int p = 0; p < kKB16 / 2; ++p) P[p] = pack_acc2<T>(S[2 * p], S[2 * p + 1]);

    // ---------------------------------------------------------------- phase 2 ----------------------------------------------------------
    f32x4 c4[4];
#pragma unroll
    for (int db = 0; db < 4; ++db) c4[db] = f32x4{0.f, 0.f, 0.f, 0.f};
    X8 wv[4];
    // staging: 8 pieces per wave (4096 slots): slot s = row s / 6, position s % 6 (positions 4, 5 are padding)
    int xoff2[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int s = (wave * 8 + j) * 64 + lane;
        const int row = s / 6, c = s - 6 * row;
        xoff2[j] = (min(row, a.N - 1) * kFoldD + min(c, 3) * 8) * 2;
    }
    __syncthreads();                                              // every wave is done with phase 1's buffers
    {
        char* base = smem + wave * 8 * 1024;
#pragma unroll
        for (int j = 0; j < 8; ++j) FOLD_DMA(rs_x, xoff2[j], 0, base + j * 1024);
    }
    const int troff = (4 * g + (l16 >> 2)) * kStrideP2 + (4 * (l16 & 3)) * 2;
    for (int u = 0; u < 24; ++u) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        const bool more = u + 1 < 24;
        auto mem_op = [&](int j) {                                // this unit's 4 W_v fragments (needed by G4 at its end), then the next unit's 8 DMA pieces
            if (j < 4) wv[j] = wload<X8>(rs_v, wlane, ((u * 4 + j) * 12 + head) * 1024);
            else if (j < 12 && more) FOLD_DMA(rs_x, xoff2[j - 4], (u + 1) * 64, smem + ((u + 1) & 1) * kBuf16 + (wave * 8 + j - 4) * 1024);
        };
        const unsigned xaddr = (unsigned)(size_t)(lptr_t)(smem + (u & 1) * kBuf16 + troff);
        f32x4 a3[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
        u32x2 lo[3], hi[3];
#define FOLD16_TR(SLOT, I)                                                                                                      \
        asm volatile("ds_read_b64_tr_b16 %0, %2 offset:%3\n\tds_read_b64_tr_b16 %1, %2 offset:%4"                                 \
                     : "=&v"(lo[SLOT]), "=&v"(hi[SLOT]) : "v"(xaddr), "n"((32 * ((I) >> 1)) * kStrideP2 + (16 * ((I) & 1)) * 2),     \
                       "n"((32 * ((I) >> 1) + 16) * kStrideP2 + (16 * ((I) & 1)) * 2) : "memory")
        FOLD16_TR(0, 0);
        FOLD16_TR(1, 1);
        // (expanded by macro, not by `#pragma unroll`: the asm immediates need the step index as a literal, and with a 38-case switch inside
        //  the loop hipcc gave up unrolling - P and the fragment registers then lived in scratch)
#define FOLD16_STEP(I)                                                                                                            \
        {                                                                                                                          \
            if ((I) + 2 < kKB16) FOLD16_TR(((I) + 2) % 3, (I) + 2 < kKB16 ? (I) + 2 : 0);                                           \
            mem_op(I);                                                                                                             \
            if ((I) + 2 < kKB16) asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(lo[(I) % 3]), "+v"(hi[(I) % 3]));                       \
            else if ((I) + 1 < kKB16) asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(lo[(I) % 3]), "+v"(hi[(I) % 3]));                  \
            else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(lo[(I) % 3]), "+v"(hi[(I) % 3]));                                       \
            __builtin_amdgcn_sched_barrier(0);                                                                                     \
            const u32x4 both = {lo[(I) % 3].x, lo[(I) % 3].y, hi[(I) % 3].x, hi[(I) % 3].y};                                        \
            a3[(I) & 1] = Elem<T>::mfma16(__builtin_bit_cast(X8, both), P[(I) >> 1], a3[(I) & 1]);                                  \
            __builtin_amdgcn_sched_barrier(0);                                                                                     \
        }
#define FOLD16_STEP4(I) FOLD16_STEP(I) FOLD16_STEP((I) + 1) FOLD16_STEP((I) + 2) FOLD16_STEP((I) + 3)
        FOLD16_STEP4(0) FOLD16_STEP4(4) FOLD16_STEP4(8) FOLD16_STEP4(12) FOLD16_STEP4(16) FOLD16_STEP4(20) FOLD16_STEP4(24) FOLD16_STEP4(28) FOLD16_STEP4(32)
        FOLD16_STEP(36) FOLD16_STEP(37)
#undef FOLD16_STEP4
#undef FOLD16_STEP
#undef FOLD16_TR
        // G4: ctx^T += W_v[:, these 32 features (k-slot order)] C'^T
        const X8 bq = pack_acc2<T>(a3[0], a3[1]);
#pragma unroll
        for (int db = 0; db < 4; ++db) c4[db] = Elem<T>::mfma16(wv[db], bq, c4[db]);
        __builtin_amdgcn_sched_barrier(0);
    }

    // ---------------------------------------------------------------- epilogue ---------------------------------------------------------
    if (tok < a.L) {
        T* orow = reinterpret_cast<T*>(a.out) + (int64_t)t * a.o_st + (int64_t)b * a.o_sb + (int64_t)tok * a.o_sr + head * 64 + 4 * g;
        const float* brow = a.bv + b * kFoldD + head * 64 + 4 * g;
#pragma unroll
        for (int db = 0; db < 4; ++db) {
            const float4 b4 = *reinterpret_cast<const float4*>(brow + 16 * db);
            u32x2 o;
            o.x = pack2<T>(fmaf(c4[db][0], rinv, b4.x), fmaf(c4[db][1], rinv, b4.y));
            o.y = pack2<T>(fmaf(c4[db][2], rinv, b4.z), fmaf(c4[db][3], rinv, b4.w));
            *reinterpret_cast<u32x2*>(orow + 16 * db) = o;
        }
    }
}

int launch_fold16(const FoldArgs& a, int dtype, hipStream_t s) {
    const size_t lds = 2 * kBuf16 + 32 * kStrideQ16;
    const int64_t per_branch = 3 * (int64_t)a.T;
    dim3 grid((unsigned)(8 * ((per_branch + 3) / 4))), block(512);
#define CIR_FOLD16_LAUNCH(TT, MK)                                                                                                                 \
    do {                                                                                                                                          \
        const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&xattn_fold16_kernel<TT, MK>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
        if (e != hipSuccess) return (int)e;                                                                                                       \
        hipLaunchKernelGGL((xattn_fold16_kernel<TT, MK>), grid, block, lds, s, a);                                                                \
    } while (0)
    if (dtype == CIR_BF16) { if (a.mask) CIR_FOLD16_LAUNCH(__bf16, true); else CIR_FOLD16_LAUNCH(__bf16, false); }
    else { if (a.mask) CIR_FOLD16_LAUNCH(_Float16, true); else CIR_FOLD16_LAUNCH(_Float16, false); }
#undef CIR_FOLD16_LAUNCH
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? CIR_OK : (int)e;
}

}  // namespace cir
