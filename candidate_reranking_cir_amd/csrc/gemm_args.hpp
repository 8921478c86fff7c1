// Shared GEMM argument block and launcher declarations (gemm.hip = 128x128 tiles, gemm256.hip = 256x256 tiles).
#pragma once
#include "common.hpp"

namespace cir {

struct GemmArgs {
    const void* A; int64_t lda, sA;
    const void* W; int64_t ldw, sW;
    const float* bias; int64_t sBias;
    const void* R; int64_t ldr, sR;    // residual: fp32 or fp16 (the element type of C in the fp32-layout epilogues)
    void* C; int64_t ldc, sC;
    int64_t M; int N, K, batch, act, tiles_m, tiles_n;
    int group_w;   // gemm256: tiles are walked in column groups of this many n-panels (weights stay L2-resident)
};

// four consecutive elements of a row of C / R in the fp32-layout epilogues
template <typename ST> struct RowVec;
template <> struct RowVec<float> {
    using type = float4;
    static __device__ __forceinline__ float4 zero() { return make_float4(0.f, 0.f, 0.f, 0.f); }
    static __device__ __forceinline__ f32x4 to_f32(const float4& r) { return f32x4{r.x, r.y, r.z, r.w}; }
};
template <> struct RowVec<_Float16> {
    using type = u32x2;
    static __device__ __forceinline__ u32x2 zero() { return u32x2{0u, 0u}; }
    static __device__ __forceinline__ f32x4 to_f32(const u32x2& r) {
        typedef __attribute__((ext_vector_type(2))) _Float16 h2;
        const unsigned a = r[0], b = r[1];   // (indexing, not .x / .y: hipcc 7.2 folds bit_cast(r.y) of a vector reference onto r.x)
        const h2 lo = __builtin_bit_cast(h2, a);
        const h2 hi = __builtin_bit_cast(h2, b);
        return f32x4{(float)lo[0], (float)lo[1], (float)hi[0], (float)hi[1]};
    }
};

typedef __attribute__((address_space(1))) const void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

// 256x256x64 tiles, 8 waves, staggered 8-phase schedule (gemm256.hip)
void launch_gemm256(const GemmArgs& a, int in_dtype, int out_kind, hipStream_t s);

}  // namespace cir
