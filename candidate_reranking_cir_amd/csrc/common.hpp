// Shared device/host helpers for libcirrank (gfx950 only: wave64, MFMA, LDS-DMA).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "cirrank.h"

namespace cir {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(4))) _Float16 f16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(8))) short s16x8;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;

constexpr int kWave = 64;

// ---- 16-bit element traits -------------------------------------------------------------------
template <typename T> struct Elem;
template <> struct Elem<__bf16> {
    using x8 = bf16x8;
    using x4 = bf16x4;
    static __device__ __forceinline__ f32x4 mfma16(x8 a, x8 b, f32x4 c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
    }
    static __device__ __forceinline__ f32x16 mfma32(x8 a, x8 b, f32x16 c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
    }
    // c += a * b IN PLACE, at this exact spot of the instruction stream (hand-scheduled loops: the builtin lets the
    // register allocator rename accumulators, which costs registers the 256-VGPR kernels do not have)
    static __device__ __forceinline__ void mfma16_acc(f32x4& c, x8 a, x8 b) {
        asm("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b));
    }
};
template <> struct Elem<_Float16> {
    using x8 = f16x8;
    using x4 = f16x4;
    static __device__ __forceinline__ f32x4 mfma16(x8 a, x8 b, f32x4 c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
    }
    static __device__ __forceinline__ f32x16 mfma32(x8 a, x8 b, f32x16 c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
    }
    static __device__ __forceinline__ void mfma16_acc(f32x4& c, x8 a, x8 b) {
        asm("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b));
    }
};

template <typename T> __device__ __forceinline__ T from_f32(float x) { return static_cast<T>(x); }
template <typename T> __device__ __forceinline__ float to_f32(T x) { return static_cast<float>(x); }

// pack two floats into one dword of two 16-bit elements (low = a, high = b)
template <typename T> __device__ __forceinline__ unsigned int pack2(float a, float b) {
    typedef __attribute__((ext_vector_type(2))) T t2;
    t2 v = {static_cast<T>(a), static_cast<T>(b)};
    return __builtin_bit_cast(unsigned int, v);
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// erf with |abs error| <= 1.5e-7 (Abramowitz & Stegun 7.1.26); enough for a 16-bit GELU output.
__device__ __forceinline__ float fast_erf(float x) {
    const float ax = fabsf(x);
    const float t = __frcp_rn(fmaf(0.3275911f, ax, 1.0f));
    float p = fmaf(1.061405429f, t, -1.453152027f);
    p = fmaf(p, t, 1.421413741f);
    p = fmaf(p, t, -0.284496736f);
    p = fmaf(p, t, 0.254829592f);
    p *= t;
    const float e = 1.0f - p * __expf(-ax * ax);
    return copysignf(e, x);
}
__device__ __forceinline__ float gelu_erf_as(float x) { return 0.5f * x * (1.0f + fast_erf(x * 0.70710678118654752f)); }   // (scalar form of gelu_erf_as8: tools / reference for it)

// erf-GELU x * Phi(x) through a logistic fit of the normal CDF, Phi(x) ~ 1 / (1 + 2^-q(x)), q an odd degree-7
// polynomial (minimax fit of x*Phi(x) on [-7, 7], max |error| 1.2e-5 in fp32 - an order of magnitude under the
// 16-bit rounding of the GEMM output it feeds).  8 plain VALU + 2 transcendental ops instead of ~13 + 2: the GELU
// epilogue of the 3072-wide FFN GEMMs is VALU-bound (128 outputs per lane with the MFMA pipe idle).
__device__ __forceinline__ float gelu_erf(float x) {
    const float xc = __builtin_amdgcn_fmed3f(x, -5.5f, 5.5f);   // q is monotone on the clamp range; Phi(+-5.5) = 1/0 to 1e-8
    const float x2 = xc * xc;
    float p = fmaf(x2, -2.48362952e-05f, -7.36062896e-04f);     // coefficients pre-multiplied by log2(e)
    p = fmaf(x2, p, 1.05982735e-01f);
    p = fmaf(x2, p, 2.30164715e+00f);
    const float e = __builtin_amdgcn_exp2f(-(xc * p));
    return x * __builtin_amdgcn_rcpf(1.0f + e);
}

// two GELUs at once on packed fp32 (v_pk_mul / v_pk_fma / v_pk_add): the epilogue has no MFMAs to disturb
typedef __attribute__((ext_vector_type(2))) float f32x2;
__device__ __forceinline__ f32x2 gelu_erf2(f32x2 x) {
    f32x2 xc;
    xc.x = __builtin_amdgcn_fmed3f(x.x, -5.5f, 5.5f);
    xc.y = __builtin_amdgcn_fmed3f(x.y, -5.5f, 5.5f);
    const f32x2 x2 = xc * xc;
    const f32x2 c3 = {-2.48362952e-05f, -2.48362952e-05f}, c2 = {-7.36062896e-04f, -7.36062896e-04f};
    const f32x2 c1 = {1.05982735e-01f, 1.05982735e-01f}, c0 = {2.30164715e+00f, 2.30164715e+00f};
    f32x2 p = __builtin_elementwise_fma(x2, c3, c2);
    p = __builtin_elementwise_fma(x2, p, c1);
    p = __builtin_elementwise_fma(x2, p, c0);
    const f32x2 q = xc * p;
    f32x2 e;
    e.x = __builtin_amdgcn_exp2f(-q.x);
    e.y = __builtin_amdgcn_exp2f(-q.y);
    const f32x2 d = e + 1.0f;
    f32x2 r;
    r.x = __builtin_amdgcn_rcpf(d.x);
    r.y = __builtin_amdgcn_rcpf(d.y);
    return x * r;
}

// eight GELUs as FOUR interleaved packed chains, stage by stage (a scheduling barrier between the stages keeps the compiler
// from re-serialising them to save registers): in the GEMM epilogue every stage of gelu_erf2 depends on the one before, so
// one pair at a time issues at the dependent-op latency plus the packed-fp32 / transcendental hazard wait states (388 s_nop
// in the GELU epilogue before); four independent pairs per stage fill those slots.
__device__ __forceinline__ void gelu_erf8(float (&v)[8]) {
    f32x2 x[4], xc[4], t[4], p[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) x[j] = f32x2{v[2 * j], v[2 * j + 1]};
#pragma unroll
    for (int j = 0; j < 4; ++j) { xc[j].x = __builtin_amdgcn_fmed3f(x[j].x, -5.5f, 5.5f); xc[j].y = __builtin_amdgcn_fmed3f(x[j].y, -5.5f, 5.5f); }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int j = 0; j < 4; ++j) t[j] = xc[j] * xc[j];
    __builtin_amdgcn_sched_barrier(0);
    const f32x2 c3 = {-2.48362952e-05f, -2.48362952e-05f}, c2 = {-7.36062896e-04f, -7.36062896e-04f};
    const f32x2 c1 = {1.05982735e-01f, 1.05982735e-01f}, c0 = {2.30164715e+00f, 2.30164715e+00f};
#pragma unroll
    for (int j = 0; j < 4; ++j) p[j] = __builtin_elementwise_fma(t[j], c3, c2);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int j = 0; j < 4; ++j) p[j] = __builtin_elementwise_fma(t[j], p[j], c1);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int j = 0; j < 4; ++j) p[j] = __builtin_elementwise_fma(t[j], p[j], c0);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int j = 0; j < 4; ++j) t[j] = xc[j] * p[j];
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int j = 0; j < 4; ++j) { t[j].x = __builtin_amdgcn_exp2f(-t[j].x); t[j].y = __builtin_amdgcn_exp2f(-t[j].y); }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int j = 0; j < 4; ++j) t[j] = t[j] + 1.0f;
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int j = 0; j < 4; ++j) { t[j].x = __builtin_amdgcn_rcpf(t[j].x); t[j].y = __builtin_amdgcn_rcpf(t[j].y); }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int j = 0; j < 4; ++j) { const f32x2 y = x[j] * t[j]; v[2 * j] = y.x; v[2 * j + 1] = y.y; }
}

// eight erf-GELUs to ~fp32 accuracy (Abramowitz & Stegun 7.1.26: |erf error| <= 1.5e-7) as four interleaved packed chains - the form the
// split8 path of the text32 mode uses everywhere (GEMM epilogues of both tile sizes and the stand-alone pass: one function, the same bits):
//   t = 1 / (1 + p |x| / sqrt 2),  e = 1 - t (a1 + t (a2 + t (a3 + t (a4 + t a5)))) exp(-x^2 / 2),  gelu = (x + |x| e) / 2
__device__ __forceinline__ void gelu_erf_as8(float (&v)[8]) {
    f32x2 x[4], t[4], p[4], q[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) x[j] = f32x2{v[2 * j], v[2 * j + 1]};
#pragma unroll
    for (int j = 0; j < 4; ++j) { t[j].x = fmaf(fabsf(x[j].x), 0.3275911f * 0.70710678118654752f, 1.0f); t[j].y = fmaf(fabsf(x[j].y), 0.3275911f * 0.70710678118654752f, 1.0f); }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int j = 0; j < 4; ++j) { t[j].x = __builtin_amdgcn_rcpf(t[j].x); t[j].y = __builtin_amdgcn_rcpf(t[j].y); }
#pragma unroll
    for (int j = 0; j < 4; ++j) q[j] = x[j] * x[j];
    __builtin_amdgcn_sched_barrier(0);
    const f32x2 a5 = {1.061405429f, 1.061405429f}, a4 = {-1.453152027f, -1.453152027f}, a3 = {1.421413741f, 1.421413741f};
    const f32x2 a2 = {-0.284496736f, -0.284496736f}, a1 = {0.254829592f, 0.254829592f};
#pragma unroll
    for (int j = 0; j < 4; ++j) p[j] = __builtin_elementwise_fma(a5, t[j], a4);
#pragma unroll
    for (int j = 0; j < 4; ++j) q[j] = q[j] * f32x2{-0.72134752044448170f, -0.72134752044448170f};     // -x^2 / 2 in the log2 domain
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int j = 0; j < 4; ++j) p[j] = __builtin_elementwise_fma(p[j], t[j], a3);
#pragma unroll
    for (int j = 0; j < 4; ++j) { q[j].x = __builtin_amdgcn_exp2f(q[j].x); q[j].y = __builtin_amdgcn_exp2f(q[j].y); }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int j = 0; j < 4; ++j) p[j] = __builtin_elementwise_fma(p[j], t[j], a2);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int j = 0; j < 4; ++j) p[j] = __builtin_elementwise_fma(p[j], t[j], a1);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int j = 0; j < 4; ++j) t[j] = t[j] * q[j];
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int j = 0; j < 4; ++j) p[j] = __builtin_elementwise_fma(-p[j], t[j], f32x2{1.0f, 1.0f});          // e = 1 - poly t exp
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int j = 0; j < 4; ++j) { p[j].x = fmaf(fabsf(x[j].x), p[j].x, x[j].x); p[j].y = fmaf(fabsf(x[j].y), p[j].y, x[j].y); }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int j = 0; j < 4; ++j) { const f32x2 y = p[j] * f32x2{0.5f, 0.5f}; v[2 * j] = y.x; v[2 * j + 1] = y.y; }
}

// ---- "split8" operand rows (round 6) ------------------------------------------------------------------------------------------------
// A fp32 value y travels to a GEMM as three terms: hi = fp16(y), lo8 = e4m3((y - hi) * 2^12), hi8 = e4m3(hi); a row of K values is stored
// as [K x hi (2 K bytes) | K x lo8 | K x hi8] = 4 K bytes.  Against weight rows [W_hi | e4m3(W_hi 2^e1) | e4m3(W_lo 2^e2)] one GEMM forms
//     A_hi W_hi^T   on v_mfma_f32_16x16x32_f16            (K / 64 K-tiles of 128 bytes)
//   + A_lo W_hi^T   on v_mfma_scale_f32_16x16x128_f8f6f4  (K / 128 K-tiles of 128 bytes; E8M0 scales 2^-12, 2^-e1)
//   + A_hi W_lo^T   on the same instruction               (K / 128 K-tiles; scales 1, 2^-e2)
// in ONE fp32 accumulator: the two correction products are 2^-11 of the result, so e4m3's 4 significant bits per factor leave an error
// of ~2^-16 of the result - at twice the fp16 MFMA rate (the three-product fp16 form of round 5 costs 3 K / 64 tiles, this one 2 K / 64).
// e4m3 has no infinity and v_cvt_pk_fp8_f32 does not saturate (480 -> NaN, measured: tools/f8_probe.hip): both fp8 terms are clamped to
// +-448 first (a clamped element loses correction accuracy, never correctness of the leading product).
constexpr float kSplitLoScale = 4096.0f;    // 2^12
constexpr int kSplitLoExp = 12;
__device__ __forceinline__ float clamp_e4m3(float x) { return __builtin_amdgcn_fmed3f(x, -448.0f, 448.0f); }
// four values -> 4 halves (two dwords), 4 lo8 bytes, 4 hi8 bytes (element j in byte j)
struct Split4 { unsigned h01, h23, lo8, hi8; };
__device__ __forceinline__ Split4 split8_x4(const float (&y)[4]) {
    float hf[4], lf[4];
    _Float16 hh[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        float v = y[j];
        asm volatile("" : "+v"(v));          // ONE value of y feeds both terms (a re-evaluation under another contraction can cross a rounding tie)
        hh[j] = (_Float16)v;
        float f = (float)hh[j];
        asm volatile("" : "+v"(f));
        hf[j] = clamp_e4m3(f);
        lf[j] = clamp_e4m3((v - f) * kSplitLoScale);
    }
    typedef __attribute__((ext_vector_type(2))) _Float16 h2;
    Split4 r;
    r.h01 = __builtin_bit_cast(unsigned, h2{hh[0], hh[1]});
    r.h23 = __builtin_bit_cast(unsigned, h2{hh[2], hh[3]});
    unsigned l = 0, h = 0;
    l = __builtin_amdgcn_cvt_pk_fp8_f32(lf[0], lf[1], l, false);
    l = __builtin_amdgcn_cvt_pk_fp8_f32(lf[2], lf[3], l, true);
    h = __builtin_amdgcn_cvt_pk_fp8_f32(hf[0], hf[1], h, false);
    h = __builtin_amdgcn_cvt_pk_fp8_f32(hf[2], hf[3], h, true);
    r.lo8 = l;
    r.hi8 = h;
    return r;
}
// c += A B on the block-scaled fp8 MFMA, IN PLACE (32 e4m3 values of k per lane and operand; sa / sb: E8M0 scale bytes, replicated)
typedef __attribute__((ext_vector_type(8))) int i32x8;
__device__ __forceinline__ void mfma_f8_acc(f32x4& c, i32x8 a, i32x8 b, int sa, int sb) {
    asm("v_mfma_scale_f32_16x16x128_f8f6f4 %0, %1, %2, %0, %3, %4 op_sel_hi:[0,0,0]" : "+v"(c) : "v"(a), "v"(b), "v"(sa), "v"(sb));
}

// ---- counter-based dropout of the fused training kernels (train_attn.hip, train_fused.hip) ---------------------------------------
// Element (row, col) of a launch is DROPPED iff its 16 random bits are below round(p * 65536).  One 32-bit hash gives the bits of the
// two adjacent columns (2j, 2j + 1) of a row: bits = hash32(row_key ^ j), row_key = a Weyl sequence over the rows offset by the seed -
// so a lane that owns 4 consecutive columns pays two hashes (4 integer multiplies), where the splitmix64 of (seed, flat index) of the
// stand-alone kernels (train.hip) costs three 64-bit multiplies per element: at 16 score elements per lane and key tile that hash alone
// was ~10x the tile's MFMA time in the attention kernels.  The forward and backward kernels regenerate the same mask from
// (seed, row, col); tests/test_train_ops_gpu.py holds the host replica.
__device__ __forceinline__ uint32_t hash32(uint32_t x) {      // "lowbias32": full avalanche, two multiplies
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}
constexpr uint32_t kDropWeyl = 0x9E3779B9u;
__device__ __forceinline__ uint32_t drop_threshold(float p) { return (uint32_t)(p * 65536.0f + 0.5f); }
__device__ __forceinline__ uint32_t drop_row_key(uint64_t seed, uint64_t row) {
    return (uint32_t)seed + (uint32_t)(seed >> 32) * 0x85EBCA6Bu + (uint32_t)row * kDropWeyl + (uint32_t)(row >> 32) * 0xC2B2AE35u;
}
__device__ __forceinline__ uint32_t drop_bits(uint32_t row_key, uint32_t col) { return hash32(row_key ^ (col >> 1)); }
__device__ __forceinline__ bool drop_kept(uint32_t bits, uint32_t col, uint32_t thr) { return ((col & 1u) ? (bits >> 16) : (bits & 0xffffu)) >= thr; }

// bijective XCD remap: consecutive "logical" ids land on one XCD (blocks b and b+8 share an XCD)
__device__ __forceinline__ int xcd_remap(int bid, int nblk) {
    const int q = nblk >> 3, r = nblk & 7, x = bid & 7;
    const int base = (x < r) ? x * (q + 1) : r * (q + 1) + (x - r) * q;
    return base + (bid >> 3);
}

}  // namespace cir

// ---- host side ----------------------------------------------------------------------------------
namespace cir {
// kernel-selection overrides (cir_set_tuning; misc.hip owns the storage): 0 = automatic
extern int g_tune[3];
}  // namespace cir
#define CIR_CHECK_PTR(p) do { if ((p) == nullptr) return CIR_EINVAL; } while (0)
#define CIR_LAUNCH_RESULT() do { hipError_t e_ = hipGetLastError(); return e_ == hipSuccess ? CIR_OK : (int)e_; } while (0)
static inline bool cir_aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }
