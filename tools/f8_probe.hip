// Probe of the block-scaled fp8 MFMA on gfx950 (round 6; calibration tool, never part of libcirrank):
//   1. operand lane map of v_mfma_scale_f32_16x16x128_f8f6f4 with e4m3 operands (exact small integers, asymmetric B);
//   2. what the E8M0 scale operands do;
//   3. v_cvt_pk_fp8_f32 on values outside e4m3's range (saturate or NaN?);
//   4. cycles per MFMA (s_memtime, one wave per SIMD) and chip-wide sustained rate on random operands next to the fp16 16x16x32 form.
//   hipcc --offload-arch=gfx950 -O3 -o tools/f8_probe tools/f8_probe.hip && tools/f8_probe
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

typedef __attribute__((ext_vector_type(8))) int i32x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

__global__ void mm_one(const i32x8* a, const i32x8* b, f32x4* c, int sa, int sb) {
    const int l = threadIdx.x;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    acc = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a[l], b[l], acc, 0, 0, 0, sa, 0, sb);
    c[l] = acc;
}

__global__ void cvt(const float* x, unsigned* y, int n) {
    const int l = threadIdx.x;
    if (l < n) y[l] = __builtin_amdgcn_cvt_pk_fp8_f32(x[l], 0.f, 0u, false) & 0xff;
}

template <int KIND>   // 0: fp16 16x16x32, 1: scaled fp8 16x16x128
__global__ __launch_bounds__(256) void rate(const unsigned* seed, float* out, unsigned long long* cyc, int iters) {
    const int l = threadIdx.x & 63;
    unsigned s = seed[(blockIdx.x * 256 + threadIdx.x) & 4095];
    i32x8 a, b;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        s = s * 1664525u + 1013904223u; a[j] = (int)(s & (KIND ? 0x3f3f3f3fu : 0x3bff3bffu));
        s = s * 1664525u + 1013904223u; b[j] = (int)(s & (KIND ? 0x3f3f3f3fu : 0x3bff3bffu));
    }
    f32x4 acc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int one = 0x7f7f7f7f;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            if constexpr (KIND == 1) {
                acc[j] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, acc[j], 0, 0, 0, one, 0, one);
            } else {
                f16x8 ah = __builtin_bit_cast(f16x8, __builtin_shufflevector(a, a, 0, 1, 2, 3)), bh = __builtin_bit_cast(f16x8, __builtin_shufflevector(b, b, 0, 1, 2, 3));
                acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh, acc[j], 0, 0, 0);
            }
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float r = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) r += acc[j][0] + acc[j][1] + acc[j][2] + acc[j][3];
    out[blockIdx.x * 256 + threadIdx.x] = r;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[KIND] = t1 - t0;
}

static unsigned char e4m3(float v) {   // exact encoder for the small integers used below
    if (v == 0) return 0;
    unsigned char sgn = v < 0 ? 0x80 : 0;
    float a = fabsf(v);
    int e = (int)floorf(log2f(a));
    float m = a / ldexpf(1.f, e) - 1.f;            // [0, 1)
    return sgn | (unsigned char)(((e + 7) << 3) | (int)(m * 8));
}

int main() {
    // ---- 1 + 2: lane map and scales -----------------------------------------------------------------------------------
    float A[16][128], B[128][16];
    srand(5);
    for (int i = 0; i < 16; ++i) for (int k = 0; k < 128; ++k) A[i][k] = (float)((rand() % 7) - 3);
    for (int k = 0; k < 128; ++k) for (int j = 0; j < 16; ++j) B[k][j] = (float)((rand() % 5) - 2) * (j % 3 == 0 ? 2.f : 1.f);
    std::vector<unsigned char> ha(64 * 32), hb(64 * 32);
    for (int l = 0; l < 64; ++l) for (int j = 0; j < 32; ++j) {
        ha[l * 32 + j] = e4m3(A[l & 15][32 * (l >> 4) + j]);
        hb[l * 32 + j] = e4m3(B[32 * (l >> 4) + j][l & 15]);
    }
    i32x8 *da, *db; f32x4* dc;
    CK(hipMalloc(&da, 64 * 32)); CK(hipMalloc(&db, 64 * 32)); CK(hipMalloc(&dc, 64 * 16));
    CK(hipMemcpy(da, ha.data(), 64 * 32, hipMemcpyHostToDevice)); CK(hipMemcpy(db, hb.data(), 64 * 32, hipMemcpyHostToDevice));
    for (int pass = 0; pass < 3; ++pass) {
        const int sa = pass == 0 ? 0x7f7f7f7f : (pass == 1 ? 0x82828282 : 0x7f7f7f7f), sb = pass == 2 ? 0x7c7c7c7c : 0x7f7f7f7f;
        hipLaunchKernelGGL(mm_one, dim3(1), dim3(64), 0, 0, da, db, dc, sa, sb);
        float hc[64][4];
        CK(hipMemcpy(hc, dc, sizeof(hc), hipMemcpyDeviceToHost));
        const float mul = pass == 0 ? 1.f : (pass == 1 ? 8.f : 0.125f);
        int bad = 0;
        for (int l = 0; l < 64; ++l) for (int r = 0; r < 4; ++r) {
            const int row = 4 * (l >> 4) + r, col = l & 15;
            float ref = 0; for (int k = 0; k < 128; ++k) ref += A[row][k] * B[k][col];
            if (hc[l][r] != ref * mul) ++bad;
        }
        printf("lane map k = 32 (lane >> 4) + byte, D[4 (lane >> 4) + r][lane & 15], scale_a %#x scale_b %#x (x %g): %s (%d of 256 wrong)\n", sa & 0xff, sb & 0xff, mul,
               bad ? "MISMATCH" : "ok", bad);
    }
    // ---- 3: conversion at the edge of the range ------------------------------------------------------------------------
    float xs[] = {448.f, 449.f, 463.9f, 464.f, 480.f, 500.f, 1e6f, -1e6f, INFINITY, NAN, 0.001953125f, 0.0009765625f, 0.00097f, 1.0625f, 1.1875f, -0.3f};
    const int nx = sizeof(xs) / 4;
    float* dx; unsigned* dy;
    CK(hipMalloc(&dx, sizeof(xs))); CK(hipMalloc(&dy, nx * 4));
    CK(hipMemcpy(dx, xs, sizeof(xs), hipMemcpyHostToDevice));
    hipLaunchKernelGGL(cvt, dim3(1), dim3(64), 0, 0, dx, dy, nx);
    unsigned ys[32];
    CK(hipMemcpy(ys, dy, nx * 4, hipMemcpyDeviceToHost));
    for (int i = 0; i < nx; ++i) printf("cvt_pk_fp8_f32(%g) = 0x%02x\n", xs[i], ys[i]);
    // ---- 4: rate ------------------------------------------------------------------------------------------------------------
    unsigned* dseed; float* dout; unsigned long long* dcyc;
    std::vector<unsigned> hs(4096);
    for (auto& v : hs) v = (unsigned)rand() * 2654435761u;
    CK(hipMalloc(&dseed, 4096 * 4)); CK(hipMalloc(&dout, 1024 * 256 * 4)); CK(hipMalloc(&dcyc, 16));
    CK(hipMemcpy(dseed, hs.data(), 4096 * 4, hipMemcpyHostToDevice));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int iters = 20000;
    for (int kind = 0; kind < 2; ++kind) {
        for (int blocks : {1, 512}) {
            for (int rep = 0; rep < 3; ++rep) {
                CK(hipEventRecord(e0));
                if (kind == 0) hipLaunchKernelGGL((rate<0>), dim3(blocks), dim3(256), 0, 0, dseed, dout, dcyc, iters);
                else hipLaunchKernelGGL((rate<1>), dim3(blocks), dim3(256), 0, 0, dseed, dout, dcyc, iters);
                CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                unsigned long long cy[2]; CK(hipMemcpy(cy, dcyc, 16, hipMemcpyDeviceToHost));
                const double flop = 2.0 * 16 * 16 * (kind ? 128 : 32) * 8.0 * iters * 4 * blocks;
                if (rep == 2) printf("%s, %d workgroup(s) x 4 waves: %.1f s_memtime ticks per MFMA (100 MHz ticks x clock ratio), %.1f TFLOP/s, %.2f ms\n", kind ? "fp8 scaled 16x16x128" : "fp16 16x16x32", blocks,
                                     (double)cy[kind] / (8.0 * iters), flop / (ms * 1e-3) / 1e12, ms);
            }
        }
    }
    return 0;
}
