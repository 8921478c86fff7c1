// Probe (not part of libcirrank): what does the part SUSTAIN on a register-only MFMA stream under its power cap, per instruction shape
// and operand type?  One workgroup of 8 waves per CU (2 per SIMD, as the GEMM runs), every wave issues independent accumulator chains
// of one MFMA shape on random operands held in registers - no LDS, no memory.  The GEMMs of the step sit at 1.6-1.8 GHz with the matrix
// pipe 60-70 % busy: if a shape sustains more TFLOP/s here, it delivers more flops per joule.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/mfma_power_probe.hip -o tools/mfma_power_probe && tools/mfma_power_probe
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

template <int SHAPE, bool BF>   // SHAPE 16: v_mfma_f32_16x16x32, 32: v_mfma_f32_32x32x16
__global__ __launch_bounds__(512) void probe(const uint32_t* seed, float* sink, int iters, unsigned long long* clk) {
    using X8 = typename std::conditional<BF, bf16x8, f16x8>::type;
    const int tid = blockIdx.x * blockDim.x + threadIdx.x;
    X8 a[4], b[4];
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 8; ++j) {
            uint32_t h = seed[(tid * 64 + i * 16 + j) & 0xffff];
            const float va = ((int)(h & 0xffff) - 32768) / 32768.0f, vb = ((int)(h >> 16) - 32768) / 32768.0f;
            a[i][j] = (decltype(a[i][j] + a[i][j]))va;
            b[i][j] = (decltype(b[i][j] + b[i][j]))vb;
        }
    unsigned long long t0 = 0, r0 = 0, t1 = 0, r1 = 0;
    if (threadIdx.x == 0) asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0), "=s"(r0) :: "memory");
    float out = 0.f;
    if constexpr (SHAPE == 16) {
        f32x4 c[32];
        for (int i = 0; i < 32; ++i) c[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 32; ++i) {
                if constexpr (BF) c[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i & 3], b[(i >> 2) & 3], c[i], 0, 0, 0);
                else c[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[i & 3], b[(i >> 2) & 3], c[i], 0, 0, 0);
            }
        }
        for (int i = 0; i < 32; ++i) out += c[i][0] + c[i][3];
    } else {
        f32x16 c[8];
        for (int i = 0; i < 8; ++i)
            for (int j = 0; j < 16; ++j) c[i][j] = 0.f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int r = 0; r < 2; ++r)
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    if constexpr (BF) c[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[(i + r) & 3], b[(i >> 1) & 3], c[i], 0, 0, 0);
                    else c[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[(i + r) & 3], b[(i >> 1) & 3], c[i], 0, 0, 0);
                }
        }
        for (int i = 0; i < 8; ++i) out += c[i][0] + c[i][15];
    }
    if (threadIdx.x == 0) {
        asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1), "=s"(r1) :: "memory");
        clk[blockIdx.x * 2] = t1 - t0;
        clk[blockIdx.x * 2 + 1] = r1 - r0;
    }
    sink[tid] = out;
}

template <int SHAPE, bool BF>
static void run(const char* name, const uint32_t* seed, float* sink, unsigned long long* clk, int cus) {
    // flops per wave and iteration: 16 shape: 32 MFMAs x 16*16*32*2; 32 shape: 16 MFMAs x 32*32*16*2 - the same 524 288
    const int iters = 40000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    double best = 0, last_clk = 0;
    for (int rep = 0; rep < 6; ++rep) {      // ~0.35 s each: the later repetitions run at the clock the part holds under this load
        hipEventRecord(e0);
        hipLaunchKernelGGL((probe<SHAPE, BF>), dim3(cus), dim3(512), 0, 0, seed, sink, iters, clk);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        const double tf = (double)cus * 8 * iters * 524288.0 / (ms * 1e-3) / 1e12;
        unsigned long long h[2];
        hipMemcpy(h, clk, sizeof(h), hipMemcpyDeviceToHost);
        last_clk = (double)h[0] / ((double)h[1] / 100e6) / 1e9;     // shader cycles over 100-MHz real-time ticks
        printf("%-28s rep %d: %8.1f ms  %7.1f TFLOP/s  in-kernel clock %.3f GHz  (matrix pipe issuing %.1f %% of the cycles: 2 waves per SIMD x 16 / 32 cycles per MFMA)\n", name, rep, ms, tf, last_clk,
               100.0 * ((double)iters * (SHAPE == 16 ? 32 * 16 : 16 * 32) * 2) / (double)h[0]);
        if (rep >= 3 && tf > best) best = tf;
    }
    printf("%-28s sustained (best of the last 3): %.1f TFLOP/s at %.3f GHz\n\n", name, best, last_clk);
}

int main() {
    int cus = 0;
    hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
    uint32_t* seed; float* sink; unsigned long long* clk;
    hipMalloc(&seed, 65536 * 4); hipMalloc(&sink, (size_t)cus * 512 * 4); hipMalloc(&clk, (size_t)cus * 16);
    uint32_t* h = (uint32_t*)malloc(65536 * 4);
    uint32_t x = 12345;
    for (int i = 0; i < 65536; ++i) { x = x * 1664525u + 1013904223u; h[i] = x; }
    hipMemcpy(seed, h, 65536 * 4, hipMemcpyHostToDevice);
    printf("%d CUs, 8 waves per CU (2 per SIMD), register-only MFMA streams on random operands\n", cus);
    run<16, false>("f16  16x16x32", seed, sink, clk, cus);
    run<32, false>("f16  32x32x16", seed, sink, clk, cus);
    run<16, true>("bf16 16x16x32", seed, sink, clk, cus);
    run<32, true>("bf16 32x32x16", seed, sink, clk, cus);
    run<16, false>("f16  16x16x32 (again)", seed, sink, clk, cus);
    return 0;
}
