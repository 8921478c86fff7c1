// Probe (not part of libcirrank): does a 4-wave GEMM - ONE wave per SIMD, each owning 128 x 128 outputs of a 256 x 256
// tile in 256 accumulator registers (AGPRs) - keep the matrix pipe busier than the shipped 8-wave kernel, whose K loop
// is bound by LDS-DMA issue?  Same tile, same LDS layout/swizzle, same 16x16x32 MFMA; per K-tile each wave issues
// 16 LDS-DMA pieces, 32 ds_read_b128 and 128 MFMAs, two workgroup barriers.  LDS is a ring of four 32-deep
// k-steps (32 KiB each), so up to 96 KiB of operands are in flight per CU (the delivery rate is bytes in flight / ~1 us).
//
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/gemm4w_probe.hip -o tools/gemm4w_probe && tools/gemm4w_probe
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <math.h>
#include <stdio.h>
#include <string.h>
#include <stdlib.h>
#include <vector>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((address_space(1))) const void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

constexpr int T = 256;
constexpr int kBuf = 32768;   // one stage = one 32-deep k-step: A 256 x 32 + B 256 x 32, 16-bit (64-byte rows)
constexpr int kB = 16384;     // offset of the weight half inside a stage
constexpr int NS = 4;         // ring of stages: one being read, up to three in flight (96 KiB)

struct Args {
    const __bf16* A; const __bf16* W; __bf16* C;
    int M, N, K, tiles_m, tiles_n;
};

__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void gemm4w(const Args a) {
    __shared__ __attribute__((aligned(16))) char smem[NS * kBuf];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 1, wc = wave & 1;
    const int r15 = lane & 15, g = lane >> 4;
    const int srow = lane >> 2, schunk = (lane & 3) ^ ((lane >> 4) & 3);   // DMA piece = 16 rows x 64 B
    const int nk = a.K >> 5;                                               // k-steps
    const int ntiles = a.tiles_m * a.tiles_n;

    // fragment read addresses (lane part): row r15 of a 16-row block, k-step chunk swizzled with the row
    const unsigned rd0 = r15 * 64 + ((g ^ ((r15 >> 2) & 3)) << 4);
    const char* const a_rd = smem + wr * 128 * 64;
    const char* const b_rd = smem + kB + wc * 128 * 64;

    f32x4 acc[8][8];
    bf16x8 af[2][8], wf[2][8];

#define READ_SET(S, BUF)                                                                                          \
    _Pragma("unroll") for (int i = 0; i < 8; ++i) {                                                               \
        af[S][i] = *reinterpret_cast<const bf16x8*>(a_rd + (BUF) * kBuf + i * 1024 + rd0);                        \
        wf[S][i] = *reinterpret_cast<const bf16x8*>(b_rd + (BUF) * kBuf + i * 1024 + rd0);                        \
    }
#define MMA_ROWS(S, M0, M1)                                                                                       \
    _Pragma("unroll") for (int mi = (M0); mi < (M1); ++mi)                                                        \
        _Pragma("unroll") for (int ni = 0; ni < 8; ++ni)                                                          \
            asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc[mi][ni]) : "v"(wf[S][ni]), "v"(af[S][mi]));
#define SYNC()                                   \
    __builtin_amdgcn_sched_barrier(0);           \
    __builtin_amdgcn_s_barrier();                \
    __builtin_amdgcn_sched_barrier(0);
// piece J (0..7) of k-step KT into slot BUF: pieces 0-3 = this wave's 64 activation rows, 4-7 = its 64 weight rows
#define DMA(J, BUF, KT)                                                                                           \
    {                                                                                                             \
        const char* src_ = (J) < 4 ? A_t + (size_t)((J) * 16) * a.K * 2 + (KT) * 64 + a_off                       \
                                   : W_t + (size_t)(((J) - 4) * 16) * a.K * 2 + (KT) * 64 + a_off;                \
        char* dst_ = smem + (BUF) * kBuf + ((J) < 4 ? 0 : kB) + (wave * 64 + ((J) & 3) * 16) * 64;                \
        __builtin_amdgcn_global_load_lds((gptr_t)src_, (lptr_t)dst_, 16, 0, 0);                                   \
    }

    for (int t = blockIdx.x; t < ntiles; t += gridDim.x) {
        const int tm = t / a.tiles_n, tn = t % a.tiles_n;
        const int m0 = tm * T, n0 = tn * T;
        const char* A_t = reinterpret_cast<const char*>(a.A + (size_t)(m0 + wave * 64) * a.K);
        const char* W_t = reinterpret_cast<const char*>(a.W + (size_t)(n0 + wave * 64) * a.K);
        const unsigned a_off = (unsigned)((srow * a.K + schunk * 8) * 2);   // row srow of a 16-row piece, swizzled 16-byte chunk
#pragma unroll
        for (int mi = 0; mi < 8; ++mi)
#pragma unroll
            for (int ni = 0; ni < 8; ++ni) acc[mi][ni] = f32x4{0.f, 0.f, 0.f, 0.f};

        __syncthreads();   // previous tile's LDS reads are done
#pragma unroll
        for (int st = 0; st < NS; ++st) {
            const int ks = st < nk ? st : nk - 1;
#pragma unroll
            for (int j = 0; j < 8; ++j) DMA(j, st, ks)
        }
        asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
        SYNC();
        READ_SET(0, 0)

        // Branch-free K loop, NS k-steps per trip (static slot / register-set indices; the probe takes K % 128 == 0).
        // Step i: stage i+1 has landed and every wave is done reading stage i's slot (one barrier); fragments of stage
        // i+1 are read while the 64 MFMAs of stage i run; the freed slot is refilled with stage i+NS, one piece per
        // MFMA row.  Refills past the end re-fetch the last k-step (every step issues and waits for the same count).
#define STEP(I, KS_NEXT)                                                                                            \
        asm volatile("s_waitcnt vmcnt(16)\n\ts_waitcnt lgkmcnt(0)" ::: "memory");                                   \
        SYNC();                                                                                                     \
        READ_SET(((I) + 1) & 1, ((I) + 1) % NS)                                                                     \
        __builtin_amdgcn_sched_barrier(0);                                                                          \
        MMA_ROWS((I) & 1, 0, 1) __builtin_amdgcn_sched_barrier(0); DMA(0, (I) % NS, KS_NEXT) __builtin_amdgcn_sched_barrier(0); \
        MMA_ROWS((I) & 1, 1, 2) __builtin_amdgcn_sched_barrier(0); DMA(4, (I) % NS, KS_NEXT) __builtin_amdgcn_sched_barrier(0); \
        MMA_ROWS((I) & 1, 2, 3) __builtin_amdgcn_sched_barrier(0); DMA(1, (I) % NS, KS_NEXT) __builtin_amdgcn_sched_barrier(0); \
        MMA_ROWS((I) & 1, 3, 4) __builtin_amdgcn_sched_barrier(0); DMA(5, (I) % NS, KS_NEXT) __builtin_amdgcn_sched_barrier(0); \
        MMA_ROWS((I) & 1, 4, 5) __builtin_amdgcn_sched_barrier(0); DMA(2, (I) % NS, KS_NEXT) __builtin_amdgcn_sched_barrier(0); \
        MMA_ROWS((I) & 1, 5, 6) __builtin_amdgcn_sched_barrier(0); DMA(6, (I) % NS, KS_NEXT) __builtin_amdgcn_sched_barrier(0); \
        MMA_ROWS((I) & 1, 6, 7) __builtin_amdgcn_sched_barrier(0); DMA(3, (I) % NS, KS_NEXT) __builtin_amdgcn_sched_barrier(0); \
        MMA_ROWS((I) & 1, 7, 8) __builtin_amdgcn_sched_barrier(0); DMA(7, (I) % NS, KS_NEXT) __builtin_amdgcn_sched_barrier(0);

        for (int kt = 0; kt < nk; kt += NS) {
            const int k4 = kt + 4 < nk ? kt + 4 : nk - 1, k5 = kt + 5 < nk ? kt + 5 : nk - 1;
            const int k6 = kt + 6 < nk ? kt + 6 : nk - 1, k7 = kt + 7 < nk ? kt + 7 : nk - 1;
            STEP(0, k4)
            STEP(1, k5)
            STEP(2, k6)
            STEP(3, k7)
        }
        asm volatile("s_waitcnt vmcnt(0)\n\ts_waitcnt lgkmcnt(0)" ::: "memory");   // junk refills / reads of the last trip

        // ---- probe epilogue: direct 8-byte stores (lane = activation row r15, features 4g..4g+3 of each 16-block) ---
#pragma unroll
        for (int mi = 0; mi < 8; ++mi) {
            const int m = m0 + wr * 128 + mi * 16 + r15;
#pragma unroll
            for (int ni = 0; ni < 8; ++ni) {
                const int n = n0 + wc * 128 + ni * 16 + g * 4;
                typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
                bf16x4 o = {(__bf16)acc[mi][ni][0], (__bf16)acc[mi][ni][1], (__bf16)acc[mi][ni][2], (__bf16)acc[mi][ni][3]};
                if (m < a.M && n + 4 <= a.N) *reinterpret_cast<bf16x4*>(a.C + (size_t)m * a.N + n) = o;
            }
        }
    }
}

static float bf16_to_f(uint16_t v) { uint32_t u = (uint32_t)v << 16; float f; memcpy(&f, &u, 4); return f; }
static uint16_t f_to_bf16(float f) { uint32_t u; memcpy(&u, &f, 4); u += 0x7fff + ((u >> 16) & 1); return (uint16_t)(u >> 16); }

int main(int argc, char** argv) {
    const int shapes[][3] = {{318464, 768, 768}, {318464, 3072, 768}, {318464, 768, 3072}, {8192, 8192, 8192}, {4096, 4096, 4096}};
    int cus = 256;
    hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
    for (auto& sh : shapes) {
        const int M = sh[0], N = sh[1], K = sh[2];
        std::vector<uint16_t> hA((size_t)M * K), hW((size_t)N * K);
        uint32_t s = 12345;
        auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((int)((s >> 16) & 0xff) - 128) / 128.0f; };
        for (auto& v : hA) v = f_to_bf16(rnd());
        for (auto& v : hW) v = f_to_bf16(rnd() * 0.1f);
        __bf16 *dA, *dW, *dC;
        hipMalloc(&dA, hA.size() * 2); hipMalloc(&dW, hW.size() * 2); hipMalloc(&dC, (size_t)M * N * 2);
        hipMemcpy(dA, hA.data(), hA.size() * 2, hipMemcpyHostToDevice);
        hipMemcpy(dW, hW.data(), hW.size() * 2, hipMemcpyHostToDevice);
        Args a{dA, dW, dC, M, N, K, M / T, N / T};
        const int ntiles = a.tiles_m * a.tiles_n;
        dim3 grid(ntiles < cus ? ntiles : cus), block(256);
        for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(gemm4w, grid, block, 0, 0, a);
        hipEvent_t e0, e1;
        hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0);
        const int reps = 10;
        for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(gemm4w, grid, block, 0, 0, a);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        const hipError_t err = hipGetLastError();
        std::vector<uint16_t> hC(1024);
        // spot check: 64 outputs against a CPU dot product
        double max_err = 0;
        for (int i = 0; i < 64; ++i) {
            const int m = (int)(((uint64_t)i * 2654435761u) % M), n = (int)(((uint64_t)i * 40503u + 17) % N);
            uint16_t c;
            hipMemcpy(&c, dC + (size_t)m * N + n, 2, hipMemcpyDeviceToHost);
            double ref = 0;
            for (int k = 0; k < K; ++k) ref += (double)bf16_to_f(hA[(size_t)m * K + k]) * bf16_to_f(hW[(size_t)n * K + k]);
            const double e = fabs(ref - bf16_to_f(c)) / (fabs(ref) + 1.0);
            if (e > max_err) max_err = e;
        }
        const double us = ms * 1e3 / reps;
        printf("M=%d N=%d K=%d: %9.1f us %8.1f TF/s  spot-check max rel err %.4f  (%s)\n", M, N, K, us,
               2.0 * M * N * K / us / 1e6, max_err, hipGetErrorString(err));
        fflush(stdout);
        hipFree(dA); hipFree(dW); hipFree(dC);
    }
    return 0;
}
