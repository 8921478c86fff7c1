// Probe (not part of libcirrank): how fast can one workgroup per CU stream GEMM operand tiles into LDS with LDS-DMA,
// with NO MFMA and NO ds_read beside it?  Same walk as the GEMM: persistent workgroups, tile t -> (tm, tn), K swept in
// 64-deep K-tiles of A 256 x 64 + W 256 x 64 (64 KiB), a ring of two 64-KiB buffers, one barrier per K-tile.
// Knobs: waves per workgroup (4 / 8), pieces allowed in flight when the wave waits, raster (n-fastest or m x n blocks).
//
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/dma_probe.hip -o tools/dma_probe && tools/dma_probe
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

typedef __attribute__((address_space(1))) const void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

struct Args {
    const char* A; const char* W; unsigned* sink;
    int M, N, K, tiles_m, tiles_n, gw;
};

template <int WAVES, int KEEP, int AUX_A = 0>   // KEEP = pieces of the newest K-tile a wave leaves in flight when it waits (0 = drain)
__global__ __launch_bounds__(WAVES * 64) void stream_kernel(const Args a) {
    __shared__ __attribute__((aligned(16))) char smem[2 * 65536];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    constexpr int PIECES = 64 / WAVES;            // 1-KiB pieces per wave and K-tile (half activation, half weight rows)
    constexpr int ROWS = 256 / WAVES;             // rows of each operand this wave fetches
    const int srow = lane >> 3, schunk = (lane & 7) ^ srow;
    const int nk = a.K >> 6;
    const int ntiles = a.tiles_m * a.tiles_n;
    const unsigned off = (unsigned)((srow * a.K + schunk * 8) * 2);
    unsigned acc = 0;
    for (int t = blockIdx.x; t < ntiles; t += gridDim.x) {
        // column groups of gw n-tiles (the GEMM's raster), XCD = blockIdx % 8 walks neighbouring ids
        const int q = ntiles >> 3, r = ntiles & 7, x = t & 7;
        int id = ((x < r) ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (t >> 3);
        const int gsz = a.tiles_m * a.gw;
        const int grp = id / gsz, rem = id - grp * gsz;
        const int tm = rem / a.gw, tn = grp * a.gw + rem % a.gw;
        const char* A_t = a.A + ((size_t)tm * 256 + wave * ROWS) * a.K * 2;
        const char* W_t = a.W + ((size_t)tn * 256 + wave * ROWS) * a.K * 2;
        for (int kt = 0; kt < nk; ++kt) {
            char* buf = smem + (kt & 1) * 65536;
#pragma unroll
            for (int j = 0; j < PIECES / 2; ++j) {
                __builtin_amdgcn_global_load_lds((gptr_t)(A_t + (size_t)(j * 8) * a.K * 2 + kt * 128 + off),
                                                 (lptr_t)(buf + (wave * ROWS + j * 8) * 128), 16, 0, AUX_A);
                __builtin_amdgcn_global_load_lds((gptr_t)(W_t + (size_t)(j * 8) * a.K * 2 + kt * 128 + off),
                                                 (lptr_t)(buf + 32768 + (wave * ROWS + j * 8) * 128), 16, 0, 0);
            }
            // the K-tile issued one trip earlier has landed (the newest may stay in flight)
            if (KEEP == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(%0)" :: "n"(KEEP) : "memory");
            __builtin_amdgcn_s_barrier();
            if (lane == 0 && wave == 0) acc += *reinterpret_cast<volatile unsigned*>(smem + ((kt & 1) ^ 1) * 65536);   // keep it honest
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }
    if (acc == 0x12345678u) a.sink[0] = acc;
}

template <int WAVES, int KEEP, int AUX_A = 0>
static void run(const char* name, Args a, int cus) {
    const int ntiles = a.tiles_m * a.tiles_n;
    dim3 grid(ntiles < cus ? ntiles : cus), block(WAVES * 64);
    for (int i = 0; i < 2; ++i) hipLaunchKernelGGL((stream_kernel<WAVES, KEEP, AUX_A>), grid, block, 0, 0, a);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    const int reps = 5;
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((stream_kernel<WAVES, KEEP, AUX_A>), grid, block, 0, 0, a);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double us = ms * 1e3 / reps;
    const double bytes = (double)ntiles * (a.K / 64) * 65536.0;
    printf("  %-28s gw=%d: %9.1f us  %6.1f GB/s per CU  %6.2f TB/s chip  (as a GEMM: %7.1f TF/s)  %s\n", name, a.gw, us,
           bytes / us / 1e3 / grid.x, bytes / us / 1e6, 2.0 * a.M * a.N * a.K / us / 1e6, hipGetErrorString(hipGetLastError()));
    fflush(stdout);
}

int main() {
    int cus = 256;
    hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
    const int shapes[][3] = {{318464, 3072, 768}, {318464, 768, 3072}, {318464, 2304, 768}};
    for (auto& sh : shapes) {
        Args a;
        a.M = sh[0]; a.N = sh[1]; a.K = sh[2]; a.tiles_m = a.M / 256; a.tiles_n = a.N / 256;
        char *dA, *dW; unsigned* sink;
        hipMalloc(&dA, (size_t)a.M * a.K * 2); hipMalloc(&dW, (size_t)a.N * a.K * 2); hipMalloc(&sink, 64);
        hipMemset(dA, 1, (size_t)a.M * a.K * 2); hipMemset(dW, 1, (size_t)a.N * a.K * 2);
        a.A = dA; a.W = dW; a.sink = sink;
        printf("M=%d N=%d K=%d\n", a.M, a.N, a.K);
        for (int gw : {1, 2, 3, 4, 6, 12}) {
            if (a.tiles_n % gw) continue;
            a.gw = gw;
            run<8, 0>("8 waves, drain each K-tile", a, cus);
            run<8, 8>("8 waves, 1 K-tile in flight", a, cus);
            run<8, 8, 2>("8 waves, 1 in flight, nt on A", a, cus);
        }
        hipFree(dA); hipFree(dW); hipFree(sink);
    }
    return 0;
}
