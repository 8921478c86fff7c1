// Training-mode operators of the two-branch encoder (BLIP_NLVR.img_txt_fusion in train() mode + backward,
// blip_stage2.py:65-99 driven by stage2_train.py:202-216).  The dense Linear layers keep running on the MFMA GEMM
// (cir_gemm_bias_act: forward and dgrad, the latter through a transposed weight copy; wgrad is cir_bmm reading dy and x as stored); this file holds
// what the training pass needs besides: 16-bit transposes, a batched matmul on the matrix cores for products of any extents and
// storage orders (weight gradients dy^T x read as stored; the un-fused attention and its four adjoints per (candidate or
// triplet, head)), row softmax with additive mask and counter-based dropout (+ backward), LayerNorm backward, GELU / ReLU
// forward-backward, dropout, column sums, embedding scatter-add, AdamW.  None of it is on the inference path.

#include <algorithm>
#include "common.hpp"

namespace cir {

// counter-based uniform in [0, 1): one 64-bit mix (splitmix64) of (seed, element index) - the same mask is regenerated in
// the backward pass from the same (seed, index), no mask tensor is stored
__device__ __forceinline__ float uniform01(uint64_t seed, uint64_t idx) {
    uint64_t z = seed + 0x9E3779B97F4A7C15ull * (idx + 1);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z ^= z >> 31;
    return (float)(z >> 40) * (1.0f / 16777216.0f);
}

template <typename T> __device__ __forceinline__ float ldf(const T* p) { return static_cast<float>(*p); }

// ---- transpose: dst[b][c][r] = src[b][r][c] (16-bit elements; 32 x 32 tiles through LDS) --------------------------------
template <typename T>
__global__ __launch_bounds__(256) void transpose_kernel(const T* src, T* dst, int rows, int cols, int64_t ld_src, int64_t ld_dst,
                                                        int64_t s_src, int64_t s_dst) {
    __shared__ T tile[32][33];
    const int b = blockIdx.z;
    const int r0 = blockIdx.y * 32, c0 = blockIdx.x * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 8 rows per pass
    for (int i = ty; i < 32; i += 8) {
        const int r = r0 + i, c = c0 + tx;
        tile[i][tx] = (r < rows && c < cols) ? src[b * s_src + (int64_t)r * ld_src + c] : static_cast<T>(0.f);
    }
    __syncthreads();
    for (int i = ty; i < 32; i += 8) {
        const int c = c0 + i, r = r0 + tx;
        if (c < cols && r < rows) dst[b * s_dst + (int64_t)c * ld_dst + r] = tile[tx][i];
    }
}

// ---- many transposes in one launch: matrix i (rows_i x cols_i, contiguous) at element offset off_i of src -> its transpose at the SAME
// offset of dst.  table = 4 int64 per matrix: {offset, rows, cols, first 32 x 32 tile}; a block finds its matrix by binary search.
// (The trainer's dgrad operands: every trained weight of the flat 16-bit parameter buffer, once per step.)
template <typename T>
__global__ __launch_bounds__(256) void transpose_multi_kernel(const T* src, T* dst, const int64_t* table, int count) {
    __shared__ T tile[32][33];
    int lo = 0, hi = count - 1;
    const int64_t bid = blockIdx.x;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (table[4 * mid + 3] <= bid) lo = mid; else hi = mid - 1;
    }
    const int64_t off = table[4 * lo], t = bid - table[4 * lo + 3];
    const int rows = (int)table[4 * lo + 1], cols = (int)table[4 * lo + 2];
    const int tiles_c = (cols + 31) / 32;
    const int r0 = (int)(t / tiles_c) * 32, c0 = (int)(t % tiles_c) * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int i = ty; i < 32; i += 8) {
        const int r = r0 + i, c = c0 + tx;
        tile[i][tx] = (r < rows && c < cols) ? src[off + (int64_t)r * cols + c] : static_cast<T>(0.f);
    }
    __syncthreads();
    for (int i = ty; i < 32; i += 8) {
        const int c = c0 + i, r = r0 + tx;
        if (c < cols && r < rows) dst[off + (int64_t)c * rows + r] = tile[tx][i];
    }
}

// ---- small batched matmul: C[b] = alpha * op(A[b]) * op(B[b]) (+ C[b]) ----------------------------------------------------
// op(A) is (M, K): A stored (M, K) [ta = 0] or (K, M) [ta = 1]; op(B) is (K, N): B stored (K, N) [tb = 0] or (N, K) [tb = 1].
// 16 x 16 output tile per 256-thread block, fp32 accumulate; inputs TI (16-bit or float), output TO.
template <typename TI, typename TO>
__global__ __launch_bounds__(256) void bmm_kernel(const TI* A, const TI* B, TO* C, int M, int N, int K, int64_t lda, int64_t ldb, int64_t ldc,
                                                  int ta, int tb, int nb2, int64_t sA1, int64_t sA2, int64_t sB1, int64_t sB2, int64_t sC1, int64_t sC2,
                                                  float alpha, int accumulate) {
    __shared__ float As[16][17], Bs[16][17];
    const int z1 = blockIdx.z / nb2, z2 = blockIdx.z % nb2;
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    const int m = blockIdx.y * 16 + ty, n = blockIdx.x * 16 + tx;
    const TI* Ab = A + z1 * sA1 + z2 * sA2;
    const TI* Bb = B + z1 * sB1 + z2 * sB2;
    float acc = 0.f;
    for (int k0 = 0; k0 < K; k0 += 16) {
        {   // A tile: element (ty = m-local, tx = k-local)
            const int kk = k0 + tx;
            As[ty][tx] = (m < M && kk < K) ? ldf(ta ? Ab + (int64_t)kk * lda + m : Ab + (int64_t)m * lda + kk) : 0.f;
        }
        {   // B tile: element (ty = k-local, tx = n-local)
            const int kk = k0 + ty;
            Bs[ty][tx] = (kk < K && n < N) ? ldf(tb ? Bb + (int64_t)n * ldb + kk : Bb + (int64_t)kk * ldb + n) : 0.f;
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 16; ++k) acc = fmaf(As[ty][k], Bs[k][tx], acc);
        __syncthreads();
    }
    if (m < M && n < N) {
        TO* cp = C + z1 * sC1 + z2 * sC2 + (int64_t)m * ldc + n;
        const float v = alpha * acc + (accumulate ? static_cast<float>(*cp) : 0.f);
        *cp = static_cast<TO>(v);
    }
}

// ---- the same product on the matrix cores (16-bit operands) ------------------------------------------------------------------
// 64 x 64 output tile per 256-thread block (4 waves, 32 x 32 each = 2 x 2 MFMA 16x16x32 tiles), K in steps of 32 through LDS.
// Every extent is arbitrary (edges are zero-filled / predicated) and each operand may be stored either way, so the adjoints
// need no transposed copies: wgrad reads dy (rows, N) and x (rows, K) as they are (ta = 1), attention reads head slices of the
// (rows, D) projections in place.  Two batch levels (z = z1 * nb2 + z2, a stride per level and operand): (candidate, head) or
// (triplet, head) in one launch; split-K of a weight gradient is a batch over row chunks into partial sums.
// 16-byte global loads when the host says the operand allows them (leading dimension and strides multiples of 8 elements,
// base 16-byte aligned; a row then also has room for the 8-element read past a ragged edge), 2-byte loads otherwise.
struct BmmArgs {
    const void* A; const void* B; void* C;
    int M, N, K;
    int64_t lda, ldb, ldc;
    int ta, tb, nb2;
    int64_t sA1, sA2, sB1, sB2, sC1, sC2;
    float alpha;
    int accumulate, vecA, vecB;
};

constexpr int kBmmLd = 40;      // LDS row: 32 k-elements + 8 pad (80 bytes: 16-byte aligned fragments, rows spread over the banks)

// 64 rows of an operand tile (rows r0 .. r0+63 of the M or N extent x 32 of K) from global memory into 8 registers per thread
// kmajor = 0: stored (R, K), K contiguous: thread -> (row = tid / 4, 8 consecutive k)
// kmajor = 1: stored (K, R), R contiguous: thread -> (k = tid / 8, 8 consecutive rows)
template <typename T>
__device__ __forceinline__ typename Elem<T>::x8 bmm_fetch(const T* base, int64_t ld, int kmajor, int R, int K, int r0, int k0, int vec, int tid) {
    typedef typename Elem<T>::x8 X8;
    X8 v;
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = static_cast<T>(0.f);
    if (!kmajor) {
        const int r = r0 + (tid >> 2), k = k0 + (tid & 3) * 8;
        if (r < R && k < K) {
            const T* p = base + (int64_t)r * ld + k;
            if (vec) {
                v = *reinterpret_cast<const X8*>(p);
#pragma unroll
                for (int i = 0; i < 8; ++i) if (k + i >= K) v[i] = static_cast<T>(0.f);
            } else {
#pragma unroll
                for (int i = 0; i < 8; ++i) if (k + i < K) v[i] = p[i];
            }
        }
    } else {
        const int k = k0 + (tid >> 3), r = r0 + (tid & 7) * 8;
        if (k < K && r < R) {
            const T* p = base + (int64_t)k * ld + r;
            if (vec) {
                v = *reinterpret_cast<const X8*>(p);
#pragma unroll
                for (int i = 0; i < 8; ++i) if (r + i >= R) v[i] = static_cast<T>(0.f);
            } else {
#pragma unroll
                for (int i = 0; i < 8; ++i) if (r + i < R) v[i] = p[i];
            }
        }
    }
    return v;
}

// LDS images of one operand tile (BT rows of the M / N extent x 32 of K):
//   stored (R, K) [kmajor 0]: [BT][40] - a row's 32 k-elements + 8 pad; the MFMA fragment (row r15, k = 8g .. 8g+7) is one ds_read_b128
//   stored (K, R) [kmajor 1]: [32][BT + 16] - as it comes from memory (16-byte writes, no scatter); the fragment is two
//     ds_read_b64_tr_b16 (hardware transpose: a 16-lane group reads 4 k-rows x 16 r-columns and lane i receives column i)
typedef __attribute__((address_space(3))) s16x4* bmm_lds_s16x4_ptr;

template <typename T, int BT>
__device__ __forceinline__ void bmm_stash(char* tile, typename Elem<T>::x8 v, int kmajor, int tid, int h) {
    typedef typename Elem<T>::x8 X8;
    if (!kmajor) *reinterpret_cast<X8*>(tile + (((tid >> 2) + h * 64) * kBmmLd + (tid & 3) * 8) * 2) = v;
    else *reinterpret_cast<X8*>(tile + ((tid >> 3) * (BT + 16) + (tid & 7) * 8 + h * 64) * 2) = v;
}

template <typename T, int BT>
__device__ __forceinline__ typename Elem<T>::x8 bmm_frag(const char* tile, int kmajor, int row0, int lane) {
    typedef typename Elem<T>::x8 X8;
    const int r15 = lane & 15, g = lane >> 4;
    if (!kmajor) return *reinterpret_cast<const X8*>(tile + ((row0 + r15) * kBmmLd + g * 8) * 2);
    const int q = r15 >> 2, p = r15 & 3;
    const char* base = tile + ((8 * g + q) * (BT + 16) + row0 + 4 * p) * 2;
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((bmm_lds_s16x4_ptr)(base));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((bmm_lds_s16x4_ptr)(base + 4 * (BT + 16) * 2));
    s16x8 both;
    both.s0 = lo.x; both.s1 = lo.y; both.s2 = lo.z; both.s3 = lo.w;
    both.s4 = hi.x; both.s5 = hi.y; both.s6 = hi.z; both.s7 = hi.w;
    return __builtin_bit_cast(X8, both);
}

// F = MFMA tiles per wave and dimension: F = 2 -> 64 x 64 block tile (small products: one head of self-attention is 32 x 32),
// F = 4 -> 128 x 128 (16 MFMAs per wave between two barriers: the weight gradients and the stacked cross-attention products)
// KS = 32-wide K slabs per step (each its own LDS image).  KS = 2 (twice the bytes in flight per block, half the barriers) costs a
// third of the occupancy (141 VGPRs, 80 KB LDS) and measured 1.8x SLOWER on the weight gradients: both instantiations use KS = 1.
template <typename T, typename TO, int F, int KS>
__global__ __launch_bounds__(256, 3) void bmm_mfma_kernel(BmmArgs a) {      // three blocks per CU: 168 registers per lane
    typedef typename Elem<T>::x8 X8;
    constexpr int BT = 32 * F, H = F / 2;                   // block tile extent; 64-row fetch slabs per operand
    constexpr int kImg = (BT * kBmmLd > 32 * (BT + 16) ? BT * kBmmLd : 32 * (BT + 16)) * 2;
    constexpr int BK = 32 * KS;
    __shared__ __attribute__((aligned(16))) char As[2][KS][kImg]; // two image sets per operand: tile k+1 is written while tile k is read,
    __shared__ __attribute__((aligned(16))) char Bs[2][KS][kImg]; // one barrier per K step
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r15 = lane & 15, g = lane >> 4, wm = wave >> 1, wn = wave & 1;
    // XCD-aware block order: workgroups are dealt to the 8 XCDs round-robin by linear id, each XCD has its own 4 MB L2.  When the
    // batch count is a multiple of 8, all tiles of one batch item (one row chunk of a weight gradient: 1.5 MB of dy and x; one
    // (candidate, head) of the attention products) run on ONE XCD, so its operands are fetched into that L2 once instead of
    // streaming from the Infinity Cache into all eight (measured: weight gradients +8 %, the attention products +10 .. 30 %).
    // (Tried and dropped: a second register set for a two-tile-deep prefetch - hipcc then duplicates the accumulators,
    //  240 VGPRs + 128 AGPRs, one wave per SIMD, 2.5x slower.)
    int bx = blockIdx.x, by = blockIdx.y, bz = blockIdx.z;
    if (gridDim.z % 8 == 0) {
        const int tiles = gridDim.x * gridDim.y;
        const int lin = bx + gridDim.x * (by + gridDim.y * bz);
        const int j = lin >> 3, t = j % tiles;
        bz = (lin & 7) + 8 * (j / tiles);
        bx = t % gridDim.x;
        by = t / gridDim.x;
    }
    const int z1 = bz / a.nb2, z2 = bz % a.nb2;
    const T* A = reinterpret_cast<const T*>(a.A) + z1 * a.sA1 + z2 * a.sA2;
    const T* B = reinterpret_cast<const T*>(a.B) + z1 * a.sB1 + z2 * a.sB2;
    TO* C = reinterpret_cast<TO*>(a.C) + z1 * a.sC1 + z2 * a.sC2;
    const int m0 = by * BT, n0 = bx * BT;
    const int ka = a.ta, kb = !a.tb;                        // A stored (K, M) when ta = 1; B stored (K, N) when tb = 0
    f32x4 acc[F][F];
#pragma unroll
    for (int i = 0; i < F; ++i)
#pragma unroll
        for (int j = 0; j < F; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    // interior tiles: one 16-byte load per slab from a pointer computed once (the generic fetch re-derives a 64-bit address and
    // masks eight elements per load - more VALU issue per K step than the step's MFMAs take)
    const T* pa[H]; const T* pb[H];
    bool fa[H], fb[H];
#pragma unroll
    for (int h = 0; h < H; ++h) {
        pa[h] = ka ? A + (int64_t)(tid >> 3) * a.lda + m0 + h * 64 + (tid & 7) * 8 : A + (int64_t)(m0 + h * 64 + (tid >> 2)) * a.lda + (tid & 3) * 8;
        pb[h] = kb ? B + (int64_t)(tid >> 3) * a.ldb + n0 + h * 64 + (tid & 7) * 8 : B + (int64_t)(n0 + h * 64 + (tid >> 2)) * a.ldb + (tid & 3) * 8;
        fa[h] = a.vecA && (ka ? m0 + h * 64 + (tid & 7) * 8 + 8 <= a.M : m0 + h * 64 + (tid >> 2) < a.M);
        fb[h] = a.vecB && (kb ? n0 + h * 64 + (tid & 7) * 8 + 8 <= a.N : n0 + h * 64 + (tid >> 2) < a.N);
    }
    const int64_t stepA = ka ? a.lda : 1, stepB = kb ? a.ldb : 1;       // elements per unit of k
    auto fetch_a = [&](int h, int k) -> X8 {
        if (fa[h] && k + 32 <= a.K) return *reinterpret_cast<const X8*>(pa[h] + k * stepA);
        return bmm_fetch<T>(A, a.lda, ka, a.M, a.K, m0 + h * 64, k, a.vecA, tid);
    };
    auto fetch_b = [&](int h, int k) -> X8 {
        if (fb[h] && k + 32 <= a.K) return *reinterpret_cast<const X8*>(pb[h] + k * stepB);
        return bmm_fetch<T>(B, a.ldb, kb, a.N, a.K, n0 + h * 64, k, a.vecB, tid);
    };
    X8 ra[KS][H], rb[KS][H];
#pragma unroll
    for (int sl = 0; sl < KS; ++sl)
#pragma unroll
        for (int h = 0; h < H; ++h) {
            ra[sl][h] = fetch_a(h, sl * 32);
            rb[sl][h] = fetch_b(h, sl * 32);
        }
#pragma unroll
    for (int sl = 0; sl < KS; ++sl)
#pragma unroll
        for (int h = 0; h < H; ++h) {
            bmm_stash<T, BT>(As[0][sl], ra[sl][h], ka, tid, h);
            bmm_stash<T, BT>(Bs[0][sl], rb[sl][h], kb, tid, h);
        }
    if (BK < a.K) {
#pragma unroll
        for (int sl = 0; sl < KS; ++sl)
#pragma unroll
            for (int h = 0; h < H; ++h) {
                ra[sl][h] = fetch_a(h, BK + sl * 32);
                rb[sl][h] = fetch_b(h, BK + sl * 32);
            }
    }
    __syncthreads();
    int buf = 0;
    for (int k0 = 0; k0 < a.K; k0 += BK, buf ^= 1) {
#pragma unroll
        for (int sl = 0; sl < KS; ++sl) {
            X8 af[F], bf[F];
#pragma unroll
            for (int i = 0; i < F; ++i) {
                af[i] = bmm_frag<T, BT>(As[buf][sl], ka, wm * 16 * F + i * 16, lane);
                bf[i] = bmm_frag<T, BT>(Bs[buf][sl], kb, wn * 16 * F + i * 16, lane);
            }
            if (sl == 0 && k0 + BK < a.K) {                  // tile k+1 (in registers since the last step) into the other image set,
#pragma unroll
                for (int s2 = 0; s2 < KS; ++s2)
#pragma unroll
                    for (int h = 0; h < H; ++h) {
                        bmm_stash<T, BT>(As[buf ^ 1][s2], ra[s2][h], ka, tid, h);
                        bmm_stash<T, BT>(Bs[buf ^ 1][s2], rb[s2][h], kb, tid, h);
                    }
                if (k0 + 2 * BK < a.K) {                     // tile k+2's loads fly under this tile's MFMAs
#pragma unroll
                    for (int s2 = 0; s2 < KS; ++s2)
#pragma unroll
                        for (int h = 0; h < H; ++h) {
                            ra[s2][h] = fetch_a(h, k0 + 2 * BK + s2 * 32);
                            rb[s2][h] = fetch_b(h, k0 + 2 * BK + s2 * 32);
                        }
                }
            }
#pragma unroll
            for (int mi = 0; mi < F; ++mi)
#pragma unroll
                for (int ni = 0; ni < F; ++ni) acc[mi][ni] = Elem<T>::mfma16(af[mi], bf[ni], acc[mi][ni]);
        }
        __syncthreads();                                     // image set buf^1 complete; set buf free for the step after next
    }
    // lane (r15, g) holds C[m = 4 g + jj][n = r15] of each 16 x 16 tile
#pragma unroll
    for (int mi = 0; mi < F; ++mi)
#pragma unroll
        for (int ni = 0; ni < F; ++ni) {
            const int n = n0 + wn * 16 * F + ni * 16 + r15;
            if (n >= a.N) continue;
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
                const int m = m0 + wm * 16 * F + mi * 16 + g * 4 + jj;
                if (m >= a.M) continue;
                TO* cp = C + (int64_t)m * a.ldc + n;
                if constexpr (sizeof(TO) == 4) {
                    if (a.accumulate == 2) { atomicAdd(cp, a.alpha * acc[mi][ni][jj]); continue; }     // split-K batches summing into ONE C
                }
                const float v = a.alpha * acc[mi][ni][jj] + (a.accumulate ? static_cast<float>(*cp) : 0.f);
                *cp = static_cast<TO>(v);
            }
        }
}

// ---- row softmax with additive key mask and dropout: P = softmax(S * scale + mask), Pd = dropout(P) ----------------------
// one wave per row; S fp32 (rows, cols) with leading dimension lds_; mask fp32 (cols) per group of `rows_per_mask` rows or NULL.
// Rows up to 64 * kSmCols columns are held in registers (one pass over S / dPd); longer rows take the re-reading loops.
constexpr int kSmCols = 12;      // 768 columns: the 577 / 197 image tokens and any caption length
template <typename T>
__global__ __launch_bounds__(256) void softmax_fwd_kernel(const float* S, int64_t ld_s, const float* mask, int64_t rows_per_mask, int64_t ld_mask,
                                                          T* P, T* Pd, int64_t ld_p, int64_t rows, int cols, float scale, float p_drop, uint64_t seed) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float* sr = S + row * ld_s;
    const float* mr = mask ? mask + (row / rows_per_mask) * ld_mask : nullptr;
    const float keep = 1.0f / (1.0f - p_drop);
    if (cols <= 64 * kSmCols) {
        float v[kSmCols];
        float mx = -INFINITY;
#pragma unroll
        for (int i = 0; i < kSmCols; ++i) {
            const int c = lane + i * 64;
            v[i] = c < cols ? fmaf(sr[c], scale, mr ? mr[c] : 0.f) : -INFINITY;
            mx = fmaxf(mx, v[i]);
        }
        mx = wave_max(mx);
        float sum = 0.f;
#pragma unroll
        for (int i = 0; i < kSmCols; ++i) { v[i] = (lane + i * 64 < cols) ? __expf(v[i] - mx) : 0.f; sum += v[i]; }
        const float inv = 1.0f / wave_sum(sum);
#pragma unroll
        for (int i = 0; i < kSmCols; ++i) {
            const int c = lane + i * 64;
            if (c < cols) {
                const float p = v[i] * inv;
                P[row * ld_p + c] = static_cast<T>(p);
                const bool kept = p_drop <= 0.f || uniform01(seed, (uint64_t)row * (uint64_t)cols + c) >= p_drop;
                Pd[row * ld_p + c] = static_cast<T>(kept ? p * (p_drop > 0.f ? keep : 1.0f) : 0.f);
            }
        }
        return;
    }
    float mx = -INFINITY;
    for (int c = lane; c < cols; c += 64) mx = fmaxf(mx, fmaf(sr[c], scale, mr ? mr[c] : 0.f));
    mx = wave_max(mx);
    float sum = 0.f;
    for (int c = lane; c < cols; c += 64) sum += __expf(fmaf(sr[c], scale, mr ? mr[c] : 0.f) - mx);
    sum = wave_sum(sum);
    const float inv = 1.0f / sum;
    for (int c = lane; c < cols; c += 64) {
        const float p = __expf(fmaf(sr[c], scale, mr ? mr[c] : 0.f) - mx) * inv;
        P[row * ld_p + c] = static_cast<T>(p);
        const bool kept = p_drop <= 0.f || uniform01(seed, (uint64_t)row * (uint64_t)cols + c) >= p_drop;
        Pd[row * ld_p + c] = static_cast<T>(kept ? p * (p_drop > 0.f ? keep : 1.0f) : 0.f);
    }
}

// dS = scale * P * (dP - sum_c dP * P) with dP = dropout-backward(dPd): rows as in the forward, dPd fp32, dS 16-bit
template <typename T>
__global__ __launch_bounds__(256) void softmax_bwd_kernel(const T* P, int64_t ld_p, const float* dPd, int64_t ld_d, T* dS, int64_t ld_ds,
                                                          int64_t rows, int cols, float scale, float p_drop, uint64_t seed) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float keep = p_drop > 0.f ? 1.0f / (1.0f - p_drop) : 1.0f;
    if (cols <= 64 * kSmCols) {
        float dp[kSmCols], pv[kSmCols];
        float dot = 0.f;
#pragma unroll
        for (int i = 0; i < kSmCols; ++i) {
            const int c = lane + i * 64;
            dp[i] = 0.f; pv[i] = 0.f;
            if (c < cols) {
                const bool kept = p_drop <= 0.f || uniform01(seed, (uint64_t)row * (uint64_t)cols + c) >= p_drop;
                dp[i] = kept ? dPd[row * ld_d + c] * keep : 0.f;
                pv[i] = static_cast<float>(P[row * ld_p + c]);
                dot += dp[i] * pv[i];
            }
        }
        dot = wave_sum(dot);
#pragma unroll
        for (int i = 0; i < kSmCols; ++i) {
            const int c = lane + i * 64;
            if (c < cols) dS[row * ld_ds + c] = static_cast<T>(scale * pv[i] * (dp[i] - dot));
        }
        return;
    }
    float dot = 0.f;
    for (int c = lane; c < cols; c += 64) {
        const bool kept = p_drop <= 0.f || uniform01(seed, (uint64_t)row * (uint64_t)cols + c) >= p_drop;
        const float dp = kept ? dPd[row * ld_d + c] * keep : 0.f;
        dot += dp * static_cast<float>(P[row * ld_p + c]);
    }
    dot = wave_sum(dot);
    for (int c = lane; c < cols; c += 64) {
        const bool kept = p_drop <= 0.f || uniform01(seed, (uint64_t)row * (uint64_t)cols + c) >= p_drop;
        const float dp = kept ? dPd[row * ld_d + c] * keep : 0.f;
        dS[row * ld_ds + c] = static_cast<T>(scale * static_cast<float>(P[row * ld_p + c]) * (dp - dot));
    }
}

// ---- LayerNorm backward: y = (x - mean) * rstd * gamma + beta over `cols`; x fp32 (the saved pre-LN sum) -----------------
// dx fp32 (written), dgamma / dbeta fp32 (atomically accumulated: zero them first).  One wave per row, 32 rows per block: the
// per-column sums of those rows are kept in registers (cols <= 64 * kLnCols) and added once per block, not once per row.
constexpr int kLnCols = 16;      // columns per lane: cols <= 1024
__global__ __launch_bounds__(256) void layernorm_bwd_kernel(const float* x, const float* gamma, const float* dy, float* dx, float* dgamma, float* dbeta,
                                                            int64_t rows, int cols, float eps) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float pg[kLnCols], pb[kLnCols];
#pragma unroll
    for (int i = 0; i < kLnCols; ++i) { pg[i] = 0.f; pb[i] = 0.f; }
    for (int it = 0; it < 8; ++it) {
        const int64_t row = (int64_t)blockIdx.x * 32 + it * 4 + wave;
        if (row >= rows) break;
        const float* xr = x + row * cols;
        const float* dr = dy + row * cols;
        float xv[kLnCols], dv[kLnCols];
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < kLnCols; ++i) {
            const int c = lane + i * 64;
            xv[i] = c < cols ? xr[c] : 0.f;
            dv[i] = c < cols ? dr[c] : 0.f;
            s += xv[i];
        }
        const float mean = wave_sum(s) / cols;
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < kLnCols; ++i) { const float d = (lane + i * 64 < cols) ? xv[i] - mean : 0.f; q += d * d; }
        const float rstd = rsqrtf(wave_sum(q) / cols + eps);
        float a = 0.f, b = 0.f;           // mean(dy*gamma), mean(dy*gamma*xhat)
#pragma unroll
        for (int i = 0; i < kLnCols; ++i) {
            const int c = lane + i * 64;
            if (c < cols) {
                const float xh = (xv[i] - mean) * rstd, gq = dv[i] * gamma[c];
                a += gq; b += gq * xh;
            }
        }
        a = wave_sum(a) / cols; b = wave_sum(b) / cols;
#pragma unroll
        for (int i = 0; i < kLnCols; ++i) {
            const int c = lane + i * 64;
            if (c < cols) {
                const float xh = (xv[i] - mean) * rstd, gq = dv[i] * gamma[c];
                dx[row * cols + c] = rstd * (gq - a - xh * b);
                pg[i] += dv[i] * xh;
                pb[i] += dv[i];
            }
        }
    }
    __shared__ float red[2][4][64 * kLnCols];
#pragma unroll
    for (int i = 0; i < kLnCols; ++i) { red[0][wave][lane + i * 64] = pg[i]; red[1][wave][lane + i * 64] = pb[i]; }
    __syncthreads();
    for (int c = threadIdx.x; c < cols; c += 256) {
        atomicAdd(dgamma + c, red[0][0][c] + red[0][1][c] + red[0][2][c] + red[0][3][c]);
        atomicAdd(dbeta + c, red[1][0][c] + red[1][1][c] + red[1][2][c] + red[1][3][c]);
    }
}

// ---- elementwise ------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ float gelu_exact(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752f)); }
__device__ __forceinline__ float gelu_grad(float x) {
    return 0.5f * (1.0f + erff(x * 0.70710678118654752f)) + x * 0.3989422804014327f * __expf(-0.5f * x * x);
}
// mode 0: y = gelu(z); 1: dz = dy * gelu'(z); 2: y = relu(z); 3: dz = dy * (z > 0); 4: y = dropout(z) [dy unused]; 5: y = z + dy (add);
// 6: y = p_drop * z (scale by the factor passed in p_drop)
// (Tried in round 4: the logistic-fit GELU of the inference epilogue for 16-bit outputs - the 2R x 3072 pass went 58 -> 54 us, it is bound
//  by its 200 MB of traffic, not by erff - at the price of moving train197's worst same-piece gradient 1.5e-2 -> 1.75e-2: dropped.)
__device__ __forceinline__ float eltwise_op(float v, float d, int mode, float p_drop, float keep, uint64_t seed, int64_t i) {
    switch (mode) {
        case 0: return gelu_exact(v);
        case 1: return d * gelu_grad(v);
        case 2: return fmaxf(v, 0.f);
        case 3: return v > 0.f ? d : 0.f;
        case 4: return (p_drop <= 0.f || uniform01(seed, (uint64_t)i) >= p_drop) ? v * keep : 0.f;
        case 5: return v + d;
        default: return v * p_drop;
    }
}

template <typename TZ, typename TO>
__global__ __launch_bounds__(256) void eltwise_kernel(const TZ* z, const float* dy, TO* out, int64_t n, int mode, float p_drop, uint64_t seed, int vec) {
    const int64_t i0 = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;         // four consecutive elements per thread
    if (i0 >= n) return;
    const float keep = p_drop > 0.f ? 1.0f / (1.0f - p_drop) : 1.0f;
    const bool use_dy = mode == 1 || mode == 3 || mode == 5;
    if (vec && i0 + 4 <= n) {                                                   // 16-byte fp32 accesses (host checked the alignment)
        float v[4], d[4] = {0.f, 0.f, 0.f, 0.f}, r[4];
        if constexpr (sizeof(TZ) == 4) {
            const float4 z4 = *reinterpret_cast<const float4*>(z + i0);
            v[0] = z4.x; v[1] = z4.y; v[2] = z4.z; v[3] = z4.w;
        } else {
            typedef __attribute__((ext_vector_type(4))) TZ tz4;                // one 8-byte load (i0 is a multiple of 4, the base 16-byte aligned)
            const tz4 z4 = *reinterpret_cast<const tz4*>(z + i0);
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = static_cast<float>(z4[e]);
        }
        if (use_dy) {
            const float4 d4 = *reinterpret_cast<const float4*>(dy + i0);
            d[0] = d4.x; d[1] = d4.y; d[2] = d4.z; d[3] = d4.w;
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) r[e] = eltwise_op(v[e], d[e], mode, p_drop, keep, seed, i0 + e);
        if constexpr (sizeof(TO) == 4) {
            *reinterpret_cast<float4*>(out + i0) = make_float4(r[0], r[1], r[2], r[3]);
        } else {
            typedef __attribute__((ext_vector_type(4))) TO to4;
            to4 o = {static_cast<TO>(r[0]), static_cast<TO>(r[1]), static_cast<TO>(r[2]), static_cast<TO>(r[3])};
            *reinterpret_cast<to4*>(out + i0) = o;
        }
        return;
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int64_t i = i0 + e;
        if (i >= n) break;
        out[i] = static_cast<TO>(eltwise_op(static_cast<float>(z[i]), use_dy ? dy[i] : 0.f, mode, p_drop, keep, seed, i));
    }
}

// column sums: out[c] += sum_r x[r][c] (fp32 x, atomics: zero `out` first)
__global__ __launch_bounds__(256) void colsum_kernel(const float* x, int64_t ld, float* out, int64_t rows, int cols, int64_t rows_per_block) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= cols) return;
    const int64_t r0 = (int64_t)blockIdx.y * rows_per_block;
    const int64_t r1 = r0 + rows_per_block < rows ? r0 + rows_per_block : rows;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;          // four independent loads in flight per thread
    int64_t r = r0;
    for (; r + 3 < r1; r += 4) {
        s0 += x[r * ld + c]; s1 += x[(r + 1) * ld + c]; s2 += x[(r + 2) * ld + c]; s3 += x[(r + 3) * ld + c];
    }
    for (; r < r1; ++r) s0 += x[r * ld + c];
    atomicAdd(out + c, (s0 + s1) + (s2 + s3));
}

// embedding backward: dword[ids[r]][c] += dy[r][c], dpos[r % L][c] += dy[r][c]  (fp32, atomics: zero first)
__global__ __launch_bounds__(256) void embed_bwd_kernel(const int64_t* ids, const float* dy, float* dword, float* dpos, int64_t rows, int L, int cols) {
    const int64_t r = blockIdx.x;
    if (r >= rows) return;
    const int64_t id = ids[r];
    for (int c = threadIdx.x; c < cols; c += 256) {
        const float g = dy[r * cols + c];
        atomicAdd(dword + id * cols + c, g);
        atomicAdd(dpos + (r % L) * cols + c, g);
    }
}

// AdamW (decoupled weight decay, torch.optim.AdamW semantics): in place on fp32 param / moments
__global__ __launch_bounds__(256) void adamw_kernel(float* p, const float* g, float* m, float* v, int64_t n, float lr, float b1, float b2, float eps,
                                                    float wd, float bc1, float bc2) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float gi = g[i];
    const float mi = b1 * m[i] + (1.f - b1) * gi, vi = b2 * v[i] + (1.f - b2) * gi * gi;
    m[i] = mi; v[i] = vi;
    float w = p[i] * (1.f - lr * wd);
    w -= lr * (mi / bc1) / (sqrtf(vi / bc2) + eps);
    p[i] = w;
}

// ---- the optimizer step without a host read (round 6) -------------------------------------------------------------------------------
// state (device, 8 x 32 bit): [0] found_inf of the step in flight, [1] applied steps t, [2] skipped steps, [3] / [4] the fp32 bias
// corrections 1 - beta^t of the step in flight.  cir_grads_check ORs into [0]; cir_adamw_begin turns the flag into t / skipped and the
// corrections; cir_adamw_step_dev returns at once when the flag is set - GradScaler.step's found_inf skip (stage2_train.py:215-218) with
// the decision taken where the gradients are.
__global__ __launch_bounds__(256) void grads_check_kernel(float* g, int64_t n4, int64_t n, float scale, int* state) {
    int bad = 0;
    const int64_t stride = (int64_t)gridDim.x * 256;
    auto nf = [](float f) { return (int)((__float_as_uint(f) & 0x7f800000u) == 0x7f800000u); };      // (finite <=> exponent field not all ones)
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += 4 * stride) {
        float4 x[4];
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (i + u * stride < n4) x[u] = reinterpret_cast<float4*>(g)[i + u * stride];
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (i + u * stride < n4) {
                if (scale != 1.f) {
                    x[u].x *= scale; x[u].y *= scale; x[u].z *= scale; x[u].w *= scale;
                    reinterpret_cast<float4*>(g)[i + u * stride] = x[u];
                }
                bad |= nf(x[u].x) | nf(x[u].y) | nf(x[u].z) | nf(x[u].w);
            }
    }
    if (blockIdx.x == 0 && threadIdx.x < (int)(n - 4 * n4)) {           // tail of a length that is not a multiple of 4
        float x = g[4 * n4 + threadIdx.x];
        if (scale != 1.f) { x *= scale; g[4 * n4 + threadIdx.x] = x; }
        bad |= nf(x);
    }
    if (__any(bad) && (threadIdx.x & 63) == 0) atomicOr(state, 1);
}

__global__ void adamw_begin_kernel(int* state, float b1, float b2) {
    if (threadIdx.x || blockIdx.x) return;
    if (state[0]) { state[2] += 1; return; }
    const int t = state[1] + 1;
    state[1] = t;
    reinterpret_cast<float*>(state)[3] = (float)(1.0 - pow((double)b1, (double)t));
    reinterpret_cast<float*>(state)[4] = (float)(1.0 - pow((double)b2, (double)t));
}

template <typename T16>
__global__ __launch_bounds__(256) void adamw_dev_kernel(float* p, const float* g, float* m, float* v, int64_t n4, int64_t n, float lr, float b1, float b2,
                                                        float eps, float wd, const int* state, T16* p16) {
    if (state[0]) return;                                               // non-finite gradients somewhere in this step: nothing is applied
    const float bc1 = reinterpret_cast<const float*>(state)[3], bc2 = reinterpret_cast<const float*>(state)[4];
    auto upd = [&](float& pi, float gi, float& mi, float& vi) {
        mi = b1 * mi + (1.f - b1) * gi;
        vi = b2 * vi + (1.f - b2) * gi * gi;
        float w = pi * (1.f - lr * wd);
        w -= lr * (mi / bc1) / (sqrtf(vi / bc2) + eps);
        pi = w;
    };
    // two float4 per thread, 256 apart: all eight 16-byte loads of a thread are in flight before the first update
    const int64_t i0 = (int64_t)blockIdx.x * 512 + threadIdx.x;
    float4 pp[2], mm[2], vv[2], gg[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int64_t i = i0 + 256 * u;
        if (i < n4) {
            pp[u] = reinterpret_cast<float4*>(p)[i]; mm[u] = reinterpret_cast<float4*>(m)[i]; vv[u] = reinterpret_cast<float4*>(v)[i];
            gg[u] = reinterpret_cast<const float4*>(g)[i];
        }
    }
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int64_t i = i0 + 256 * u;
        if (i < n4) {
            upd(pp[u].x, gg[u].x, mm[u].x, vv[u].x); upd(pp[u].y, gg[u].y, mm[u].y, vv[u].y);
            upd(pp[u].z, gg[u].z, mm[u].z, vv[u].z); upd(pp[u].w, gg[u].w, mm[u].w, vv[u].w);
            reinterpret_cast<float4*>(p)[i] = pp[u]; reinterpret_cast<float4*>(m)[i] = mm[u]; reinterpret_cast<float4*>(v)[i] = vv[u];
            if (p16) {
                u32x2 h;
                h.x = pack2<T16>(pp[u].x, pp[u].y); h.y = pack2<T16>(pp[u].z, pp[u].w);
                reinterpret_cast<u32x2*>(p16)[i] = h;
            }
        }
    }
    if (blockIdx.x == 0 && threadIdx.x < (int)(n - 4 * n4)) {
        const int64_t j = 4 * n4 + threadIdx.x;
        float pi = p[j], mi = m[j], vi = v[j];
        upd(pi, g[j], mi, vi);
        p[j] = pi; m[j] = mi; v[j] = vi;
        if (p16) p16[j] = static_cast<T16>(pi);
    }
}

}  // namespace cir

using namespace cir;

extern "C" int cir_grads_check(float* g, int64_t n, float scale, int32_t* state, void* stream) {
    CIR_CHECK_PTR(g); CIR_CHECK_PTR(state);
    if (n <= 0) return CIR_EINVAL;
    if (!cir_aligned16(g)) return CIR_EALIGN;
    const int64_t n4 = n / 4;
    const unsigned blocks = (unsigned)std::min<int64_t>(std::max<int64_t>((n4 + 1023) / 1024, 1), 256 * 32);
    hipLaunchKernelGGL(grads_check_kernel, dim3(blocks), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), g, n4, n, scale, state);
    CIR_LAUNCH_RESULT();
}

extern "C" int cir_adamw_begin(int32_t* state, float beta1, float beta2, void* stream) {
    CIR_CHECK_PTR(state);
    hipLaunchKernelGGL(adamw_begin_kernel, dim3(1), dim3(64), 0, reinterpret_cast<hipStream_t>(stream), state, beta1, beta2);
    CIR_LAUNCH_RESULT();
}

extern "C" int cir_adamw_step_dev(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2, float eps,
                                  float weight_decay, const int32_t* state, void* p16, int dtype16, void* stream) {
    CIR_CHECK_PTR(p); CIR_CHECK_PTR(g); CIR_CHECK_PTR(m); CIR_CHECK_PTR(v); CIR_CHECK_PTR(state);
    if (n <= 0) return CIR_EINVAL;
    if (p16 && dtype16 != CIR_BF16 && dtype16 != CIR_F16) return CIR_EDTYPE;
    if (!cir_aligned16(p) || !cir_aligned16(g) || !cir_aligned16(m) || !cir_aligned16(v) || (p16 && (reinterpret_cast<uintptr_t>(p16) & 7))) return CIR_EALIGN;
    const int64_t n4 = n / 4;
    const dim3 grid((unsigned)std::max<int64_t>((n4 + 511) / 512, 1)), block(256);
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (p16 && dtype16 == CIR_BF16) hipLaunchKernelGGL((adamw_dev_kernel<__bf16>), grid, block, 0, s, p, g, m, v, n4, n, lr, beta1, beta2, eps, weight_decay, state, reinterpret_cast<__bf16*>(p16));
    else hipLaunchKernelGGL((adamw_dev_kernel<_Float16>), grid, block, 0, s, p, g, m, v, n4, n, lr, beta1, beta2, eps, weight_decay, state, reinterpret_cast<_Float16*>(p16));
    CIR_LAUNCH_RESULT();
}

extern "C" int cir_transpose16(const void* src, void* dst, int rows, int cols, int64_t ld_src, int64_t ld_dst, int batch, int64_t s_src,
                               int64_t s_dst, int dtype, void* stream) {
    CIR_CHECK_PTR(src); CIR_CHECK_PTR(dst);
    if (rows <= 0 || cols <= 0 || batch <= 0) return CIR_EINVAL;
    if (dtype != CIR_BF16 && dtype != CIR_F16) return CIR_EDTYPE;
    dim3 grid((cols + 31) / 32, (rows + 31) / 32, batch), block(256);
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    // (bf16 and fp16 are both 16-bit payloads: one instantiation moves either)
    hipLaunchKernelGGL((transpose_kernel<unsigned short>), grid, block, 0, s, reinterpret_cast<const unsigned short*>(src),
                       reinterpret_cast<unsigned short*>(dst), rows, cols, ld_src, ld_dst, s_src, s_dst);
    CIR_LAUNCH_RESULT();
}

extern "C" int cir_transpose16_multi(const void* src, void* dst, const int64_t* table, int count, int64_t total_tiles, int dtype, void* stream) {
    CIR_CHECK_PTR(src); CIR_CHECK_PTR(dst); CIR_CHECK_PTR(table);
    if (count <= 0 || total_tiles <= 0 || total_tiles > 0x7fffffffLL) return CIR_EINVAL;
    if (dtype != CIR_BF16 && dtype != CIR_F16) return CIR_EDTYPE;
    hipLaunchKernelGGL((transpose_multi_kernel<unsigned short>), dim3((unsigned)total_tiles), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                       reinterpret_cast<const unsigned short*>(src), reinterpret_cast<unsigned short*>(dst), table, count);
    CIR_LAUNCH_RESULT();
}

extern "C" int cir_bmm(const void* A, const void* B, void* C, int M, int N, int K, int64_t lda, int64_t ldb, int64_t ldc, int trans_a, int trans_b,
                       int nb1, int nb2, int64_t sA1, int64_t sA2, int64_t sB1, int64_t sB2, int64_t sC1, int64_t sC2, float alpha, int accumulate,
                       int in_dtype, int out_dtype, void* stream) {
    CIR_CHECK_PTR(A); CIR_CHECK_PTR(B); CIR_CHECK_PTR(C);
    if (M <= 0 || N <= 0 || K <= 0 || nb1 <= 0 || nb2 <= 0) return CIR_EINVAL;
    if ((int64_t)nb1 * nb2 > 65535) return CIR_ESHAPE;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (accumulate < 0 || accumulate > 2) return CIR_EINVAL;
    if (in_dtype == CIR_F32) {                               // fp32 operands: the plain kernel
        if (out_dtype != CIR_F32) return CIR_EDTYPE;
        if (accumulate == 2) return CIR_EDTYPE;
        dim3 grid((N + 15) / 16, (M + 15) / 16, nb1 * nb2), block(256);
        hipLaunchKernelGGL((bmm_kernel<float, float>), grid, block, 0, s, reinterpret_cast<const float*>(A), reinterpret_cast<const float*>(B),
                           reinterpret_cast<float*>(C), M, N, K, lda, ldb, ldc, trans_a, trans_b, nb2, sA1, sA2, sB1, sB2, sC1, sC2, alpha, accumulate);
        CIR_LAUNCH_RESULT();
    }
    if (in_dtype != CIR_BF16 && in_dtype != CIR_F16) return CIR_EDTYPE;
    if (out_dtype != CIR_F32 && out_dtype != in_dtype) return CIR_EDTYPE;
    if (accumulate == 2 && out_dtype != CIR_F32) return CIR_EDTYPE;
    auto vec_ok = [](const void* p, int64_t ld, int64_t s1, int64_t s2) {
        return (reinterpret_cast<uintptr_t>(p) % 16 == 0 && ld % 8 == 0 && s1 % 8 == 0 && s2 % 8 == 0) ? 1 : 0;
    };
    BmmArgs a{A, B, C, M, N, K, lda, ldb, ldc, trans_a ? 1 : 0, trans_b ? 1 : 0, nb2, sA1, sA2, sB1, sB2, sC1, sC2, alpha, accumulate,
              vec_ok(A, lda, sA1, sA2), vec_ok(B, ldb, sB1, sB2)};
    dim3 block(256);
    const bool big = M > 64 && N > 64;                       // 128 x 128 tiles unless one extent fits a 64 tile anyway
    const int bt = big ? 128 : 64;
    dim3 grid((N + bt - 1) / bt, (M + bt - 1) / bt, nb1 * nb2);
#define CIR_BMM_F(T, TO) do { if (big) hipLaunchKernelGGL((bmm_mfma_kernel<T, TO, 4, 1>), grid, block, 0, s, a); \
                              else hipLaunchKernelGGL((bmm_mfma_kernel<T, TO, 2, 1>), grid, block, 0, s, a); } while (0)
    if (in_dtype == CIR_BF16) {
        if (out_dtype == CIR_F32) CIR_BMM_F(__bf16, float); else CIR_BMM_F(__bf16, __bf16);
    } else {
        if (out_dtype == CIR_F32) CIR_BMM_F(_Float16, float); else CIR_BMM_F(_Float16, _Float16);
    }
#undef CIR_BMM_F
    CIR_LAUNCH_RESULT();
}

extern "C" int cir_softmax_dropout(const float* S, int64_t ld_s, const float* mask, int64_t rows_per_mask, int64_t ld_mask, void* P, void* Pd,
                                   int64_t ld_p, int64_t rows, int cols, float scale, float p_drop, uint64_t seed, int dtype, void* stream) {
    CIR_CHECK_PTR(S); CIR_CHECK_PTR(P); CIR_CHECK_PTR(Pd);
    if (rows <= 0 || cols <= 0 || p_drop < 0.f || p_drop >= 1.f || (mask && rows_per_mask <= 0)) return CIR_EINVAL;
    if (dtype != CIR_BF16 && dtype != CIR_F16) return CIR_EDTYPE;
    dim3 grid((unsigned)((rows + 3) / 4)), block(256);
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (dtype == CIR_BF16) hipLaunchKernelGGL((softmax_fwd_kernel<__bf16>), grid, block, 0, s, S, ld_s, mask, rows_per_mask, ld_mask, reinterpret_cast<__bf16*>(P), reinterpret_cast<__bf16*>(Pd), ld_p, rows, cols, scale, p_drop, seed);
    else hipLaunchKernelGGL((softmax_fwd_kernel<_Float16>), grid, block, 0, s, S, ld_s, mask, rows_per_mask, ld_mask, reinterpret_cast<_Float16*>(P), reinterpret_cast<_Float16*>(Pd), ld_p, rows, cols, scale, p_drop, seed);
    CIR_LAUNCH_RESULT();
}

extern "C" int cir_softmax_dropout_bwd(const void* P, int64_t ld_p, const float* dPd, int64_t ld_d, void* dS, int64_t ld_ds, int64_t rows, int cols,
                                       float scale, float p_drop, uint64_t seed, int dtype, void* stream) {
    CIR_CHECK_PTR(P); CIR_CHECK_PTR(dPd); CIR_CHECK_PTR(dS);
    if (rows <= 0 || cols <= 0 || p_drop < 0.f || p_drop >= 1.f) return CIR_EINVAL;
    if (dtype != CIR_BF16 && dtype != CIR_F16) return CIR_EDTYPE;
    dim3 grid((unsigned)((rows + 3) / 4)), block(256);
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (dtype == CIR_BF16) hipLaunchKernelGGL((softmax_bwd_kernel<__bf16>), grid, block, 0, s, reinterpret_cast<const __bf16*>(P), ld_p, dPd, ld_d, reinterpret_cast<__bf16*>(dS), ld_ds, rows, cols, scale, p_drop, seed);
    else hipLaunchKernelGGL((softmax_bwd_kernel<_Float16>), grid, block, 0, s, reinterpret_cast<const _Float16*>(P), ld_p, dPd, ld_d, reinterpret_cast<_Float16*>(dS), ld_ds, rows, cols, scale, p_drop, seed);
    CIR_LAUNCH_RESULT();
}

extern "C" int cir_layernorm_bwd(const float* x, const float* gamma, const float* dy, float* dx, float* dgamma, float* dbeta, int64_t rows, int cols,
                                 float eps, void* stream) {
    CIR_CHECK_PTR(x); CIR_CHECK_PTR(gamma); CIR_CHECK_PTR(dy); CIR_CHECK_PTR(dx); CIR_CHECK_PTR(dgamma); CIR_CHECK_PTR(dbeta);
    if (rows <= 0 || cols <= 0) return CIR_EINVAL;
    if (cols > 64 * kLnCols) return CIR_ESHAPE;
    dim3 grid((unsigned)((rows + 31) / 32)), block(256);
    hipLaunchKernelGGL(layernorm_bwd_kernel, grid, block, 0, reinterpret_cast<hipStream_t>(stream), x, gamma, dy, dx, dgamma, dbeta, rows, cols, eps);
    CIR_LAUNCH_RESULT();
}

extern "C" int cir_eltwise(const void* z, int z_dtype, const float* dy, void* out, int out_dtype, int64_t n, int mode, float p_drop, uint64_t seed,
                           void* stream) {
    CIR_CHECK_PTR(z); CIR_CHECK_PTR(out);
    if (n <= 0 || mode < 0 || mode > 6 || (mode != 6 && (p_drop < 0.f || p_drop >= 1.f))) return CIR_EINVAL;
    if ((mode == 1 || mode == 3 || mode == 5) && dy == nullptr) return CIR_EINVAL;
    dim3 grid((unsigned)((n + 1023) / 1024)), block(256);
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const int vec = (reinterpret_cast<uintptr_t>(z) % 16 == 0 && reinterpret_cast<uintptr_t>(out) % 16 == 0 && reinterpret_cast<uintptr_t>(dy) % 16 == 0) ? 1 : 0;
#define CIR_ELT(TZ, TO) hipLaunchKernelGGL((eltwise_kernel<TZ, TO>), grid, block, 0, s, reinterpret_cast<const TZ*>(z), dy, reinterpret_cast<TO*>(out), n, mode, p_drop, seed, vec)
#define CIR_ELT_OUT(TZ) do { if (out_dtype == CIR_F32) CIR_ELT(TZ, float); else if (out_dtype == CIR_BF16) CIR_ELT(TZ, __bf16); else if (out_dtype == CIR_F16) CIR_ELT(TZ, _Float16); else return CIR_EDTYPE; } while (0)
    if (z_dtype == CIR_F32) CIR_ELT_OUT(float);
    else if (z_dtype == CIR_BF16) CIR_ELT_OUT(__bf16);
    else if (z_dtype == CIR_F16) CIR_ELT_OUT(_Float16);
    else return CIR_EDTYPE;
#undef CIR_ELT_OUT
#undef CIR_ELT
    CIR_LAUNCH_RESULT();
}

extern "C" int cir_colsum(const float* x, int64_t ld, float* out, int64_t rows, int cols, void* stream) {
    CIR_CHECK_PTR(x); CIR_CHECK_PTR(out);
    if (rows <= 0 || cols <= 0) return CIR_EINVAL;
    const int64_t rpb = 32;
    dim3 grid((cols + 255) / 256, (unsigned)((rows + rpb - 1) / rpb)), block(256);
    hipLaunchKernelGGL(colsum_kernel, grid, block, 0, reinterpret_cast<hipStream_t>(stream), x, ld, out, rows, cols, rpb);
    CIR_LAUNCH_RESULT();
}

extern "C" int cir_embed_bwd(const int64_t* ids, const float* dy, float* dword, float* dpos, int64_t rows, int L, int cols, void* stream) {
    CIR_CHECK_PTR(ids); CIR_CHECK_PTR(dy); CIR_CHECK_PTR(dword); CIR_CHECK_PTR(dpos);
    if (rows <= 0 || L <= 0 || cols <= 0) return CIR_EINVAL;
    hipLaunchKernelGGL(embed_bwd_kernel, dim3((unsigned)rows), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), ids, dy, dword, dpos, rows, L, cols);
    CIR_LAUNCH_RESULT();
}

extern "C" int cir_adamw_step(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2, float eps, float weight_decay,
                              int step, void* stream) {
    CIR_CHECK_PTR(p); CIR_CHECK_PTR(g); CIR_CHECK_PTR(m); CIR_CHECK_PTR(v);
    if (n <= 0 || step <= 0) return CIR_EINVAL;
    const float bc1 = 1.f - powf(beta1, (float)step), bc2 = 1.f - powf(beta2, (float)step);
    hipLaunchKernelGGL(adamw_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), p, g, m, v, n, lr, beta1,
                       beta2, eps, weight_decay, bc1, bc2);
    CIR_LAUNCH_RESULT();
}
