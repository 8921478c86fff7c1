// Shared pieces of the folded cross-attention kernels (xattn_fold.hip: N <= 224 keys, 48 query rows per wave; xattn_fold16.hip: N <= 608
// keys, 16 query rows per wave).
#pragma once
#include "common.hpp"
#include "gemm_args.hpp"

namespace cir {

struct FoldArgs {
    const void* q; int64_t q_sb, q_rs;          // element (branch b, row t L + tok, col) at q + b q_sb + row q_rs + col
    const void* x; int64_t x_s1;                // tokens (T, N, 768), rows contiguous
    const void* wkt; const void* wvp; int64_t w_sb;   // W_k^T (2, 768 f, 768 (h, d)); W_v (2, 768 (h, d), 768 f permuted)
    const float* bv;                            // (2, 768)
    void* out; int64_t o_st, o_sr, o_sb;        // element (t, tok, b, col) at out + t o_st + tok o_sr + b o_sb + col
    int T, L, N;
    float scale;
    const float* mask = nullptr; int64_t m_st = 0;   // optional additive key mask (T, N) fp32, rows m_st apart (round 6): logits = S scale + mask
};

typedef __attribute__((address_space(3))) s16x4* lds_s16x4_ptr_f;
constexpr int kFoldD = 768, kFoldChunks = 12, kFoldKB = 14;      // width, 64-feature chunks, 16-key blocks (224 keys)
constexpr int kFoldBuf = 2560 * 16;                              // one X-chunk buffer: 2560 16-byte slots (phase 2: 224 rows x 10 slots + slack)
constexpr int kStride2 = 160;                                    // phase 2's LDS row stride: conflict-free transposing reads
constexpr int kStrideQ = 1568;                                   // q rows in LDS (1536 B + 32: the 16-lane groups of ds_read_b128 hit 64 distinct banks)

// LDS-DMA piece through a buffer descriptor (buffer_load_dwordx4 ... lds): wave-uniform resource + uniform byte offset + the lane's 32-bit
// byte offset.  A load hipcc COUNTS - its vmcnt waits for the weight fragments then leave younger DMA pieces in flight (behind an asm
// piece every compiler wait degenerates to "everything", i.e. to the HBM latency of the chunk just requested)
#define FOLD_DMA(RS, VOFF, SOFF, LDS_DST) __builtin_amdgcn_raw_ptr_buffer_load_lds(RS, (lptr_t)(LDS_DST), 16, VOFF, SOFF, 0, 0)

// 16-byte weight fragment through a buffer descriptor: wave-uniform base (SGPR resource) + uniform byte offset + the lane's 32-bit byte offset
template <typename X8>
__device__ __forceinline__ X8 wload(__amdgpu_buffer_rsrc_t rs, int voff, int soff) {
    return __builtin_bit_cast(X8, __builtin_amdgcn_raw_buffer_load_b128(rs, voff, soff, 0));
}

template <typename T>
__device__ __forceinline__ typename Elem<T>::x8 pack_acc2(const f32x4& lo, const f32x4& hi) {
    typename Elem<T>::x8 r;
#pragma unroll
    for (int j = 0; j < 4; ++j) { r[j] = static_cast<T>(lo[j]); r[4 + j] = static_cast<T>(hi[j]); }
    return r;
}


// N <= 608 keys (the reference's 384-px geometry: 577 tokens): one 16-row block per wave, three workgroups per (candidate, branch)
int launch_fold16(const FoldArgs& a, int dtype, hipStream_t s);

}  // namespace cir
