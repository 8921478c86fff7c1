// cir_cross_attention_folded: the cross-attention of the two-branch BertLayer with the KEY and VALUE projections folded out of the
// image-token side (round 5; nlvr_encoder.py:150-168, 183-217 with encoder_hidden_states = the candidate's image tokens X):
//     S_h   = q_h K_h^T / 8          K_h = X W_k,h^T + b_k,h      ->  S_h = (q_h W_k,h) X^T / 8 + const_row      (const drops out of the softmax)
//     ctx_h = P_h V_h                V_h = X W_v,h^T + b_v,h      ->  ctx_h = (P_h X) W_v,h^T + b_v,h             (rows of P_h sum to 1)
// With L = 32 caption tokens against N = 197 image tokens the projection is 6x cheaper on the query side: per (candidate, layer, both
// branches) 614 MFLOP instead of 969 (K|V GEMM 930 + attention 39) - and neither the (T N, 4 D) K|V tensor (8.1 GB per layer at 6720
// candidates) nor its re-read by the attention kernel exists.  cir_cls_cross_attention does the same for the CLS rows of the last layer.
//
// One workgroup (8 waves) per (candidate, branch).  The branch's 12 heads x 32 tokens are 384 stacked query rows; wave w owns rows
// of 1.5 heads = three 16-row blocks (a wave pair shares the middle head).  Everything is computed TRANSPOSED on v_mfma_f32_16x16x32 so that every product's
// accumulators are the B operand of the next one without a trip through LDS (a lane of a 16x16 accumulator holds rows 4g .. 4g+3 of one
// column; two such tiles give the 8 k-slots of a B operand - the consumer's A operand is read with the same k-slot map):
//   phase 1, per 64-feature chunk c of the 768 (X chunk = 224 keys x 64 features staged in LDS by LDS-DMA, two buffers):
//     G1  Q'^T[f][row]   = sum_d  WkT[f][h(row) 64 + d] q[row][h 64 + d]            (A: W_k^T from L2, 16 B per lane; B: q fragments)
//     G2  S^T[key][row] += sum_f  X[key][f] Q'^T[f][row]                            (A: one 16-byte LDS read per fragment; 168 accumulator registers)
//   softmax over the keys of each row: in-register over 56 values + two cross-lane steps (lane ^ 16, lane ^ 32); P^T packed to 16 bit
//   phase 2, per 64-feature chunk c (X chunk staged again, rows 160 B apart for the transposing reads):
//     G3  C'^T[f][row]   = sum_key X[key][f] P^T[key][row]                          (A: ds_read_b64_tr_b16 of the same row-major tile)
//     G4  ctx^T[d][row] += sum_f  Wv[h 64 + d][f] C'^T[f][row]                      (A: W_v with its columns pre-permuted to the k-slot map)
//   epilogue: ctx / rowsum + b_v -> out[candidate][token][branch][h 64 + d].
// Bound: MFMA.  20 736 MFMAs (16x16x32) per workgroup = 83 k cycles per CU at one MFMA per 16 cycles and SIMD; operand traffic from L2
// ~3.9 MB per workgroup (W_k^T, W_v: 1.33 x 1.18 MB each - neighbouring waves share a head -, X twice).

#include "xattn_fold.hpp"

#ifndef FOLD_DBG
#define FOLD_DBG 0      // diagnostic builds only (make folddbg; timing, wrong results): 1 no weight re-loads, 2 no LDS fragment re-reads, 4 no DMA / barriers after the first chunk
#endif

namespace cir {

#if FOLD_DBG & 8
__device__ unsigned long long g_fold_stamps[8 * 64];        // [workgroup < 4][wave 0 / 4][stamp < 64]
#define FOLD_STAMP(K)                                                                                          \
    do { if (blockIdx.x < 4 && (wave & 3) == 0 && lane == 0 && (K) < 64) {                                      \
             unsigned long long t_; __builtin_amdgcn_sched_barrier(0);                                         \
             asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory");                       \
             __builtin_amdgcn_sched_barrier(0);                                                                 \
             g_fold_stamps[(blockIdx.x * 2 + (wave >> 2)) * 64 + (K)] = t_; } } while (0)
#else
#define FOLD_STAMP(K) do {} while (0)
#endif

// MASKED (round 6): an additive key mask per candidate (padded candidate token sets: (1 - attention_mask) * finfo.min, nlvr_encoder.py:863-868)
// joins the logits in the softmax; the unmasked instantiation is unchanged.
template <typename T, bool MASKED = false>
__global__ __launch_bounds__(512, 2) void xattn_fold_kernel(const FoldArgs a) {
    using X8 = typename Elem<T>::x8;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l16 = lane & 15, g = lane >> 4;
    // XCDs 0-3 run branch 0, XCDs 4-7 branch 1 (workgroup i lands on XCD i % 8): an XCD's 4-MB L2 then holds ONE branch's W_k^T / W_v
    // (1.18 MB each) beside the X stream, instead of both branches' 4.7 MB; the two workgroups of a candidate run at about the same
    // time, so the second reader of its X finds the lines in the Infinity Cache
    const int b = (blockIdx.x >> 2) & 1;
    const int t = (blockIdx.x >> 3) * 4 + (blockIdx.x & 3);
    if (t >= a.T) return;                                      // (whole workgroup; the grid is rounded up to a multiple of 8)

    const T* X = reinterpret_cast<const T*>(a.x) + (int64_t)t * a.x_s1;
    const T* Wk = reinterpret_cast<const T*>(a.wkt) + (int64_t)b * a.w_sb;
    const T* Wv = reinterpret_cast<const T*>(a.wvp) + (int64_t)b * a.w_sb;

    // This wave's three 16-row blocks of the 384 stacked (head, token) rows.  A wave pair (2 k, 2 k + 1) covers heads 3 k .. 3 k + 2: the even
    // wave takes head 3 k whole and the first half of 3 k + 1, the odd wave head 3 k + 2 whole and the second half of 3 k + 1 - so EVERY wave
    // has blocks 0, 1 = the two halves of one head (hA) and block 2 = one half of another (hC), and no per-wave operand select is needed
    // (a v_cndmask per fragment register costs vector-issue cycles the 16-cycle MFMA gaps do not have).
    const int pair = wave >> 1, odd = wave & 1;
    const int hA = 3 * pair + 2 * odd, hC = 3 * pair + 1;
    int hq[3], tok[3];
    hq[0] = hA; hq[1] = hA; hq[2] = hC;
    tok[0] = l16; tok[1] = 16 + l16; tok[2] = 16 * odd + l16;
    // q of this (candidate, branch) in LDS: 32 token rows x 768, rows 1552 B apart (b128 fragment reads of 16 rows: distinct banks),
    // token rows beyond L zero.  G1's B operands are read from here per unit instead of living in 24 registers.
    char* const qs = smem + 2 * kFoldBuf;
    {
        const T* qb_ = reinterpret_cast<const T*>(a.q) + (int64_t)b * a.q_sb + (int64_t)t * a.L * a.q_rs;
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            const int c = tid + i * 512;                      // 3072 16-byte pieces
            const int row = c / 96, ch = c - row * 96;
            X8 v;
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = static_cast<T>(0.f);
            if (row < a.L) v = *reinterpret_cast<const X8*>(qb_ + (int64_t)row * a.q_rs + ch * 8);
            *reinterpret_cast<X8*>(qs + row * kStrideQ + ch * 16) = v;
        }
    }
    // this lane's q fragment addresses: block qb, k-step ks -> row (16 qb' + l16) of the 32, columns h 64 + 32 ks + 8 g
    int qoff[3];
#pragma unroll
    for (int qb = 0; qb < 3; ++qb) qoff[qb] = tok[qb] * kStrideQ + (hq[qb] * 64 + 8 * g) * 2;

    // X-chunk staging by LDS-DMA: the buffer is a linear array of 16-byte slots (lane-linear per wave-instruction); slot s holds 16-byte
    // piece c = s % SPR of row s / SPR (pieces >= 8 are padding, rows >= N repeat row N - 1: finite values that meet probability 0)
    // phase 1: 128-byte rows, LDS slot s of row r holds the row's 16-byte piece s ^ (r & 7) (the swizzle sits on the DMA's SOURCE
    // address; conflict-free ds_read_b128 for the 16x16x32 operand pattern, as in gemm.hip)
    int xoff1[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int s = (wave * 4 + j) * 64 + lane;
        const int row = s >> 3, c = (s & 7) ^ (row & 7);
        xoff1[j] = (min(row, a.N - 1) * kFoldD + c * 8) * 2;   // bytes
    }
    const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(X), 0, a.N * kFoldD * 2, 0x00020000);
    auto stage1 = [&](int chunk, int buf) {
        char* base = smem + buf * kFoldBuf + wave * 4 * 1024;
#pragma unroll
        for (int j = 0; j < 4; ++j) FOLD_DMA(rs_x, xoff1[j], chunk * 128, base + j * 1024);
    };

    // ---------------------------------------------------------------- phase 1: S^T (224 keys x 48 rows per wave) ------------------
    // 24 units (chunk kc, 32-feature half): G1 (12 MFMAs) then G2 (42 MFMAs).  The W_k^T fragments of unit u + 1 are requested when
    // unit u's G2 starts (their L2 latency sits under its 42 MFMAs), the X fragments two key blocks ahead of their MFMAs.
    f32x4 S[kFoldKB][3];
#pragma unroll
    for (int kb = 0; kb < kFoldKB; ++kb)
#pragma unroll
        for (int qb = 0; qb < 3; ++qb) S[kb][qb] = f32x4{0.f, 0.f, 0.f, 0.f};

    X8 w[2][2][2];                                              // [16-feature block][k-step][head hA / hC]
    const int wlane = lane * 16;                               // weights are stored in FRAGMENT order: one 1-KiB block per wave-load, lane-linear
    // G1's two tiles of a 32-feature unit INTERLEAVE the features: tile fbh row 4 g' + r = feature 8 g' + 4 fbh + r.  A lane of the two
    // accumulators (rows 4 g .. 4 g + 3 of each) then holds features 8 g .. 8 g + 7 - the natural k-slot order - and G2's A operand is ONE
    // 16-byte LDS read of 8 consecutive features (ds_read2_b64 of two 8-byte pieces runs at half the LDS rate).
    // Both weight tensors arrive PRE-PACKED per MFMA fragment (ops.fold_pack_key / fold_pack_value): block ((unit 2 + fbh) 12 + head) 2 + ks
    // of W_k^T resp. (unit 4 + db) 12 + head of W_v is the 1 KiB a wave-load needs, lane-linear - 8 whole cache lines per request where the
    // row-major tensors gave 16 half-lines (the CU's vector-memory path, not the MFMA pipe, was setting the pace).
    const __amdgpu_buffer_rsrc_t rs_k = __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(Wk), 0, kFoldD * kFoldD * 2, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_v = __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(Wv), 0, kFoldD * kFoldD * 2, 0x00020000);
    auto load_wk = [&](int unit) {                              // unit = 2 kc + half: features [32 unit, 32 unit + 32)
#pragma unroll
        for (int fbh = 0; fbh < 2; ++fbh) {
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                w[fbh][ks][0] = wload<X8>(rs_k, wlane, (((unit * 2 + fbh) * 12 + hA) * 2 + ks) * 1024);
                w[fbh][ks][1] = wload<X8>(rs_k, wlane, (((unit * 2 + fbh) * 12 + hC) * 2 + ks) * 1024);
            }
        }
    };
    FOLD_STAMP(0);
    stage1(0, 0);
    load_wk(0);
    for (int kc = 0; kc < kFoldChunks; ++kc) {
        if (kc < 4) FOLD_STAMP(1 + 6 * kc);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // this chunk's DMA pieces have landed (explicit: see gemm.hip)
        if (kc < 4) FOLD_STAMP(2 + 6 * kc);
        if (!(FOLD_DBG & 4) || kc == 0) __syncthreads();
        if (kc < 4) FOLD_STAMP(3 + 6 * kc);
        const char* xs = smem + (kc & 1) * kFoldBuf;
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            // G1: Q'^T tiles of features [64 kc + 32 half, + 32) for the three row blocks
            f32x4 a1[2][3];
#pragma unroll
            for (int fbh = 0; fbh < 2; ++fbh)
#pragma unroll
                for (int qb = 0; qb < 3; ++qb) a1[fbh][qb] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                X8 qf[3];
#pragma unroll
                for (int qb = 0; qb < 3; ++qb) qf[qb] = *reinterpret_cast<const X8*>(qs + qoff[qb] + 64 * ks);
#pragma unroll
                for (int fbh = 0; fbh < 2; ++fbh) {
                    a1[fbh][0] = Elem<T>::mfma16(w[fbh][ks][0], qf[0], a1[fbh][0]);
                    a1[fbh][1] = Elem<T>::mfma16(w[fbh][ks][0], qf[1], a1[fbh][1]);
                    a1[fbh][2] = Elem<T>::mfma16(w[fbh][ks][1], qf[2], a1[fbh][2]);
                }
            }
            X8 bq[3];
#pragma unroll
            for (int qb = 0; qb < 3; ++qb) bq[qb] = pack_acc2<T>(a1[0][qb], a1[1][qb]);
            if (kc < 4 && half == 0) FOLD_STAMP(4 + 6 * kc);
            __builtin_amdgcn_sched_barrier(0);
            // This unit's vector-memory requests - the 8 W_k^T fragments of the next unit, then (first half only) the 4 LDS-DMA pieces of the
            // next chunk - are issued ONE PER KEY BLOCK inside G2, not as a burst in front of it: the CU's vector-memory path takes a 1-KiB
            // request every ~16 cycles, 8 waves x 12 requests at once queue for ~1.5 k cycles, and a wave stuck at its request cannot issue
            // the MFMAs behind it (stamps: the SIMD's second wave spent 6.9 k cycles in a 0.7 k-cycle G2).  The DMA pieces go BEHIND the
            // weight requests (vector memory retires in order: the wait for the weights at the next unit's G1 does not cover them); their
            // buffer was last read in chunk kc - 1, before this chunk's barrier.
            const int nu = 2 * kc + half + 1;                       // next unit
            const bool more_w = nu < 2 * kFoldChunks && !(FOLD_DBG & 1), more_x = half == 0 && kc + 1 < kFoldChunks && !(FOLD_DBG & 4);
            auto mem_op = [&](int j) {
                if (j < 8) {
                    if (more_w) {
                        const int fbh = j >> 2, ks = (j >> 1) & 1, hc = j & 1;
                        w[fbh][ks][hc] = wload<X8>(rs_k, wlane, (((nu * 2 + fbh) * 12 + (hc ? hC : hA)) * 2 + ks) * 1024);
                    }
                } else if (j < 12) {
                    if (more_x) FOLD_DMA(rs_x, xoff1[j - 8], (kc + 1) * 128, smem + ((kc + 1) & 1) * kFoldBuf + (wave * 4 + j - 8) * 1024);
                }
            };
            // G2: k-slot (g, j) of this 32-feature step = feature 32 half + 8 g + j: piece 4 half + g of the key's row
            const char* xr = xs + l16 * 128 + (((4 * half + g) ^ (l16 & 7)) << 4);
            auto rd = [&](int kb) { return *reinterpret_cast<const X8*>(xr + kb * 16 * 128); };
            X8 xa[3];
            xa[0] = rd(0);
            xa[1] = rd(1);
#pragma unroll
            for (int kb = 0; kb < kFoldKB; ++kb) {
                if (kb + 2 < kFoldKB && !(FOLD_DBG & 2)) xa[(kb + 2) % 3] = rd(kb + 2);
                mem_op(kb);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int qb = 0; qb < 3; ++qb) S[kb][qb] = Elem<T>::mfma16(xa[kb % 3], bq[qb], S[kb][qb]);
                __builtin_amdgcn_sched_barrier(0);
            }
            if (kc < 4) FOLD_STAMP(5 + 6 * kc + half);
        }
    }
    FOLD_STAMP(25);

    // ---------------------------------------------------------------- softmax over the keys of each row (log2 domain) ---------------
    const float sl = a.scale * 1.4426950408889634f;
    float rinv[3];
    X8 P[7][3];
#pragma unroll
    for (int qb = 0; qb < 3; ++qb) {
        float m = -INFINITY;
#pragma unroll
        for (int kb = 0; kb < kFoldKB; ++kb)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int key = 16 * kb + 4 * g + r;
                float v = key < a.N ? S[kb][qb][r] : -INFINITY;
                if constexpr (MASKED) {   // log2-domain logit incl. the mask (finfo.min-style masks stay finite: all-masked rows uniform, like the reference)
                    const float mk = a.mask[(int64_t)t * a.m_st + min(key, a.N - 1)];
                    v = key < a.N ? fmaf(fmaxf(mk, -2.0e38f), 1.4426950408889634f, S[kb][qb][r] * sl) : -INFINITY;
                }
                S[kb][qb][r] = v;
                m = fmaxf(m, v);
            }
        m = fmaxf(m, __shfl_xor(m, 16, 64));
        m = fmaxf(m, __shfl_xor(m, 32, 64));
        const float ms = MASKED ? m : m * sl;
        float sum = 0.f;
#pragma unroll
        for (int kb = 0; kb < kFoldKB; ++kb)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float p = MASKED ? __builtin_amdgcn_exp2f(S[kb][qb][r] - ms) : __builtin_amdgcn_exp2f(fmaf(S[kb][qb][r], sl, -ms));
                S[kb][qb][r] = p;
                sum += p;
            }
        sum += __shfl_xor(sum, 16, 64);
        sum += __shfl_xor(sum, 32, 64);
        rinv[qb] = 1.0f / sum;
#pragma unroll
        for (int p = 0; p < 7; ++p) P[p][qb] = pack_acc2<T>(S[2 * p][qb], S[2 * p + 1][qb]);
    }

    // ---------------------------------------------------------------- phase 2: ctx^T (64 d x 48 rows per wave) ----------------------
    // 24 units (chunk nc, 32-feature half): G3 on the half's two 16-feature blocks (42 MFMAs), then G4 (12 MFMAs); W_v fragments of
    // unit u + 1 requested when unit u's G3 starts.
    f32x4 c4[4][3];
#pragma unroll
    for (int db = 0; db < 4; ++db)
#pragma unroll
        for (int qb = 0; qb < 3; ++qb) c4[db][qb] = f32x4{0.f, 0.f, 0.f, 0.f};
    X8 wv[4][2];                                                // [16-d block][head hA / hC]
    int xoff2[5];
#pragma unroll
    for (int j = 0; j < 5; ++j) {
        const int s = (wave * 5 + j) * 64 + lane;
        const int row = min(s / 10, a.N - 1), c = min(s % 10, 7);
        xoff2[j] = (row * kFoldD + c * 8) * 2;
    }
    auto stage2 = [&](int chunk, int buf) {
        char* base = smem + buf * kFoldBuf + wave * 5 * 1024;
#pragma unroll
        for (int j = 0; j < 5; ++j) FOLD_DMA(rs_x, xoff2[j], chunk * 128, base + j * 1024);
    };
    FOLD_STAMP(26);
    __syncthreads();                                            // every wave is done with phase 1's buffers
    stage2(0, 0);
    const int troff = (4 * g + (l16 >> 2)) * kStride2 + (4 * (l16 & 3)) * 2;
    for (int nc = 0; nc < kFoldChunks; ++nc) {
        if (nc < 4) FOLD_STAMP(27 + 6 * nc);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (nc < 4) FOLD_STAMP(28 + 6 * nc);
        if (!(FOLD_DBG & 4) || nc == 0) __syncthreads();
        if (nc < 4) FOLD_STAMP(29 + 6 * nc);
        const char* xs = smem + (nc & 1) * kFoldBuf + troff;
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            // G3: C'^T tiles of features [64 nc + 32 half, + 32): 7 key pairs x 2 feature blocks
            f32x4 a3[2][3];
#pragma unroll
            for (int fbh = 0; fbh < 2; ++fbh)
#pragma unroll
                for (int qb = 0; qb < 3; ++qb) a3[fbh][qb] = f32x4{0.f, 0.f, 0.f, 0.f};
            // The transposing reads are inline asm with counted waits of their own: behind the BUILTIN hipcc puts an s_waitcnt vmcnt(0) -
            // it cannot tell the read from the LDS-DMA pieces just requested - i.e. the whole HBM latency of the next chunk, every chunk.
            // (Phase 2 has no other LDS reads; the waits name the registers they release - cdna_hip_programming.md, form (ii).)
            // vector-memory requests of this unit, one per G3 step (see phase 1): the 8 W_v fragments G4 needs at the END of this unit,
            // then (first half only) the 5 LDS-DMA pieces of the next chunk
            const int cu = 2 * nc + half;
            const bool more_x = half == 0 && nc + 1 < kFoldChunks && !(FOLD_DBG & 4);
            auto mem_op = [&](int j) {
                if (j < 8) {
                    if (!(FOLD_DBG & 1) || cu == 0) {
                        const int db = j >> 1, hc = j & 1;
                        wv[db][hc] = wload<X8>(rs_v, wlane, ((cu * 4 + db) * 12 + (hc ? hC : hA)) * 1024);
                    }
                } else if (j < 13) {
                    if (more_x) FOLD_DMA(rs_x, xoff2[j - 8], (nc + 1) * 128, smem + ((nc + 1) & 1) * kFoldBuf + (wave * 5 + j - 8) * 1024);
                }
            };
            const unsigned xaddr = (unsigned)(size_t)(lptr_t)(xs) + (32 * half) * 2;
            u32x2 lo[3], hi[3];
#define FOLD_TR(SLOT, I)                                                                                                        \
            asm volatile("ds_read_b64_tr_b16 %0, %2 offset:%3\n\tds_read_b64_tr_b16 %1, %2 offset:%4"                             \
                         : "=&v"(lo[SLOT]), "=&v"(hi[SLOT]) : "v"(xaddr), "n"((32 * ((I) >> 1)) * kStride2 + (16 * ((I) & 1)) * 2),  \
                           "n"((32 * ((I) >> 1) + 16) * kStride2 + (16 * ((I) & 1)) * 2) : "memory")
            FOLD_TR(0, 0);
            FOLD_TR(1, 1);
#pragma unroll
            for (int i = 0; i < 14; ++i) {
                if (i + 2 < 14 && !(FOLD_DBG & 2)) {
                    switch (i + 2) {      // (immediates: the unrolled index must reach the asm as a constant)
                        case 2: FOLD_TR(2, 2); break; case 3: FOLD_TR(0, 3); break; case 4: FOLD_TR(1, 4); break; case 5: FOLD_TR(2, 5); break;
                        case 6: FOLD_TR(0, 6); break; case 7: FOLD_TR(1, 7); break; case 8: FOLD_TR(2, 8); break; case 9: FOLD_TR(0, 9); break;
                        case 10: FOLD_TR(1, 10); break; case 11: FOLD_TR(2, 11); break; case 12: FOLD_TR(0, 12); break; default: FOLD_TR(1, 13); break;
                    }
                }
                mem_op(i);
                if (FOLD_DBG & 2) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(lo[i % 3]), "+v"(hi[i % 3]));
                else if (i + 2 < 14) asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(lo[i % 3]), "+v"(hi[i % 3]));
                else if (i + 1 < 14) asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(lo[i % 3]), "+v"(hi[i % 3]));
                else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(lo[i % 3]), "+v"(hi[i % 3]));
                __builtin_amdgcn_sched_barrier(0);
                const u32x4 both = {lo[i % 3].x, lo[i % 3].y, hi[i % 3].x, hi[i % 3].y};
                const X8 xa = __builtin_bit_cast(X8, both);
#pragma unroll
                for (int qb = 0; qb < 3; ++qb) a3[i & 1][qb] = Elem<T>::mfma16(xa, P[i >> 1][qb], a3[i & 1][qb]);
                __builtin_amdgcn_sched_barrier(0);
            }
#undef FOLD_TR
            if (nc < 4 && half == 0) FOLD_STAMP(30 + 6 * nc);
            // G4: ctx^T += W_v[:, these 32 features] C'^T (W_v's columns are stored in k-slot order: position 8 g + j of a 32-group =
            // feature (j < 4 ? 4 g + j : 16 + 4 g + j - 4))
            X8 bq[3];
#pragma unroll
            for (int qb = 0; qb < 3; ++qb) bq[qb] = pack_acc2<T>(a3[0][qb], a3[1][qb]);
#pragma unroll
            for (int db = 0; db < 4; ++db) {
                c4[db][0] = Elem<T>::mfma16(wv[db][0], bq[0], c4[db][0]);
                c4[db][1] = Elem<T>::mfma16(wv[db][0], bq[1], c4[db][1]);
                c4[db][2] = Elem<T>::mfma16(wv[db][1], bq[2], c4[db][2]);
            }
            __builtin_amdgcn_sched_barrier(0);
            if (nc < 4) FOLD_STAMP(31 + 6 * nc + half);
        }
    }

    FOLD_STAMP(51);
    // ---------------------------------------------------------------- epilogue -------------------------------------------------------
    T* ob = reinterpret_cast<T*>(a.out) + (int64_t)t * a.o_st + (int64_t)b * a.o_sb;
    const float* bvb = a.bv + b * kFoldD;
#pragma unroll
    for (int qb = 0; qb < 3; ++qb) {
        if (tok[qb] >= a.L) continue;
        T* orow = ob + (int64_t)tok[qb] * a.o_sr + hq[qb] * 64 + 4 * g;
        const float* brow = bvb + hq[qb] * 64 + 4 * g;
#pragma unroll
        for (int db = 0; db < 4; ++db) {
            const float4 b4 = *reinterpret_cast<const float4*>(brow + 16 * db);
            u32x2 o;
            o.x = pack2<T>(fmaf(c4[db][qb][0], rinv[qb], b4.x), fmaf(c4[db][qb][1], rinv[qb], b4.y));
            o.y = pack2<T>(fmaf(c4[db][qb][2], rinv[qb], b4.z), fmaf(c4[db][qb][3], rinv[qb], b4.w));
            *reinterpret_cast<u32x2*>(orow + 16 * db) = o;
        }
    }
}

}  // namespace cir

extern "C" int cir_cross_attention_folded(const void* q, int64_t q_sb, int64_t q_rs, const void* x, int64_t x_s1, const void* wkt, const void* wvp,
                                          int64_t w_sb, const float* bv, const float* key_mask, int64_t mask_stride, void* out, int64_t o_st, int64_t o_sr,
                                          int64_t o_sb, int T, int L, int N, int D, int H, float scale, int dtype, void* stream) {
    using namespace cir;
    CIR_CHECK_PTR(q); CIR_CHECK_PTR(x); CIR_CHECK_PTR(wkt); CIR_CHECK_PTR(wvp); CIR_CHECK_PTR(bv); CIR_CHECK_PTR(out);
    if (T <= 0 || L <= 0 || N <= 0) return CIR_EINVAL;
    if (D != kFoldD || H != 12 || L > 32 || N > 608) return CIR_ESHAPE;
    if (dtype != CIR_BF16 && dtype != CIR_F16) return CIR_EDTYPE;
    if (!cir_aligned16(q) || !cir_aligned16(x) || !cir_aligned16(wkt) || !cir_aligned16(wvp) || !cir_aligned16(bv) || (reinterpret_cast<uintptr_t>(out) & 7) ||
        q_sb % 8 || q_rs % 8 || x_s1 % 8 || w_sb % 8 || o_st % 4 || o_sr % 4 || o_sb % 4)
        return CIR_EALIGN;
    if ((int64_t)T * 2 > 0x7fffffff) return CIR_ESHAPE;
    FoldArgs a;
    a.q = q; a.q_sb = q_sb; a.q_rs = q_rs; a.x = x; a.x_s1 = x_s1; a.wkt = wkt; a.wvp = wvp; a.w_sb = w_sb; a.bv = bv;
    a.out = out; a.o_st = o_st; a.o_sr = o_sr; a.o_sb = o_sb; a.T = T; a.L = L; a.N = N; a.scale = scale;
    a.mask = key_mask; a.m_st = mask_stride;
    if (key_mask && mask_stride < N) return CIR_ESHAPE;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (N > 16 * kFoldKB) return launch_fold16(a, dtype, s);       // 225 .. 608 keys: the 16-rows-per-wave kernel (xattn_fold16.hip)
    const size_t lds = 2 * kFoldBuf + 32 * kStrideQ;
    dim3 grid((unsigned)(8 * ((T + 3) / 4))), block(512);
#define CIR_FOLD_LAUNCH(TT, MK)                                                                                                                   \
    do {                                                                                                                                          \
        const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&xattn_fold_kernel<TT, MK>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
        if (e != hipSuccess) return (int)e;                                                                                                       \
        hipLaunchKernelGGL((xattn_fold_kernel<TT, MK>), grid, block, lds, s, a);                                                                  \
    } while (0)
    if (dtype == CIR_BF16) { if (key_mask) CIR_FOLD_LAUNCH(__bf16, true); else CIR_FOLD_LAUNCH(__bf16, false); }
    else { if (key_mask) CIR_FOLD_LAUNCH(_Float16, true); else CIR_FOLD_LAUNCH(_Float16, false); }
#undef CIR_FOLD_LAUNCH
    CIR_LAUNCH_RESULT();
}

#if FOLD_DBG & 8
extern "C" int cir_debug_fold_stamps(unsigned long long* host) {
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(cir::g_fold_stamps), sizeof(unsigned long long) * 8 * 64);
}
#endif
