// cir_attention: softmax(q k^T * scale + mask) v for head_dim 64, one wave per 32 queries.
//
// Bound: MFMA in principle (4*Lq*Lk*64 flop per head), VALU (exp / rescale) and operand delivery in
// practice at these small extents; the path spends < 5 % of its flops here.
//
// Formulation ("keys on rows"): with the 32x32x16 MFMA the score tile is computed TRANSPOSED,
//   S^T[key][query] = K_tile * Q^T,
// so a lane owns one query column and 16 of the tile's 32 keys; the row max / sum of the online
// softmax is an in-register reduction plus ONE cross-lane exchange (lane ^ 32).  P^T then already
// has the register layout of the B operand of the second product
//   O^T[dh][query] += V_tile^T * P^T,
// (accumulator-as-operand, no LDS round trip for P).  V^T fragments come from a row-major LDS copy
// of V through ds_read_b64_tr_b16 (hardware transpose).
//
// Two kernels share that tile update:
//   * attn_shared_kernel - one workgroup per (item, head); the head's whole K and V (Lk <= 608 keys, 152 KiB)
//     are staged ONCE into LDS (bank-swizzled) and every 32-query wave of the workgroup reads them from
//     there: the ViT cases (197 x 197: 7 query tiles, 2 workgroups per CU; 577 x 577: 19 tiles on 10 waves, one
//     workgroup per CU), where every query tile would otherwise re-read K/V from L2.
//   * attn_stream_kernel - one wave per (item, head, 32 queries), K straight from global memory into
//     MFMA fragments, V through a private 4-KiB LDS tile, next tile prefetched into registers while the
//     current one is processed: the text-side cases (32 x 32 self-attention, 32 x 197 / 32 x 577
//     cross-attention) where nothing is shared between waves, and any Lk > 608.
// Ragged extents: rows beyond Lq / Lk are clamped or zero-filled on load; scores of keys >= Lk are -inf.
//
// Measured and dropped in round 3 (ViT 1696 x 12 heads, 197 x 197, staged kernel 640 us):
//   * a persistent STREAMING variant - one wave per query tile, two workgroups per CU walking their heads back to back, the
//     32-key K / V tiles pulled through an 8-slot LDS ring by LDS-DMA (56 KiB per workgroup in flight across head
//     boundaries, one barrier per key tile, next head's Q one head ahead in registers): bit-identical, 628-639 us.  The
//     load -> barrier -> compute -> store structure is therefore NOT what bounds this kernel; a wave's tile update is
//     one serial chain (4 MFMA -> max -> exchange -> 16 exp -> sum -> exchange -> cvt -> 4 MFMA, ~1.4 k cycles of latency for
//     ~520 cycles of issue) and 3.5 waves per SIMD cover only half of it (VALU busy 48 %, matrix pipe 19 %).
//   * v_permlane32_swap for the two half-wave exchanges instead of ds_bpermute: no change in time (and hipcc folds
//     max(r.x, r.y) of the builtin's two results into one of them when both inputs are the same register: wrong values).
//   * two key tiles per update (one running-max / rescale / exchange step per 64 keys, two independent score chains, +24
//     registers = exactly the 128 that keep 4 waves per SIMD, 2 spilled): 689 us - slower.
// Measured and dropped in round 4 (same shape; tools/attn_layout_probe.py, profiles/r4_secondary/attn_layout_probe.txt):
//   * a TWO-PASS kernel without the online softmax: 16 queries per wave on mfma_f32_16x16x32, the scores of all <= 224 keys in 56
//     registers, one row maximum / sum (two cross-lane steps each), 28 + 28 independent MFMAs, P^T fed to the second product straight
//     from the accumulators by defining the B operand's k-slots as [tile 2b keys 4g.. | tile 2b+1 keys 4g..] and reading V^T with the
//     same map through two ds_read_b64_tr_b16 (V rows XOR-swizzled on row bits 1 and 2: conflict-free); 90 registers, no spills,
//     correct to the online kernel's error on the first run.  Alone: 608 us against 679 us (fp16; 603 / 659 bf16); in the benchmark
//     step: 60.1 ms of attention kernels against 60.6 ms - nothing.  With the fused QKV tensor stored HEAD-MAJOR ((36, M, 64): a
//     head's K of one image = 25 KB contiguous instead of 128-byte pieces 4608 bytes apart) both kernels take 598 us.  Three
//     structures (staged online, streamed ring, two-pass) and two layouts all end at 590-680 us = 3.0-3.5 TB/s for the 2.05 GB the
//     kernel must move: neither the softmax chain nor the DRAM access pattern is the bound; a workgroup's life is load 82 KB ->
//     compute -> store with two or three workgroups per CU to overlap, and the time is their sum, not their maximum.

// Measured in round 6 (3392 images x 12 heads, 197 x 197, fp16; the staged kernel 1.20 ms in the step, 639 us per 1696 images alone):
//   * the kernel's memory traffic ALONE in the same workgroup shape (tools/attn_mem_probe.hip: stage K / V, read Q, write the output tile; no
//     MFMA, no softmax): 0.774 ms = 5.3 TB/s (loads only 0.505 ms, stores only 0.185 ms; LDS-DMA staging the same) - the access pattern is
//     not the bound, the missing overlap of a workgroup's load / compute / store is.
//   * a PERSISTENT, double-buffered form - one workgroup per CU, one wave per query tile, the next (image, head)'s K / V by LDS-DMA into the other
//     half of LDS and its Q into registers while the current one computes, the output through a private staging tile (140 KiB of LDS): bit-
//     identical, 696 us per 1696 images; with TWO key tiles per update (one reference / rescale / exchange per 64 keys, two chains of
//     exponentials; 170 registers are free at this occupancy) 655 us.  The loads are hidden by construction there, so the 8 us per unit that
//     remain are the tile updates themselves: 49 updates of ~600 issue cycles (4 + 4 MFMAs of 32 cycles, ~40 vector ops, 16 exponentials at
//     quarter rate, two cross-lane steps) = 29 k SIMD-cycles per unit, 3.5 us even perfectly packed over four SIMDs - and 7 waves per CU pack
//     them at ~45 %.  The staged kernel's two workgroups per CU (14 waves) pack them better and overlap loads by chance: both end at 7.5-8 us.
//     Reaching the 4.8 us the memory path allows needs >= 14 resident waves AND asynchronous loads: two double-buffered workgroups do not fit
//     160 KiB of LDS (2 x (2 x 57 + 28) KiB), and a register prefetch of the next K / V (+44 registers) drops the kernel below 4 waves per SIMD.
//   * lazy rescale (softmax_tile): kept, -2 %.
// Late round 6, the 577 x 577 case (1696 images, fp16; the staged kernel 3355 us = 517 TFLOP/s) - where the time is: one tile update issues
// ~60 plain vector instructions, 16 exponentials, 8 MFMAs and 13 LDS reads = ~500 issue cycles of the wave's SIMD, the matrix pipe is busy for
// 256 of them, and the BUSIEST SIMD of the CU sets the workgroup's time:
//   * 19 query tiles on 10 waves put 6 / 6 / 4 / 3 tiles on the four SIMDs (waves w, w + 4, .. share one); 16 waves: 5 / 5 / 5 / 4 ->
//     3087 us (12 waves 3118, 8 waves 3329).  KEPT: cir_attention picks the wave count by that rule.  Same bits (a tile's arithmetic does not
//     depend on the wave that runs it).
//   * exponent arguments / row sums as plain v_fma_f32 / v_add_f32 instead of v_pk_fma_f32 / v_pk_add_f32 (packed fp32 issues slower than
//     the two operations it replaces beside MFMAs; the file is built with -fno-slp-vectorize): 3015 -> 2915 us, 1046 -> 1019 us per 3392
//     images at 197 tokens.  KEPT, same bits.  One staging batch of 5 chunks per thread instead of 4 + 1: 2905 us.
//   * TWO query tiles per wave (two softmax chains sharing every K / V fragment read, 158 registers, 10 waves, one round): 3307 us against
//     3355 - the chains do not overlap inside a wave (the rescale vote splits the basic block) and the SIMD loads stay 6 / 6 / 4 / 4.  Dropped.
//   * K / V by LDS-DMA in key-tile order with COUNTED vmcnt waits, the tile loop starting on the first two key tiles while the rest is in
//     flight (Q by asm loads so that the compiler's vmcnt(0) does not sit behind the stage): bit-identical, 3020 us against 2915 (197
//     tokens: 1071 against 1019) - slower, as register staging's split load / write already overlaps the neighbours' compute.  Dropped.
//   * row maxima through asm v_max3_f32 (no canonicalising v_max x, x): hipcc does not insert the MFMA-result wait states in front of an
//     asm statement - the maxima were read early (still a valid softmax reference: the error was in the last bit only, and showed as a
//     mismatch between the staged and the streamed kernel).  Dropped; fmaxf stays.
// What is left is the instruction count of the update itself (16 exponentials at 8 cycles, 16 FMAs, 16 adds, 8 conversions, ~12 maxima,
// 6 address adds per 8 MFMAs): 577 tokens now run at 597 TFLOP/s = 0.24 of the MFMA peak, 197 tokens at 0.13-0.16.

#include <type_traits>
#include "attention_args.hpp"

namespace cir {

typedef __attribute__((address_space(3))) s16x4* lds_s16x4_ptr;

// Online-softmax state of one 32-query tile: lane = (query r, key/dh half hh).
struct Softmax {
    float m_run, l_run;
    f32x16 o[2];
    __device__ __forceinline__ void init() {
        m_run = -INFINITY;
        l_run = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) { o[0][i] = 0.f; o[1][i] = 0.f; }
    }
};

// raw scores -> probabilities, running max / sum (log2 domain), rescale of O; returns P^T packed.
// MASKED = false (ViT, and text-side calls without a mask): the scale rides in the exponent's FMA,
// p = exp2(s * sl - m), and the row maximum is taken on the raw scores (sl > 0) - no per-score multiply, no key
// indices.  The ViT kernel is bound by VALU issue (PMC: its ~3.5 waves per SIMD are active 28 % of their cycles each),
// so instructions removed here are time removed.
template <typename T, bool MASKED>
__device__ __forceinline__ void softmax_tile(Softmax& st, f32x16& s, float sl, const float* mp, int key0, int hh, int Lk,
                                             typename Elem<T>::x8 (&pf)[2]) {
    float sv[16];
    if constexpr (MASKED) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int key = key0 + (i & 3) + 8 * (i >> 2) + 4 * hh;
            // clamp keeps finfo.min-style masks finite in the log2 domain (all-masked rows stay uniform, like the reference)
            sv[i] = fmaf(fmaxf(mp[min(key, Lk - 1)], -2.0e38f), kLog2e, s[i] * sl);
        }
    } else {
#pragma unroll
        for (int i = 0; i < 16; ++i) sv[i] = s[i];
    }
    if (key0 + 32 > Lk) {   // wave-uniform: only the last key tile is ragged
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int key = key0 + (i & 3) + 8 * (i >> 2) + 4 * hh;
            sv[i] = key < Lk ? sv[i] : -INFINITY;
        }
    }
    float mx = fmaxf(fmaxf(sv[0], sv[1]), sv[2]);
#pragma unroll
    for (int i = 3; i < 15; i += 2) mx = fmaxf(fmaxf(mx, sv[i]), sv[i + 1]);
    mx = fmaxf(mx, sv[15]);
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    if constexpr (!MASKED) mx *= sl;
    const float m_new = fmaxf(st.m_run, mx);     // finite: every tile holds at least one valid key
    // LAZY rescale (round 6): the running maximum is a REFERENCE point of the exponentials, not their bound - it follows the true row
    // maximum only when some row's maximum has grown by more than 2^kLazyLog2 (then every probability of that row would exceed 64 times
    // the scale the sums were formed at); otherwise the tile is accumulated at the stale reference (p <= 64: exact in the fp32 sums,
    // same relative rounding in the 16-bit P) and the 33 multiplies of the O / l rescale are skipped on almost every tile behind the
    // first.  The result is the same softmax: l and O share the reference.  Measured: 654 -> 639 us per 1696 images (197 x 197).
    constexpr float kLazyLog2 = 6.0f;
    if (__any(mx > st.m_run + kLazyLog2)) {      // (first tile: m_run = -inf)
        const float alpha = __builtin_amdgcn_exp2f(st.m_run - m_new);
        st.l_run *= alpha;
#pragma unroll
        for (int i = 0; i < 16; ++i) { st.o[0][i] *= alpha; st.o[1][i] *= alpha; }
        st.m_run = m_new;
    }
    float psum;
    if constexpr (MASKED) {
        psum = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const float p = __builtin_amdgcn_exp2f(sv[i] - st.m_run);
            sv[i] = p;
            psum += p;
        }
    } else {
        // exponent arguments and the row sum (two chains: even / odd scores) as SINGLE fp32 operations: beside MFMAs a v_pk_fma_f32 /
        // v_pk_add_f32 costs more issue time than the two plain operations it replaces (round 6: 1046 -> 1019 us per 3392 images at 197
        // tokens, 3015 -> 2915 us per 1696 at 577; same bits) - the file is built with -fno-slp-vectorize so that hipcc does not re-pack them
        float a0, a1;
        const float nm = -st.m_run;
#pragma unroll
        for (int i = 0; i < 16; i += 2) {
            const float p0 = __builtin_amdgcn_exp2f(__builtin_fmaf(sv[i], sl, nm));
            const float p1 = __builtin_amdgcn_exp2f(__builtin_fmaf(sv[i + 1], sl, nm));
            sv[i] = p0;
            sv[i + 1] = p1;
            a0 = i ? a0 + p0 : p0;
            a1 = i ? a1 + p1 : p1;
        }
        psum = a0 + a1;
    }
    psum += __shfl_xor(psum, 32, 64);
    st.l_run += psum;
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
        for (int j = 0; j < 8; ++j) pf[s2][j] = static_cast<T>(sv[8 * s2 + j]);
}

// O^T += V_tile^T * P^T with V row-major in LDS at `vt` (128-byte rows, 64-byte halves swapped on rows with
// bit 1 set so that the 4-row transposed reads of a 32-lane half hit 64 distinct banks).  `voff[dt]` = this lane's
// byte offset inside a tile for dh-tile dt (tile-invariant: computed once per wave, see pv_lane_offsets); the two
// key halves and the two 8-row blocks of a read pair are immediate offsets.
template <typename T>
__device__ __forceinline__ void pv_tile(Softmax& st, const char* vt, const int (&voff)[2], const typename Elem<T>::x8 (&pf)[2]) {
    using X8 = typename Elem<T>::x8;
#pragma unroll
    for (int dt = 0; dt < 2; ++dt) {
        const char* base0 = vt + voff[dt];
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
            const char* base = base0 + (16 * s2) * 128;
            const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(base));
            const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(base + 8 * 128));
            s16x8 both;
            both.s0 = lo.x; both.s1 = lo.y; both.s2 = lo.z; both.s3 = lo.w;
            both.s4 = hi.x; both.s5 = hi.y; both.s6 = hi.z; both.s7 = hi.w;
            st.o[dt] = Elem<T>::mfma32(__builtin_bit_cast(X8, both), pf[s2], st.o[dt]);
        }
    }
}
__device__ __forceinline__ void pv_lane_offsets(int tr_lane_off, int (&voff)[2]) {
#pragma unroll
    for (int dt = 0; dt < 2; ++dt) voff[dt] = ((tr_lane_off & 0x7f) ^ (dt * 64)) + (tr_lane_off & ~0x7f);
}

// this lane's address inside a 4-key x 16-dh transposed-read block (without the dh-tile term):
// row (4*hh + q), byte column (16*(G&1) + 4p)*2, halves swapped when the row's bit 1 is set (row = 4*hh' + q, q>>1)
__device__ __forceinline__ int tr_lane_offset(int lane) {
    const int i16 = lane & 15, hh = lane >> 5;
    const int q = i16 >> 2, p = i16 & 3;
    const int col = ((16 * ((lane >> 4) & 1) + 4 * p) * 2) ^ ((q >> 1) << 6);
    return (4 * hh + q) * 128 + col;
}

template <typename T>
__device__ __forceinline__ void store_out(const Softmax& st, T* op /* row base + h*64 + 4*hh */) {
    const float inv = 1.0f / st.l_run;
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int qd = 0; qd < 4; ++qd) {
            u32x2 p;
            p.x = pack2<T>(st.o[dt][qd * 4 + 0] * inv, st.o[dt][qd * 4 + 1] * inv);
            p.y = pack2<T>(st.o[dt][qd * 4 + 2] * inv, st.o[dt][qd * 4 + 3] * inv);
            *reinterpret_cast<u32x2*>(op + dt * 32 + 8 * qd) = p;
        }
}

// ---------------------------------------------------------------------------------------------------------------------
template <typename T, bool MASKED>
__global__ __launch_bounds__(1024) void attn_shared_kernel(const AttnArgs a, int lk_pad) {
    using X8 = typename Elem<T>::x8;
    extern __shared__ __attribute__((aligned(16))) char dyn[];
    char* Ks = dyn;
    char* Vs = dyn + lk_pad * 128;

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nwaves = blockDim.x >> 6;
    int64_t t = blockIdx.x;
    const int h = (int)(t % a.H);
    t /= a.H;
    const int b0 = (int)(t % a.B0);
    const int64_t b1 = t / a.B0;
    const int64_t kb1 = a.kv_index ? a.kv_index[b1] : b1;
    const T* kb = reinterpret_cast<const T*>(a.k) + kb1 * a.k_s1 + b0 * a.k_s0 + h * 64;
    const T* vb = reinterpret_cast<const T*>(a.v) + kb1 * a.v_s1 + b0 * a.v_s0 + h * 64;
    const float* mp = MASKED ? a.mask + b1 * a.m_s1 + b0 * a.m_s0 : nullptr;

    const int r = lane & 31, hh = lane >> 5;
    // Q fragments of this wave's first query tile: requested BEFORE the K/V staging so that their latency overlaps it
    const T* const qbase = reinterpret_cast<const T*>(a.q) + b1 * a.q_s1 + b0 * a.q_s0 + h * 64 + 8 * hh;
    X8 qf[4];
    {
        const int qrow = min(min(wave, a.nqt - 1) * 32 + r, a.Lq - 1);
#pragma unroll
        for (int s = 0; s < 4; ++s) qf[s] = *reinterpret_cast<const X8*>(qbase + (int64_t)qrow * a.q_rs + 16 * s);
    }
    // ---- stage K (chunk ^ ((row>>1)&7): conflict-free 32x32x16 A-operand reads) and V (halves swapped on bit 1) ----
    // Batches of SB chunks per thread: all 2*SB global loads of a batch are in flight before the first LDS write (a
    // rolled loop exposes one HBM latency per chunk; the whole stage is ~4 chunks per thread at 197 keys).
    // (577 keys on 16 waves: 4.75 chunks per thread - one batch of 5 instead of a second batch for the last 0.75)
    const int nchunks = lk_pad * 8;
    const int stride = blockDim.x;
    auto stage = [&](auto sbc) {
        constexpr int SB = decltype(sbc)::value;
        for (int c0 = threadIdx.x; c0 < nchunks; c0 += SB * stride) {
            X8 kv[SB], vv[SB];
#pragma unroll
            for (int i = 0; i < SB; ++i) {
                const int c = c0 + i * stride;
                const int row = c >> 3, ch = c & 7;
#pragma unroll
                for (int j = 0; j < 8; ++j) { kv[i][j] = static_cast<T>(0.f); vv[i][j] = static_cast<T>(0.f); }
                if (c < nchunks && row < a.Lk) {
                    kv[i] = *reinterpret_cast<const X8*>(kb + (int64_t)row * a.k_rs + ch * 8);
                    vv[i] = *reinterpret_cast<const X8*>(vb + (int64_t)row * a.v_rs + ch * 8);
                }
            }
#pragma unroll
            for (int i = 0; i < SB; ++i) {
                const int c = c0 + i * stride;
                const int row = c >> 3, ch = c & 7;
                if (c < nchunks) {
                    *reinterpret_cast<X8*>(Ks + row * 128 + ((ch ^ ((row >> 1) & 7)) << 4)) = kv[i];
                    *reinterpret_cast<X8*>(Vs + row * 128 + ((ch * 16) ^ (((row >> 1) & 1) << 6))) = vv[i];
                }
            }
        }
    };
    if (nchunks > 4 * stride && nchunks <= 5 * stride) stage(std::integral_constant<int, 5>{});
    else stage(std::integral_constant<int, 4>{});
    __syncthreads();

    int voff[2];
    pv_lane_offsets(tr_lane_offset(lane), voff);
    // K fragment addresses: row (key0 + r), chunk (2*sx + hh) ^ ((row >> 1) & 7); key0 is a multiple of 32, so the swizzle
    // term depends on the lane only - four tile-invariant byte offsets per lane, one add per tile
    int koff[4];
#pragma unroll
    for (int sx = 0; sx < 4; ++sx) koff[sx] = r * 128 + (((2 * sx + hh) ^ ((r >> 1) & 7)) << 4);
    const float sl = a.scale * kLog2e;
    const int nkt = (a.Lk + 31) >> 5;
    for (int qt = wave; qt < a.nqt; qt += nwaves) {
        const int q0 = qt * 32;
        if (qt != wave) {   // later rounds (577 tokens: 19 tiles on 10 waves)
            const int qrow = min(q0 + r, a.Lq - 1);
#pragma unroll
            for (int s = 0; s < 4; ++s) qf[s] = *reinterpret_cast<const X8*>(qbase + (int64_t)qrow * a.q_rs + 16 * s);
        }
        Softmax st;
        st.init();
        const char* kt_base = Ks;
        const char* vt_base = Vs;
        for (int kt = 0; kt < nkt; ++kt, kt_base += 32 * 128, vt_base += 32 * 128) {
            const int key0 = kt * 32;
            f32x16 s;
#pragma unroll
            for (int i = 0; i < 16; ++i) s[i] = 0.f;
#pragma unroll
            for (int sx = 0; sx < 4; ++sx) {
                const X8 kf = *reinterpret_cast<const X8*>(kt_base + koff[sx]);
                s = Elem<T>::mfma32(kf, qf[sx], s);
            }
            X8 pf[2];
            softmax_tile<T, MASKED>(st, s, sl, mp, key0, hh, a.Lk, pf);
            pv_tile<T>(st, vt_base, voff, pf);
        }
        if (!a.wide_store) {             // several query tiles per wave (577 tokens), odd alignments: direct stores - 8-byte pieces, 32 rows per instruction
            if (q0 + r < a.Lq) {
                T* op = reinterpret_cast<T*>(a.out) + b1 * a.o_s1 + b0 * a.o_s0 + (int64_t)(q0 + r) * a.o_rs + h * 64 + 4 * hh;
                store_out<T>(st, op);
            }
        } else {
            // ONE tile per wave (197 tokens): the O tile goes out through LDS as WHOLE 128-byte rows (round 5).  The direct form writes a row's
            // 128 bytes as 16 pieces of 8 bytes from two lanes - 8 store instructions per wave that each touch 32 rows, 256 partial-line
            // requests where 32 full lines do; on the folded cross-attention kernel the CU's vector-memory request rate, not bytes or MFMAs,
            // turned out to set the pace, and this kernel's life is load - compute - store with two workgroups per CU.  K / V are dead once
            // every wave has finished its tile, so the tile (32 rows x 128 B, chunk position XOR (row & 7)) reuses their LDS.
            __syncthreads();
            char* ot = dyn + wave * 4096;
            const float inv = 1.0f / st.l_run;
#pragma unroll
            for (int dt = 0; dt < 2; ++dt)
#pragma unroll
                for (int qd = 0; qd < 4; ++qd) {
                    u32x2 p;
                    p.x = pack2<T>(st.o[dt][qd * 4 + 0] * inv, st.o[dt][qd * 4 + 1] * inv);
                    p.y = pack2<T>(st.o[dt][qd * 4 + 2] * inv, st.o[dt][qd * 4 + 3] * inv);
                    const int col = dt * 32 + 8 * qd + 4 * hh;                      // first of this piece's 4 features
                    *reinterpret_cast<u32x2*>(ot + r * 128 + ((((col >> 3) ^ (r & 7)) << 4) | ((col & 7) << 1))) = p;
                }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            T* ob = reinterpret_cast<T*>(a.out) + b1 * a.o_s1 + b0 * a.o_s0 + h * 64;
            const int ch = lane & 7;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int row = (lane >> 3) + 8 * i;
                const u32x4 v = *reinterpret_cast<const u32x4*>(ot + row * 128 + ((ch ^ (row & 7)) << 4));
                if (q0 + row < a.Lq) *reinterpret_cast<u32x4*>(ob + (int64_t)(q0 + row) * a.o_rs + ch * 8) = v;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
template <typename T, bool MASKED>
__global__ __launch_bounds__(256) void attn_stream_kernel(const AttnArgs a) {
    using X8 = typename Elem<T>::x8;
    __shared__ __attribute__((aligned(16))) char smem[4 * 4096];  // one 32-key x 64-dh V tile per wave

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int64_t unit = (int64_t)blockIdx.x * 4 + wave;
    if (unit >= a.total) return;  // whole wave leaves: EXEC stays all-ones for the transposed LDS reads

    const int qt = (int)(unit % a.nqt);
    int64_t t = unit / a.nqt;
    const int h = (int)(t % a.H);
    t /= a.H;
    const int b0 = (int)(t % a.B0);
    const int64_t b1 = t / a.B0;

    const int r = lane & 31, hh = lane >> 5;
    const int q0 = qt * 32;
    const int qrow = min(q0 + r, a.Lq - 1);
    const T* qp = reinterpret_cast<const T*>(a.q) + b1 * a.q_s1 + b0 * a.q_s0 + (int64_t)qrow * a.q_rs + h * 64 + 8 * hh;
    const int64_t kb1 = a.kv_index ? a.kv_index[b1] : b1;
    const T* kb = reinterpret_cast<const T*>(a.k) + kb1 * a.k_s1 + b0 * a.k_s0 + h * 64 + 8 * hh;
    const T* vb = reinterpret_cast<const T*>(a.v) + kb1 * a.v_s1 + b0 * a.v_s0 + h * 64;
    const float* mp = MASKED ? a.mask + b1 * a.m_s1 + b0 * a.m_s0 : nullptr;

    X8 qf[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) qf[s] = *reinterpret_cast<const X8*>(qp + 16 * s);

    char* vl = smem + wave * 4096;
    int voff[2];
    pv_lane_offsets(tr_lane_offset(lane), voff);
    const float sl = a.scale * kLog2e;
    const int nkt = (a.Lk + 31) >> 5;
    // V staging: lane covers 16-byte chunks c = lane + 64 i -> row c>>3, chunk c&7 (rows clamped: probability 0 there)
    const int vrow_l = lane >> 3, vch = lane & 7;

    auto load_tile = [&](int kt, X8 (&kf)[4], X8 (&vr)[4]) {
        const int key0 = kt * 32;
        const T* kp = kb + (int64_t)min(key0 + r, a.Lk - 1) * a.k_rs;
#pragma unroll
        for (int s = 0; s < 4; ++s) kf[s] = *reinterpret_cast<const X8*>(kp + 16 * s);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int vrow = min(key0 + vrow_l + 8 * i, a.Lk - 1);
            vr[i] = *reinterpret_cast<const X8*>(vb + (int64_t)vrow * a.v_rs + vch * 8);
        }
    };

    Softmax st;
    st.init();
    X8 kf[4], vr[4];
    load_tile(0, kf, vr);
    for (int kt = 0; kt < nkt; ++kt) {
        const int key0 = kt * 32;
        f32x16 s;
#pragma unroll
        for (int i = 0; i < 16; ++i) s[i] = 0.f;
#pragma unroll
        for (int sx = 0; sx < 4; ++sx) s = Elem<T>::mfma32(kf[sx], qf[sx], s);
        // V tile -> this wave's LDS tile (same row / half swizzle as the shared kernel)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = vrow_l + 8 * i;
            *reinterpret_cast<X8*>(vl + row * 128 + ((vch * 16) ^ (((row >> 1) & 1) << 6))) = vr[i];
        }
        if (kt + 1 < nkt) load_tile(kt + 1, kf, vr);   // prefetch the next tile under this tile's softmax / PV
        X8 pf[2];
        softmax_tile<T, MASKED>(st, s, sl, mp, key0, hh, a.Lk, pf);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        pv_tile<T>(st, vl, voff, pf);
        __builtin_amdgcn_wave_barrier();  // keep the next tile's LDS writes behind these reads
    }
    if (q0 + r < a.Lq) {
        T* op = reinterpret_cast<T*>(a.out) + b1 * a.o_s1 + b0 * a.o_s0 + (int64_t)(q0 + r) * a.o_rs + h * 64 + 4 * hh;
        store_out<T>(st, op);
    }
}


// ---------------------------------------------------------------------------------------------------------------------
// attn_split32_kernel (round 6): cir_attention_split8's kernel - fp32 q / k / v (the text side of the text32 mode: 32 x 32 self-attention per
// head), fp32 softmax, context out as "split8" operand rows - with BOTH products on the fp16 MFMA as three-term sums of (hi, lo) fp16 pairs
// formed in registers:  S = K_hi Q_hi + K_lo Q_hi + K_hi Q_lo,  O = V_hi P_hi + V_lo P_hi + V_hi P_lo  (each ~2^-21 of the product; the
// MFMA keeps fp16 subnormals).  12 + 12 MFMAs of 32 cycles per 32-key tile instead of 32 + 32 f32-input MFMAs of 64 cycles: the fp32
// kernel (attn_f32_kernel, still the exact mode's) spent 14 ms of a text32 step at half of the f32 matrix pipe's rate.
// Structure = attn_stream_kernel: one wave per (item, head, 32 queries), K fragments straight from global memory, V through private LDS
// tiles (one for the hi terms, one for the lo terms), next tile prefetched under the current tile's softmax.
template <bool MASKED>
__global__ __launch_bounds__(256) void attn_split32_kernel(const AttnArgs a) {
    using X8 = f16x8;
    __shared__ __attribute__((aligned(16))) char smem[4 * 2 * 4096];   // per wave: V_hi tile, V_lo tile (32 keys x 64 dh each)
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int64_t unit = (int64_t)blockIdx.x * 4 + wave;
    if (unit >= a.total) return;
    const int qt = (int)(unit % a.nqt);
    int64_t t = unit / a.nqt;
    const int h = (int)(t % a.H);
    t /= a.H;
    const int b0 = (int)(t % a.B0);
    const int64_t b1 = t / a.B0;
    const int r = lane & 31, hh = lane >> 5;
    const int q0 = qt * 32;
    const int qrow = min(q0 + r, a.Lq - 1);
    const float* qp = reinterpret_cast<const float*>(a.q) + b1 * a.q_s1 + b0 * a.q_s0 + (int64_t)qrow * a.q_rs + h * 64 + 8 * hh;
    const float* kb = reinterpret_cast<const float*>(a.k) + b1 * a.k_s1 + b0 * a.k_s0 + h * 64 + 8 * hh;
    const float* vb = reinterpret_cast<const float*>(a.v) + b1 * a.v_s1 + b0 * a.v_s0 + h * 64;
    const float* mp = MASKED ? a.mask + b1 * a.m_s1 + b0 * a.m_s0 : nullptr;

    // eight consecutive fp32 values -> (hi, lo) fp16 fragments
    auto split = [](const float4& x0, const float4& x1, X8& hi, X8& lo) {
        const float v[8] = {x0.x, x0.y, x0.z, x0.w, x1.x, x1.y, x1.z, x1.w};
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const _Float16 hj = (_Float16)v[j];
            hi[j] = hj;
            lo[j] = (_Float16)(v[j] - (float)hj);
        }
    };
    X8 qh[4], ql[4];
#pragma unroll
    for (int sx = 0; sx < 4; ++sx)
        split(*reinterpret_cast<const float4*>(qp + 16 * sx), *reinterpret_cast<const float4*>(qp + 16 * sx + 4), qh[sx], ql[sx]);

    char* vt_hi = smem + wave * 8192;
    char* vt_lo = vt_hi + 4096;
    int voff[2];
    pv_lane_offsets(tr_lane_offset(lane), voff);
    const float sl = a.scale * kLog2e;
    const int nkt = (a.Lk + 31) >> 5;
    const int vrow_l = lane >> 3, vch = lane & 7;
    float4 kr[4][2], vr[4][2];
    auto load_tile = [&](int kt) {
        const int key0 = kt * 32;
        const float* kp = kb + (int64_t)min(key0 + r, a.Lk - 1) * a.k_rs;
#pragma unroll
        for (int sx = 0; sx < 4; ++sx) { kr[sx][0] = *reinterpret_cast<const float4*>(kp + 16 * sx); kr[sx][1] = *reinterpret_cast<const float4*>(kp + 16 * sx + 4); }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float* vp = vb + (int64_t)min(key0 + vrow_l + 8 * i, a.Lk - 1) * a.v_rs + vch * 8;
            vr[i][0] = *reinterpret_cast<const float4*>(vp);
            vr[i][1] = *reinterpret_cast<const float4*>(vp + 4);
        }
    };
    Softmax st;
    st.init();
    load_tile(0);
    for (int kt = 0; kt < nkt; ++kt) {
        const int key0 = kt * 32;
        f32x16 s;
#pragma unroll
        for (int i = 0; i < 16; ++i) s[i] = 0.f;
#pragma unroll
        for (int sx = 0; sx < 4; ++sx) {
            X8 kh, kl;
            split(kr[sx][0], kr[sx][1], kh, kl);
            s = Elem<_Float16>::mfma32(kl, qh[sx], s);
            s = Elem<_Float16>::mfma32(kh, ql[sx], s);
            s = Elem<_Float16>::mfma32(kh, qh[sx], s);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            X8 vh, vl;
            split(vr[i][0], vr[i][1], vh, vl);
            const int row = vrow_l + 8 * i;
            const int off = row * 128 + ((vch * 16) ^ (((row >> 1) & 1) << 6));
            *reinterpret_cast<X8*>(vt_hi + off) = vh;
            *reinterpret_cast<X8*>(vt_lo + off) = vl;
        }
        if (kt + 1 < nkt) load_tile(kt + 1);
        // ---- online softmax (log2 domain), probabilities kept in fp32 ----
        float sv[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int key = key0 + (i & 3) + 8 * (i >> 2) + 4 * hh;
            float x = s[i] * sl;
            if constexpr (MASKED) x = fmaf(fmaxf(mp[min(key, a.Lk - 1)], -2.0e38f), kLog2e, x);
            sv[i] = key < a.Lk ? x : -INFINITY;
        }
        float mx = sv[0];
#pragma unroll
        for (int i = 1; i < 16; ++i) mx = fmaxf(mx, sv[i]);
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        const float m_new = fmaxf(st.m_run, mx);
        const float alpha = exp2f(st.m_run - m_new);
        st.m_run = m_new;
        float psum = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            sv[i] = exp2f(sv[i] - m_new);
            psum += sv[i];
        }
        psum += __shfl_xor(psum, 32, 64);
        st.l_run = st.l_run * alpha + psum;
#pragma unroll
        for (int i = 0; i < 16; ++i) { st.o[0][i] *= alpha; st.o[1][i] *= alpha; }
        X8 ph[2], pl[2];
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const _Float16 hj = (_Float16)sv[8 * s2 + j];
                ph[s2][j] = hj;
                pl[s2][j] = (_Float16)(sv[8 * s2 + j] - (float)hj);
            }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        pv_tile<_Float16>(st, vt_lo, voff, ph);
        pv_tile<_Float16>(st, vt_hi, voff, pl);
        pv_tile<_Float16>(st, vt_hi, voff, ph);
        __builtin_amdgcn_wave_barrier();
    }
    // ---- context / row sum -> split8 rows (as attn_f32_kernel's SPLIT epilogue: the half-waves exchange 4-element groups) ----
    const float inv = 1.0f / st.l_run;
    const int d_model = a.H * 64;
    char* row = reinterpret_cast<char*>(a.out) + b1 * a.o_s1 + b0 * a.o_s0 + (int64_t)min(q0 + r, a.Lq - 1) * a.o_rs;
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            float lo4[4], hi4[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) { lo4[j] = st.o[dt][(2 * p) * 4 + j] * inv; hi4[j] = st.o[dt][(2 * p + 1) * 4 + j] * inv; }
#pragma unroll
            for (int j = 0; j < 4; ++j) asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(lo4[j]), "+v"(hi4[j]));
            const int f0 = h * 64 + dt * 32 + 8 * (2 * p + hh);
            const Split4 s0 = split8_x4(lo4), s1 = split8_x4(hi4);
            if (q0 + r < a.Lq) {
                *reinterpret_cast<u32x4*>(row + 2 * f0) = u32x4{s0.h01, s0.h23, s1.h01, s1.h23};
                *reinterpret_cast<u32x2*>(row + 2 * d_model + f0) = u32x2{s0.lo8, s1.lo8};
                *reinterpret_cast<u32x2*>(row + 3 * d_model + f0) = u32x2{s0.hi8, s1.hi8};
            }
        }
}

// ---------------------------------------------------------------------------------------------------------------------
// cir_cls_cross_attention - cross-attention of ONE query row per (branch, head) over a candidate's image tokens, with the
// K and V projections folded out of the token side (last fusion layer: only the two CLS rows reach cls_head,
// nlvr_encoder.py:906-908, so per candidate there are 2 * H query vectors and N keys):
//   scores[r][j] = q_r . (W_k x_j + b_k) = (W_k^T q_r) . x_j + const_r      -> qp_r = W_k^T q_r   (a 64 x D GEMM per (branch, head))
//   ctx_r        = sum_j p[r][j] (W_v x_j + b_v) = W_v (sum_j p[r][j] x_j) + b_v  -> o_r = sum_j p[r][j] x_j   (THIS kernel), then a D x 64 GEMM
// The constant drops out of the softmax.  The kernel is attention with K = V = the raw tokens (width D = 128 * NO) shared by
// all R <= 32 query rows: one workgroup of 4 waves per candidate, 32-key token tiles staged once in LDS; wave w owns the
// k-slice / column slice [w * D/4, (w+1) * D/4): partial S^T tiles are summed through LDS, every wave repeats the (cheap)
// online softmax, and accumulates its 32 x D/4 slice of O^T with transposed LDS reads of the same tile.  HBM-bound on the
// token tensor (read once) instead of the 4 D x D projection of every token of the layer.
template <typename T, int NO>
__global__ __launch_bounds__(256, 2) void cls_xattn_kernel(const T* __restrict__ x, int64_t x_s1, const int64_t* __restrict__ x_index,
                                                           const T* __restrict__ qp, T* __restrict__ out, int Lk, float scale) {
    using X8 = typename Elem<T>::x8;
    constexpr int D = 128 * NO, SL = 32 * NO;            // token width, columns (= k-slice) per wave
    constexpr int RS = D * 2 + 16;                         // padded LDS row: consecutive keys 4 banks apart
    constexpr int CH = D / 8;                              // 16-byte chunks per token row
    constexpr int PER = (32 * CH + 255) / 256;             // staging chunks per thread and tile
    __shared__ __attribute__((aligned(16))) char xs[32 * RS];
    __shared__ __attribute__((aligned(16))) float exch[4][16][64];

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int r = lane & 31, hh = lane >> 5;
    const int64_t t = blockIdx.x;
    const T* xb = x + (x_index ? x_index[t] : t) * x_s1;     // optional gather: item t attends the tokens of bank row x_index[t]
    const T* qb = qp + t * (32 * D);
    T* ob = out + t * (32 * D);
    const int nkt = (Lk + 31) >> 5;

    // Q' fragments of this wave's k-slice: B operand of S^T = X_tile * Q'^T (column = query row r, 8 consecutive k per lane)
    X8 qf[2 * NO];
#pragma unroll
    for (int ks = 0; ks < 2 * NO; ++ks) qf[ks] = *reinterpret_cast<const X8*>(qb + r * D + wave * SL + 16 * ks + 8 * hh);

    X8 stage[PER];
    auto load_tile = [&](int kt) {
#pragma unroll
        for (int i = 0; i < PER; ++i) {
            const int c = threadIdx.x + i * 256;
            const int row = c / CH, ch = c - row * CH;
            if (c < 32 * CH) stage[i] = *reinterpret_cast<const X8*>(xb + (int64_t)min(kt * 32 + row, Lk - 1) * D + ch * 8);
        }
    };
    auto store_tile = [&]() {
#pragma unroll
        for (int i = 0; i < PER; ++i) {
            const int c = threadIdx.x + i * 256;
            const int row = c / CH, ch = c - row * CH;
            if (c < 32 * CH) *reinterpret_cast<X8*>(xs + row * RS + ch * 16) = stage[i];
        }
    };

    float m_run = -INFINITY, l_run = 0.f;
    f32x16 o[NO];
#pragma unroll
    for (int dt = 0; dt < NO; ++dt)
#pragma unroll
        for (int i = 0; i < 16; ++i) o[dt][i] = 0.f;
    const float sl = scale * kLog2e;
    // transposed-read lane offset inside a 4-key x 16-column block (see tr_lane_offset), for this tile's row stride
    const int i16 = lane & 15;
    const int troff = (4 * hh + (i16 >> 2)) * RS + (16 * ((lane >> 4) & 1) + 4 * (i16 & 3)) * 2 + wave * SL * 2;

    load_tile(0);
    for (int kt = 0; kt < nkt; ++kt) {
        __syncthreads();                                   // every wave is done with the previous tile
        store_tile();
        __syncthreads();
        if (kt + 1 < nkt) load_tile(kt + 1);               // next tile in flight under this tile's arithmetic
        // ---- partial S^T over this wave's k-slice ----
        f32x16 s;
#pragma unroll
        for (int i = 0; i < 16; ++i) s[i] = 0.f;
#pragma unroll
        for (int ks = 0; ks < 2 * NO; ++ks) {
            const X8 kf = *reinterpret_cast<const X8*>(xs + r * RS + (wave * SL + 16 * ks + 8 * hh) * 2);
            s = Elem<T>::mfma32(kf, qf[ks], s);
        }
#pragma unroll
        for (int i = 0; i < 16; ++i) exch[wave][i][lane] = s[i];
        __syncthreads();
        float sv[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) sv[i] = exch[0][i][lane] + exch[1][i][lane] + exch[2][i][lane] + exch[3][i][lane];
        // ---- online softmax (log2 domain; lane = query row r, 16 of the tile's 32 keys) ----
        const int key0 = kt * 32;
        if (key0 + 32 > Lk) {
#pragma unroll
            for (int i = 0; i < 16; ++i) sv[i] = (key0 + (i & 3) + 8 * (i >> 2) + 4 * hh) < Lk ? sv[i] : -INFINITY;
        }
        float mx = sv[0];
#pragma unroll
        for (int i = 1; i < 16; ++i) mx = fmaxf(mx, sv[i]);
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64)) * sl;
        const float m_new = fmaxf(m_run, mx);
        const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
        m_run = m_new;
        float psum = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            sv[i] = __builtin_amdgcn_exp2f(fmaf(sv[i], sl, -m_new));
            psum += sv[i];
        }
        psum += __shfl_xor(psum, 32, 64);
        l_run = l_run * alpha + psum;
        X8 pf[2];
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
            for (int j = 0; j < 8; ++j) pf[s2][j] = static_cast<T>(sv[8 * s2 + j]);
        // ---- O^T slice += X_tile^T * P^T ----
#pragma unroll
        for (int dt = 0; dt < NO; ++dt) {
#pragma unroll
            for (int i = 0; i < 16; ++i) o[dt][i] *= alpha;
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                const char* base = xs + troff + dt * 64 + (16 * s2) * RS;
                const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(base));
                const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(base + 8 * RS));
                s16x8 both;
                both.s0 = lo.x; both.s1 = lo.y; both.s2 = lo.z; both.s3 = lo.w;
                both.s4 = hi.x; both.s5 = hi.y; both.s6 = hi.z; both.s7 = hi.w;
                o[dt] = Elem<T>::mfma32(__builtin_bit_cast(X8, both), pf[s2], o[dt]);
            }
        }
    }
    const float inv = 1.0f / l_run;
    T* op = ob + r * D + wave * SL + 4 * hh;
#pragma unroll
    for (int dt = 0; dt < NO; ++dt)
#pragma unroll
        for (int qd = 0; qd < 4; ++qd) {
            u32x2 p;
            p.x = pack2<T>(o[dt][qd * 4 + 0] * inv, o[dt][qd * 4 + 1] * inv);
            p.y = pack2<T>(o[dt][qd * 4 + 2] * inv, o[dt][qd * 4 + 3] * inv);
            *reinterpret_cast<u32x2*>(op + dt * 32 + 8 * qd) = p;
        }
}

}  // namespace cir

extern "C" int cir_attention(const void* q, int64_t q_s1, int64_t q_s0, int64_t q_rs, const void* k, int64_t k_s1,
                             int64_t k_s0, int64_t k_rs, const void* v, int64_t v_s1, int64_t v_s0, int64_t v_rs,
                             const float* mask, int64_t m_s1, int64_t m_s0, const int64_t* kv_index, void* out, int64_t o_s1,
                             int64_t o_s0, int64_t o_rs, int B1, int B0, int H, int Lq, int Lk, float scale, int dtype, void* stream) {
    using namespace cir;
    CIR_CHECK_PTR(q); CIR_CHECK_PTR(k); CIR_CHECK_PTR(v); CIR_CHECK_PTR(out);
    if (B1 <= 0 || B0 <= 0 || H <= 0 || Lq <= 0 || Lk <= 0) return CIR_EINVAL;
    if (dtype != CIR_BF16 && dtype != CIR_F16 && dtype != CIR_F32) return CIR_EDTYPE;
    const int64_t strides[] = {q_s1, q_s0, q_rs, k_s1, k_s0, k_rs, v_s1, v_s0, v_rs};
    const int vec = dtype == CIR_F32 ? 4 : 8;       // elements per 16-byte load
    for (int64_t s : strides)
        if (s % vec) return CIR_EALIGN;
    if (o_s1 % 4 || o_s0 % 4 || o_rs % 4) return CIR_EALIGN;
    if (!cir_aligned16(q) || !cir_aligned16(k) || !cir_aligned16(v) || (reinterpret_cast<uintptr_t>(out) & (dtype == CIR_F32 ? 15 : 7))) return CIR_EALIGN;
    AttnArgs a;
    a.q = q; a.q_s1 = q_s1; a.q_s0 = q_s0; a.q_rs = q_rs;
    a.k = k; a.k_s1 = k_s1; a.k_s0 = k_s0; a.k_rs = k_rs;
    a.v = v; a.v_s1 = v_s1; a.v_s0 = v_s0; a.v_rs = v_rs;
    a.mask = mask; a.m_s1 = m_s1; a.m_s0 = m_s0;
    a.kv_index = kv_index;
    a.out = out; a.o_s1 = o_s1; a.o_s0 = o_s0; a.o_rs = o_rs;
    a.B0 = B0; a.H = H; a.Lq = Lq; a.Lk = Lk; a.nqt = (Lq + 31) / 32;
    a.total = (int64_t)B1 * B0 * H * a.nqt;
    a.scale = scale;
    a.wide_store = 0;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (dtype == CIR_F32) return launch_attention_f32(a, s);      // "exact" mode: fp32 operands in both products (attention_f32.hip)
    const int lk_pad = (Lk + 31) & ~31;
    // K/V of a head are shared by its query tiles: stage them once per workgroup when there are several tiles
    // (up to 608 keys = 152 KiB of LDS: the 577-token ViT of the reference's 384-px scripts still fits one CU)
    const int shared_max = (g_tune[CIR_TUNE_ATTN_SHARED_MAX] == 0 || g_tune[CIR_TUNE_ATTN_SHARED_MAX] == -2) ? 608 : g_tune[CIR_TUNE_ATTN_SHARED_MAX];   // (-1: never)
    const bool shared = a.nqt >= 2 && lk_pad <= shared_max;
    if (shared) {
        const int64_t nblk = (int64_t)B1 * B0 * H;
        if (nblk > 0x7fffffff) return CIR_ESHAPE;
        // query tiles are dealt round-robin to the waves: as few rounds as 16 waves allow, then as few waves as that needs
        // (round 6: the busiest SIMD sets the time - waves w, w + 4, .. share one - so: the wave count <= 16 whose round-robin deal leaves
        // the smallest number of tiles on any SIMD, the larger count on ties: 19 tiles -> 16 waves, 5 / 5 / 5 / 4 tiles per SIMD where 10
        // waves gave 6 / 6 / 4 / 3; 3374 -> 3087 us per 1696 images at 577 tokens.  Which wave runs a tile does not change its bits.)
        const int rounds = (a.nqt + 15) / 16;
        int waves = a.nqt;
        if (a.nqt > 16) {
            int best = 1 << 30;
            for (int w = 4; w <= 16; ++w) {
                int load[4] = {0, 0, 0, 0};
                for (int qt = 0; qt < a.nqt; ++qt) load[(qt % w) & 3]++;
                const int mx = std::max(std::max(load[0], load[1]), std::max(load[2], load[3]));
                if (mx <= best) { best = mx; waves = w; }
            }
        }
        dim3 grid((unsigned)nblk), block(waves * 64);
        const size_t lds = (size_t)lk_pad * 256;
        // whole-row output stores through the (then dead) K / V region: exactly one tile per wave (every wave reaches the barrier once),
        // 16-byte-aligned output rows, 4 KiB of LDS per wave
        a.wide_store = rounds == 1 && waves == a.nqt && (size_t)waves * 4096 <= lds && cir_aligned16(out) && o_s1 % 8 == 0 && o_s0 % 8 == 0 && o_rs % 8 == 0;
        const bool bf = dtype == CIR_BF16, mk = mask != nullptr;
        const void* fn = bf ? (mk ? reinterpret_cast<const void*>(&attn_shared_kernel<__bf16, true>) : reinterpret_cast<const void*>(&attn_shared_kernel<__bf16, false>))
                            : (mk ? reinterpret_cast<const void*>(&attn_shared_kernel<_Float16, true>) : reinterpret_cast<const void*>(&attn_shared_kernel<_Float16, false>));
        if (lds > 64 * 1024) {   // opt in to more than 64 KiB of dynamic LDS (idempotent, per function)
            const hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e != hipSuccess) return (int)e;
        }
        if (bf) { if (mk) hipLaunchKernelGGL((attn_shared_kernel<__bf16, true>), grid, block, lds, s, a, lk_pad); else hipLaunchKernelGGL((attn_shared_kernel<__bf16, false>), grid, block, lds, s, a, lk_pad); }
        else { if (mk) hipLaunchKernelGGL((attn_shared_kernel<_Float16, true>), grid, block, lds, s, a, lk_pad); else hipLaunchKernelGGL((attn_shared_kernel<_Float16, false>), grid, block, lds, s, a, lk_pad); }
        CIR_LAUNCH_RESULT();
    }
    const int64_t nblk = (a.total + 3) / 4;
    if (nblk > 0x7fffffff) return CIR_ESHAPE;
    dim3 grid((unsigned)nblk), block(256);
    const bool mk = mask != nullptr;
    if (dtype == CIR_BF16) { if (mk) hipLaunchKernelGGL((attn_stream_kernel<__bf16, true>), grid, block, 0, s, a); else hipLaunchKernelGGL((attn_stream_kernel<__bf16, false>), grid, block, 0, s, a); }
    else { if (mk) hipLaunchKernelGGL((attn_stream_kernel<_Float16, true>), grid, block, 0, s, a); else hipLaunchKernelGGL((attn_stream_kernel<_Float16, false>), grid, block, 0, s, a); }
    CIR_LAUNCH_RESULT();
}

extern "C" int cir_attention_split8(const float* q, int64_t q_s1, int64_t q_s0, int64_t q_rs, const float* k, int64_t k_s1, int64_t k_s0, int64_t k_rs,
                                   const float* v, int64_t v_s1, int64_t v_s0, int64_t v_rs, const float* mask, int64_t m_s1, int64_t m_s0,
                                   void* out, int64_t o_s1_bytes, int64_t o_s0_bytes, int64_t o_rs_bytes, int B1, int B0, int H, int Lq, int Lk,
                                   float scale, void* stream) {
    using namespace cir;
    CIR_CHECK_PTR(q); CIR_CHECK_PTR(k); CIR_CHECK_PTR(v); CIR_CHECK_PTR(out);
    if (B1 <= 0 || B0 <= 0 || H <= 0 || Lq <= 0 || Lk <= 0) return CIR_EINVAL;
    const int64_t strides[] = {q_s1, q_s0, q_rs, k_s1, k_s0, k_rs, v_s1, v_s0, v_rs};
    for (int64_t s : strides)
        if (s % 4) return CIR_EALIGN;
    if (o_s1_bytes % 16 || o_s0_bytes % 16 || o_rs_bytes % 16 || o_rs_bytes < 4 * (int64_t)H * 64) return CIR_EALIGN;
    if (!cir_aligned16(q) || !cir_aligned16(k) || !cir_aligned16(v) || !cir_aligned16(out)) return CIR_EALIGN;
    AttnArgs a;
    a.q = q; a.q_s1 = q_s1; a.q_s0 = q_s0; a.q_rs = q_rs;
    a.k = k; a.k_s1 = k_s1; a.k_s0 = k_s0; a.k_rs = k_rs;
    a.v = v; a.v_s1 = v_s1; a.v_s0 = v_s0; a.v_rs = v_rs;
    a.mask = mask; a.m_s1 = m_s1; a.m_s0 = m_s0;
    a.kv_index = nullptr;
    a.out = out; a.o_s1 = o_s1_bytes; a.o_s0 = o_s0_bytes; a.o_rs = o_rs_bytes;
    a.B0 = B0; a.H = H; a.Lq = Lq; a.Lk = Lk; a.nqt = (Lq + 31) / 32;
    a.total = (int64_t)B1 * B0 * H * a.nqt;
    a.scale = scale;
    a.wide_store = 0;
    a.out_split = 1;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (g_tune[CIR_TUNE_ATTN_SHARED_MAX] == -2) return launch_attention_f32(a, s);   // (A/B and tests: the f32-input MFMA form of the same op)
    const int64_t nblk = (a.total + 3) / 4;
    if (nblk > 0x7fffffff) return CIR_ESHAPE;
    dim3 grid((unsigned)nblk), block(256);
    if (mask) hipLaunchKernelGGL((attn_split32_kernel<true>), grid, block, 0, s, a);
    else hipLaunchKernelGGL((attn_split32_kernel<false>), grid, block, 0, s, a);
    CIR_LAUNCH_RESULT();
}

extern "C" int cir_cls_cross_attention(const void* x, int64_t x_s1, const int64_t* x_index, const void* qp, void* out, int T, int Lk, int D,
                                       float scale, int dtype, void* stream) {
    using namespace cir;
    CIR_CHECK_PTR(x); CIR_CHECK_PTR(qp); CIR_CHECK_PTR(out);
    if (T <= 0 || Lk <= 0) return CIR_EINVAL;
    if (D % 128 != 0 || D < 128 || D > 768) return CIR_ESHAPE;
    if (dtype != CIR_BF16 && dtype != CIR_F16) return CIR_EDTYPE;
    if (!cir_aligned16(x) || !cir_aligned16(qp) || !cir_aligned16(out) || x_s1 % 8) return CIR_EALIGN;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    dim3 grid((unsigned)T), block(256);
#define CIR_CLSX(TT, NO) hipLaunchKernelGGL((cls_xattn_kernel<TT, NO>), grid, block, 0, s, reinterpret_cast<const TT*>(x), x_s1, x_index, \
                                            reinterpret_cast<const TT*>(qp), reinterpret_cast<TT*>(out), Lk, scale)
#define CIR_CLSX_NO(TT) switch (D / 128) { case 1: CIR_CLSX(TT, 1); break; case 2: CIR_CLSX(TT, 2); break; case 3: CIR_CLSX(TT, 3); break; \
                                            case 4: CIR_CLSX(TT, 4); break; case 5: CIR_CLSX(TT, 5); break; default: CIR_CLSX(TT, 6); break; }
    if (dtype == CIR_BF16) { CIR_CLSX_NO(__bf16) } else { CIR_CLSX_NO(_Float16) }
#undef CIR_CLSX_NO
#undef CIR_CLSX
    CIR_LAUNCH_RESULT();
}
