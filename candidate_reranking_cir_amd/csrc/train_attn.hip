// Fused attention of the TRAINING pass (nlvr_encoder.py:140-222 in train() mode and its adjoint): forward with
// key mask + dropout on the probabilities + a log-sum-exp per row, and a RECOMPUTING backward (dQ, dK, dV from Q, K, V, O, dO, LSE; the
// dropout mask is regenerated from its counter) - no score / probability tensor ever exists in memory (the un-fused pass of round 3
// materialised 9 GB of fp32 scores and their adjoints per step at 577 keys).  Head dimension 64, 16-bit operands, fp32 statistics.
//
// All three kernels are the streaming inference kernel's tile update (attention.hip: one wave per 32-row tile, the contraction partner
// streamed in 32-row tiles, scores TRANSPOSED so that a lane owns one column and the packed probabilities are already the B operand of
// the next product, the other operand of that product read transposed from a row-major LDS tile with ds_read_b64_tr_b16) with the roles
// of the tensors permuted:
//   forward  wave = 32 queries, streams keys:  S^T = K Q^T      P^T = softmax            O^T  += V^T  Pd^T      (V tile in LDS)
//   dQ       wave = 32 queries, streams keys:  S^T = K Q^T      dPd^T = V dO^T           dQ^T += K^T  dS^T      (K tile in LDS)
//   dK, dV   wave = 32 keys, streams queries:  S   = Q K^T      dPd   = dO V^T           dV^T += dO^T Pd,  dK^T += Q^T dS   (Q, dO tiles in LDS)
// with P = exp2(S * scale*log2e + mask*log2e - LSE2), Pd = dropout(P), dS = scale * P * (dropout'(dPd) - D), D = rowsum(dO * O).
// Tensors are head views (group, head, row, 64) given by three element strides each; rows beyond an extent are clamped on load and
// their probabilities forced to zero.  Dropout: element (row = (group * H + head) * Lq + query, col = key) of common.hpp's pair hash.
#include <type_traits>

#include "common.hpp"

namespace cir {

constexpr float kLog2eT = 1.4426950408889634f;
typedef __attribute__((address_space(3))) s16x4* lds_s16x4_ptr_t;

struct TAttnArgs {
    const void* q; int64_t q_sg, q_sh, q_sr;
    const void* k; int64_t k_sg, k_sh, k_sr;
    const void* v; int64_t v_sg, v_sh, v_sr;
    const void* o; const void* d_o; int64_t o_sg, o_sh, o_sr;     // O (forward: written) and dO share one layout
    float* o32;                                                   // optional fp32 twin of O in the same layout: D = rowsum(dO * O) is then formed from it
    const float* mask;                                            // additive key mask (G, Lk) contiguous, or null
    float* lse;                                                   // (G, H, Lq) log2-domain log-sum-exp (forward: written)
    float* dsum;                                                  // (G, H, Lq) D = rowsum(dO * O) (dQ kernel: written; dK/dV kernel: read)
    void* dq; int64_t dq_sg, dq_sh, dq_sr;                         // gradients: fp32, or (grad16) in the operand type - the next
    void* dk; int64_t dk_sg, dk_sh, dk_sr;                         // dense layer's dgrad / wgrad operand as it is
    void* dv; int64_t dv_sg, dv_sh, dv_sr;
    int grad16;
    int G, H, Lq, Lk, nqt, nkt;
    float scale, p_drop;
    uint64_t seed;
};

// this lane's address inside a 4-row x 16-column transposed-read block (see attention.hip: tr_lane_offset / pv_lane_offsets)
__device__ __forceinline__ void tr_offsets(int lane, int (&voff)[2]) {
    const int i16 = lane & 15, hh = lane >> 5;
    const int q = i16 >> 2, p = i16 & 3;
    const int col = ((16 * ((lane >> 4) & 1) + 4 * p) * 2) ^ ((q >> 1) << 6);
    const int off = (4 * hh + q) * 128 + col;
#pragma unroll
    for (int dt = 0; dt < 2; ++dt) voff[dt] = ((off & 0x7f) ^ (dt * 64)) + (off & ~0x7f);
}

// acc^T[64 dh][32 cols] += tile^T[dh][32 rows] * frag[rows][cols]: `tile` = 32 rows x 64 dh row-major in LDS (128-byte rows, halves swapped
// on rows with bit 1 set), frag = the packed accumulator-layout values of the 32 x 32 score-like tile (k-slot order of attention.hip)
template <typename T>
__device__ __forceinline__ void tr_accumulate(f32x16 (&acc)[2], const char* tile, const int (&voff)[2], const typename Elem<T>::x8 (&frag)[2]) {
    using X8 = typename Elem<T>::x8;
#pragma unroll
    for (int dt = 0; dt < 2; ++dt) {
        const char* base0 = tile + voff[dt];
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
            const char* base = base0 + (16 * s2) * 128;
            const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr_t)(base));
            const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr_t)(base + 8 * 128));
            s16x8 both;
            both.s0 = lo.x; both.s1 = lo.y; both.s2 = lo.z; both.s3 = lo.w;
            both.s4 = hi.x; both.s5 = hi.y; both.s6 = hi.z; both.s7 = hi.w;
            acc[dt] = Elem<T>::mfma32(__builtin_bit_cast(X8, both), frag[s2], acc[dt]);
        }
    }
}

// row-major 32 x 64 tile of 16-bit rows (row r of the source at src + min(row0 + r, rows - 1) * stride) -> registers -> the wave's LDS tile
template <typename T>
__device__ __forceinline__ void tile_load(const T* src, int64_t stride, int row0, int rows, int lane, typename Elem<T>::x8 (&reg)[4]) {
    using X8 = typename Elem<T>::x8;
    const int rl = lane >> 3, ch = lane & 7;
#pragma unroll
    for (int i = 0; i < 4; ++i) reg[i] = *reinterpret_cast<const X8*>(src + (int64_t)min(row0 + rl + 8 * i, rows - 1) * stride + ch * 8);
}
template <typename T>
__device__ __forceinline__ void tile_store(char* tile, int lane, const typename Elem<T>::x8 (&reg)[4]) {
    using X8 = typename Elem<T>::x8;
    const int rl = lane >> 3, ch = lane & 7;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = rl + 8 * i;
        *reinterpret_cast<X8*>(tile + row * 128 + ((ch * 16) ^ (((row >> 1) & 1) << 6))) = reg[i];
    }
}
// fragments of rows (lane & 31) of a row-major matrix: d-chunks 8 hh + 16 s (A operand rows / B operand columns of mfma_f32_32x32x16)
template <typename T>
__device__ __forceinline__ void frag_load(const T* row_ptr, typename Elem<T>::x8 (&f)[4]) {
    using X8 = typename Elem<T>::x8;
#pragma unroll
    for (int s = 0; s < 4; ++s) f[s] = *reinterpret_cast<const X8*>(row_ptr + 16 * s);
}
// accumulator index i of a 32 x 32 tile -> row inside the tile (this lane's column is lane & 31)
__device__ __forceinline__ int acc_row(int i, int hh) { return (i & 3) + 8 * (i >> 2) + 4 * hh; }

// fp32 store of a transposed accumulator: element (row = lane & 31, dh = 32 dt + 8 qd + 4 hh + j)
__device__ __forceinline__ void store_f32(const f32x16 (&acc)[2], float* rowp /* + 4 hh */, float mul) {
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int qd = 0; qd < 4; ++qd)
            *reinterpret_cast<float4*>(rowp + dt * 32 + 8 * qd) =
                make_float4(acc[dt][qd * 4 + 0] * mul, acc[dt][qd * 4 + 1] * mul, acc[dt][qd * 4 + 2] * mul, acc[dt][qd * 4 + 3] * mul);
}

// gradient store of a transposed accumulator (same element map) at element offset `off` of `base`: fp32 or the operand type
template <typename T>
__device__ __forceinline__ void store_grad(const f32x16 (&acc)[2], void* base, int64_t off, int grad16, float mul) {
    if (!grad16) { store_f32(acc, reinterpret_cast<float*>(base) + off, mul); return; }
    typedef __attribute__((ext_vector_type(4))) T t4;
    T* rowp = reinterpret_cast<T*>(base) + off;
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int qd = 0; qd < 4; ++qd) {
            t4 o = {static_cast<T>(acc[dt][qd * 4 + 0] * mul), static_cast<T>(acc[dt][qd * 4 + 1] * mul), static_cast<T>(acc[dt][qd * 4 + 2] * mul),
                    static_cast<T>(acc[dt][qd * 4 + 3] * mul)};
            *reinterpret_cast<t4*>(rowp + dt * 32 + 8 * qd) = o;
        }
}

// ------------------------------------------------------------------------------------------------------------------ forward
template <typename T, bool MASKED>
__global__ __launch_bounds__(256) void tattn_fwd_kernel(const TAttnArgs a) {
    using X8 = typename Elem<T>::x8;
    __shared__ __attribute__((aligned(16))) char smem[4 * 4096];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int64_t unit = (int64_t)blockIdx.x * 4 + wave;
    if (unit >= (int64_t)a.G * a.H * a.nqt) return;
    const int qt = (int)(unit % a.nqt);
    const int64_t gh = unit / a.nqt;
    const int h = (int)(gh % a.H);
    const int64_t g = gh / a.H;
    const int r = lane & 31, hh = lane >> 5;
    const int q0 = qt * 32;
    const int qrow = min(q0 + r, a.Lq - 1);
    const T* kb = reinterpret_cast<const T*>(a.k) + g * a.k_sg + h * a.k_sh;
    const T* vb = reinterpret_cast<const T*>(a.v) + g * a.v_sg + h * a.v_sh;
    const float* mp = MASKED ? a.mask + g * a.Lk : nullptr;
    X8 qf[4];
    frag_load<T>(reinterpret_cast<const T*>(a.q) + g * a.q_sg + h * a.q_sh + (int64_t)qrow * a.q_sr + 8 * hh, qf);
    char* vl = smem + wave * 4096;
    int voff[2];
    tr_offsets(lane, voff);
    const float sl = a.scale * kLog2eT;
    const float keep = a.p_drop > 0.f ? 1.0f / (1.0f - a.p_drop) : 1.0f;
    const uint32_t rkey = drop_row_key(a.seed, (uint64_t)(gh * a.Lq + qrow)), thr = drop_threshold(a.p_drop);

    float m_run = -INFINITY, l_run = 0.f;
    f32x16 o[2];
#pragma unroll
    for (int i = 0; i < 16; ++i) { o[0][i] = 0.f; o[1][i] = 0.f; }
    X8 kf[4], vr[4];
    frag_load<T>(kb + (int64_t)min(r, a.Lk - 1) * a.k_sr + 8 * hh, kf);
    tile_load<T>(vb, a.v_sr, 0, a.Lk, lane, vr);
    for (int kt = 0; kt < a.nkt; ++kt) {
        const int key0 = kt * 32;
        f32x16 s;
#pragma unroll
        for (int i = 0; i < 16; ++i) s[i] = 0.f;
#pragma unroll
        for (int sx = 0; sx < 4; ++sx) s = Elem<T>::mfma32(kf[sx], qf[sx], s);
        tile_store<T>(vl, lane, vr);
        if (kt + 1 < a.nkt) {
            frag_load<T>(kb + (int64_t)min(key0 + 32 + r, a.Lk - 1) * a.k_sr + 8 * hh, kf);
            tile_load<T>(vb, a.v_sr, key0 + 32, a.Lk, lane, vr);
        }
        float sv[16];
        float mx = -INFINITY;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int key = key0 + acc_row(i, hh);
            float x = s[i] * sl;
            if constexpr (MASKED) x = fmaf(fmaxf(mp[min(key, a.Lk - 1)], -2.0e38f), kLog2eT, x);
            sv[i] = key < a.Lk ? x : -INFINITY;
            mx = fmaxf(mx, sv[i]);
        }
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        const float m_new = fmaxf(m_run, mx);                   // finite: every tile holds at least one key < Lk
        const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
        l_run *= alpha;
#pragma unroll
        for (int i = 0; i < 16; ++i) { o[0][i] *= alpha; o[1][i] *= alpha; }
        m_run = m_new;
        float psum = 0.f;
        uint32_t bits = 0;
        X8 pf[2];
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const float p = __builtin_amdgcn_exp2f(sv[i] - m_run);
            psum += p;
            const uint32_t key = key0 + acc_row(i, hh);
            if (thr != 0 && (i & 1) == 0) bits = drop_bits(rkey, key);   // keys (2j, 2j + 1) = accumulator rows (i, i + 1)
            pf[i >> 3][i & 7] = static_cast<T>(drop_kept(bits, key, thr) ? p * keep : 0.f);
        }
        psum += __shfl_xor(psum, 32, 64);
        l_run += psum;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        tr_accumulate<T>(o, vl, voff, pf);
        __builtin_amdgcn_wave_barrier();
    }
    if (q0 + r < a.Lq) {
        const float inv = 1.0f / l_run;
        T* op = const_cast<T*>(reinterpret_cast<const T*>(a.o)) + g * a.o_sg + h * a.o_sh + (int64_t)(q0 + r) * a.o_sr + 4 * hh;
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int qd = 0; qd < 4; ++qd) {
                u32x2 pk;
                pk.x = pack2<T>(o[dt][qd * 4 + 0] * inv, o[dt][qd * 4 + 1] * inv);
                pk.y = pack2<T>(o[dt][qd * 4 + 2] * inv, o[dt][qd * 4 + 3] * inv);
                *reinterpret_cast<u32x2*>(op + dt * 32 + 8 * qd) = pk;
            }
        if (a.o32 != nullptr) store_f32(o, a.o32 + g * a.o_sg + h * a.o_sh + (int64_t)(q0 + r) * a.o_sr + 4 * hh, inv);
        if (hh == 0) a.lse[gh * a.Lq + q0 + r] = m_run + log2f(l_run);
    }
}

// probability / adjoint of one score element (shared by both backward kernels).  The probability that multiplies (dP - D) is the one
// the FORWARD used: a kept element's is its 16-bit dropout(P) value scaled back - D = rowsum(dO * O) = sum_j Pd16_ij dPd_ij was formed
// with exactly those, so sum_j dS_ij = D (1 - sum_j P_ij) vanishes as the softmax Jacobian demands; with the unrounded fp32 p the row
// sums of dS are off by the rounding of P and the (column-summed) query-bias gradients lose a factor 2-3 of accuracy.
// Both kernels are VALU-bound (head dimension 64: 512 MFMA cycles per 32 x 32 tile against 16 score elements per lane), so the element
// chain is kept minimal: x = s * sl + c with c = mask - LSE prepared per row / column (-inf for rows or keys past the extents: p = 0, no
// `valid` selects), the softmax scale is applied ONCE to the dQ / dK accumulators at the store instead of to every dS element, and
// dS / scale = Pd16 * dPd - P_f * D needs one multiply and one fma.
struct Adj { float pd, ds; };
template <typename T, bool DROP>
__device__ __forceinline__ Adj adjoint(float s, float dp, float sl, float c, float dsum, float keep, float ikeep, bool kept) {
    const float p = __builtin_amdgcn_exp2f(fmaf(s, sl, c));
    Adj r;
    if constexpr (DROP) {
        r.pd = static_cast<float>(static_cast<T>(p * (kept ? keep : 0.f)));       // what the forward's P.V product consumed
        const float pf = kept ? r.pd * ikeep : p;
        r.ds = fmaf(r.pd, dp, -(pf * dsum));
    } else {
        r.pd = static_cast<float>(static_cast<T>(p));
        r.ds = r.pd * (dp - dsum);
    }
    return r;
}
// the 16 random bits of element (row key, column): column parity picks the half
__device__ __forceinline__ bool kept16(uint32_t bits, uint32_t shift, uint32_t thr) { return __builtin_amdgcn_ubfe(bits, shift, 16) >= thr; }

// ------------------------------------------------------------------------------------------------------------------ dQ (and D)
template <typename T, bool MASKED, bool DROP>
__global__ __launch_bounds__(256, 2) void tattn_bwd_dq_kernel(const TAttnArgs a) {
    using X8 = typename Elem<T>::x8;
    __shared__ __attribute__((aligned(16))) char smem[4 * 4096];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int64_t unit = (int64_t)blockIdx.x * 4 + wave;
    if (unit >= (int64_t)a.G * a.H * a.nqt) return;
    const int qt = (int)(unit % a.nqt);
    const int64_t gh = unit / a.nqt;
    const int h = (int)(gh % a.H);
    const int64_t g = gh / a.H;
    const int r = lane & 31, hh = lane >> 5;
    const int q0 = qt * 32;
    const int qrow = min(q0 + r, a.Lq - 1);
    const T* kb = reinterpret_cast<const T*>(a.k) + g * a.k_sg + h * a.k_sh;
    const T* vb = reinterpret_cast<const T*>(a.v) + g * a.v_sg + h * a.v_sh;
    const float* mp = MASKED ? a.mask + g * a.Lk : nullptr;
    X8 qf[4], dof[4];
    frag_load<T>(reinterpret_cast<const T*>(a.q) + g * a.q_sg + h * a.q_sh + (int64_t)qrow * a.q_sr + 8 * hh, qf);
    const int64_t orow = g * a.o_sg + h * a.o_sh + (int64_t)qrow * a.o_sr + 8 * hh;
    frag_load<T>(reinterpret_cast<const T*>(a.d_o) + orow, dof);
    // D = rowsum(dO * O): this lane holds 32 of the row's 64 dh of both
    float dsum = 0.f;
    {
        if (a.o32 != nullptr) {
            // O from its fp32 twin (the exact fp32 accumulation of the ROUNDED Pd and V the forward multiplied), dO as the 16-bit
            // operand the dPd product below consumes: D then equals sum_j Pd16_ij dPd_ij term for term - the rounding of O (no twin)
            // or an unrounded fp32 dO (a twin of dO) each break that identity and with it the zero row sums of dS
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int half = 0; half < 2; ++half) {
                    const float4 x = *reinterpret_cast<const float4*>(a.o32 + orow + 16 * s + 4 * half);
                    dsum = fmaf(x.x, static_cast<float>(dof[s][4 * half + 0]), fmaf(x.y, static_cast<float>(dof[s][4 * half + 1]),
                           fmaf(x.z, static_cast<float>(dof[s][4 * half + 2]), fmaf(x.w, static_cast<float>(dof[s][4 * half + 3]), dsum))));
                }
        } else {
            X8 of[4];
            frag_load<T>(reinterpret_cast<const T*>(a.o) + orow, of);
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int j = 0; j < 8; ++j) dsum = fmaf(static_cast<float>(dof[s][j]), static_cast<float>(of[s][j]), dsum);
        }
        dsum += __shfl_xor(dsum, 32, 64);
        if (hh == 0 && q0 + r < a.Lq) a.dsum[gh * a.Lq + q0 + r] = dsum;
    }
    const bool qvalid = q0 + r < a.Lq;
    const float nlse = qvalid ? -a.lse[gh * a.Lq + qrow] : -INFINITY;     // a query row past Lq: p = 0 everywhere
    char* kl = smem + wave * 4096;
    int voff[2];
    tr_offsets(lane, voff);
    const float sl = a.scale * kLog2eT;
    const float keep = a.p_drop > 0.f ? 1.0f / (1.0f - a.p_drop) : 1.0f, ikeep = 1.0f - a.p_drop;
    const uint32_t rkey = drop_row_key(a.seed, (uint64_t)(gh * a.Lq + qrow)), thr = drop_threshold(a.p_drop);

    f32x16 dq[2];
#pragma unroll
    for (int i = 0; i < 16; ++i) { dq[0][i] = 0.f; dq[1][i] = 0.f; }
    X8 kf[4], vf[4], kr[4];
    frag_load<T>(kb + (int64_t)min(r, a.Lk - 1) * a.k_sr + 8 * hh, kf);
    frag_load<T>(vb + (int64_t)min(r, a.Lk - 1) * a.v_sr + 8 * hh, vf);
    tile_load<T>(kb, a.k_sr, 0, a.Lk, lane, kr);
    for (int kt = 0; kt < a.nkt; ++kt) {
        const int key0 = kt * 32;
        f32x16 s, dp;
#pragma unroll
        for (int i = 0; i < 16; ++i) { s[i] = 0.f; dp[i] = 0.f; }
#pragma unroll
        for (int sx = 0; sx < 4; ++sx) { s = Elem<T>::mfma32(kf[sx], qf[sx], s); dp = Elem<T>::mfma32(vf[sx], dof[sx], dp); }
        tile_store<T>(kl, lane, kr);
        if (kt + 1 < a.nkt) {
            frag_load<T>(kb + (int64_t)min(key0 + 32 + r, a.Lk - 1) * a.k_sr + 8 * hh, kf);
            frag_load<T>(vb + (int64_t)min(key0 + 32 + r, a.Lk - 1) * a.v_sr + 8 * hh, vf);
            tile_load<T>(kb, a.k_sr, key0 + 32, a.Lk, lane, kr);
        }
        X8 dsf[2];
        const bool edge = MASKED || key0 + 32 > a.Lk;              // only the last key tile can hold keys past Lk
        auto elements = [&](auto edge_c) {
            uint32_t bits = 0;
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int key = key0 + acc_row(i, hh);
                float c = nlse;
                if constexpr (decltype(edge_c)::value) {
                    if constexpr (MASKED) c = fmaf(fmaxf(mp[min(key, a.Lk - 1)], -2.0e38f), kLog2eT, nlse);
                    c = key < a.Lk ? c : -INFINITY;
                }
                if (DROP && (i & 1) == 0) bits = drop_bits(rkey, key);
                const Adj ad = adjoint<T, DROP>(s[i], dp[i], sl, c, dsum, keep, ikeep, DROP ? kept16(bits, (i & 1) * 16, thr) : true);
                dsf[i >> 3][i & 7] = static_cast<T>(ad.ds);
            }
        };
        if (edge) elements(std::true_type{}); else elements(std::false_type{});
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        tr_accumulate<T>(dq, kl, voff, dsf);                    // dQ^T[dh][q] += K^T[dh][key] (dS / scale)^T[key][q]
        __builtin_amdgcn_wave_barrier();
    }
    if (qvalid) store_grad<T>(dq, a.dq, g * a.dq_sg + h * a.dq_sh + (int64_t)(q0 + r) * a.dq_sr + 4 * hh, a.grad16, a.scale);
}

// ------------------------------------------------------------------------------------------------------------------ dK, dV
// Per wave and query tile the 32 rows' -LSE and D are staged through 256 bytes of LDS (one coalesced load per lane, prefetched a tile
// ahead; eight ds_read_b128 hand every lane the values of its 16 accumulator rows) - per-element global gathers with clamped 64-bit
// addresses were a fifth of this kernel's VALU work.  Keys past Lk need no masking here: a lane owns ONE key column of dK / dV, and the
// columns of keys past the extent are simply not stored.
template <typename T, bool MASKED, bool DROP>
__global__ __launch_bounds__(256, 2) void tattn_bwd_dkv_kernel(const TAttnArgs a) {
    using X8 = typename Elem<T>::x8;
    __shared__ __attribute__((aligned(16))) char smem[4 * (2 * 4096 + 256)];   // per wave: a Q tile, a dO tile, the rows' statistics
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int64_t unit = (int64_t)blockIdx.x * 4 + wave;
    if (unit >= (int64_t)a.G * a.H * a.nkt) return;
    const int kt = (int)(unit % a.nkt);
    const int64_t gh = unit / a.nkt;
    const int h = (int)(gh % a.H);
    const int64_t g = gh / a.H;
    const int r = lane & 31, hh = lane >> 5;
    const int key0 = kt * 32;
    const int key = key0 + r;
    const int krow = min(key, a.Lk - 1);
    const bool kvalid = key < a.Lk;
    const T* qb = reinterpret_cast<const T*>(a.q) + g * a.q_sg + h * a.q_sh;
    const T* dob = reinterpret_cast<const T*>(a.d_o) + g * a.o_sg + h * a.o_sh;
    X8 kfb[4], vfb[4];                                          // this wave's 32 keys as B operands (columns)
    frag_load<T>(reinterpret_cast<const T*>(a.k) + g * a.k_sg + h * a.k_sh + (int64_t)krow * a.k_sr + 8 * hh, kfb);
    frag_load<T>(reinterpret_cast<const T*>(a.v) + g * a.v_sg + h * a.v_sh + (int64_t)krow * a.v_sr + 8 * hh, vfb);
    float maskv = 0.f;
    if constexpr (MASKED) maskv = fmaxf(a.mask[g * a.Lk + krow], -2.0e38f) * kLog2eT;
    char* ql = smem + wave * (2 * 4096 + 256);
    char* dl = ql + 4096;
    float* stat = reinterpret_cast<float*>(dl + 4096);          // [0, 32): -LSE of the tile's rows, [32, 64): their D
    int voff[2];
    tr_offsets(lane, voff);
    const float sl = a.scale * kLog2eT;
    const float keep = a.p_drop > 0.f ? 1.0f / (1.0f - a.p_drop) : 1.0f, ikeep = 1.0f - a.p_drop;
    const float* stp = (hh == 0 ? a.lse : a.dsum) + gh * a.Lq;
    auto stat_load = [&](int q0) -> float {
        const int qi = q0 + r;
        const float v = stp[min(qi, a.Lq - 1)];
        return hh == 0 ? (qi < a.Lq ? -v : -INFINITY) : v;      // a query row past Lq: p = 0
    };
    const uint32_t thr = drop_threshold(a.p_drop);
    const uint32_t rkey0 = drop_row_key(a.seed, (uint64_t)(gh * a.Lq + 4 * hh));    // + (query - 4 hh) * kDropWeyl: the rows of a tile are a Weyl step apart
    const uint32_t kj = (uint32_t)krow >> 1, kshift = ((uint32_t)krow & 1u) * 16;

    f32x16 dk[2], dv[2];
#pragma unroll
    for (int i = 0; i < 16; ++i) { dk[0][i] = 0.f; dk[1][i] = 0.f; dv[0][i] = 0.f; dv[1][i] = 0.f; }
    X8 qfa[4], dofa[4], qr[4], dor[4];
    frag_load<T>(qb + (int64_t)min(r, a.Lq - 1) * a.q_sr + 8 * hh, qfa);
    frag_load<T>(dob + (int64_t)min(r, a.Lq - 1) * a.o_sr + 8 * hh, dofa);
    tile_load<T>(qb, a.q_sr, 0, a.Lq, lane, qr);
    tile_load<T>(dob, a.o_sr, 0, a.Lq, lane, dor);
    float st = stat_load(0);
    for (int qt = 0; qt < a.nqt; ++qt) {
        const int q0 = qt * 32;
        f32x16 s, dp;
#pragma unroll
        for (int i = 0; i < 16; ++i) { s[i] = 0.f; dp[i] = 0.f; }
#pragma unroll
        for (int sx = 0; sx < 4; ++sx) { s = Elem<T>::mfma32(qfa[sx], kfb[sx], s); dp = Elem<T>::mfma32(dofa[sx], vfb[sx], dp); }   // rows = queries, column = my key
        tile_store<T>(ql, lane, qr);
        tile_store<T>(dl, lane, dor);
        stat[lane] = st;
        if (qt + 1 < a.nqt) {
            frag_load<T>(qb + (int64_t)min(q0 + 32 + r, a.Lq - 1) * a.q_sr + 8 * hh, qfa);
            frag_load<T>(dob + (int64_t)min(q0 + 32 + r, a.Lq - 1) * a.o_sr + 8 * hh, dofa);
            tile_load<T>(qb, a.q_sr, q0 + 32, a.Lq, lane, qr);
            tile_load<T>(dob, a.o_sr, q0 + 32, a.Lq, lane, dor);
            st = stat_load(q0 + 32);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        float4 nl[4], dsv[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {                            // accumulator rows 8 j + 4 hh + (0..3)
            nl[j] = *reinterpret_cast<const float4*>(stat + 8 * j + 4 * hh);
            dsv[j] = *reinterpret_cast<const float4*>(stat + 32 + 8 * j + 4 * hh);
        }
        const uint32_t rkq = rkey0 + (uint32_t)q0 * kDropWeyl;
        X8 pdf[2], dsf[2];
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const float nli = (i & 3) == 0 ? nl[i >> 2].x : (i & 3) == 1 ? nl[i >> 2].y : (i & 3) == 2 ? nl[i >> 2].z : nl[i >> 2].w;
            const float dsi = (i & 3) == 0 ? dsv[i >> 2].x : (i & 3) == 1 ? dsv[i >> 2].y : (i & 3) == 2 ? dsv[i >> 2].z : dsv[i >> 2].w;
            bool kept = true;
            if constexpr (DROP) kept = kept16(hash32((rkq + (uint32_t)((i & 3) + 8 * (i >> 2)) * kDropWeyl) ^ kj), kshift, thr);
            const Adj ad = adjoint<T, DROP>(s[i], dp[i], sl, MASKED ? maskv + nli : nli, dsi, keep, ikeep, kept);
            pdf[i >> 3][i & 7] = static_cast<T>(ad.pd);
            dsf[i >> 3][i & 7] = static_cast<T>(ad.ds);
        }
        tr_accumulate<T>(dv, dl, voff, pdf);                    // dV^T[dh][key] += dO^T[dh][q] Pd[q][key]
        tr_accumulate<T>(dk, ql, voff, dsf);                    // dK^T[dh][key] += Q^T[dh][q]  (dS / scale)[q][key]
        __builtin_amdgcn_wave_barrier();
    }
    if (kvalid) {
        store_grad<T>(dk, a.dk, g * a.dk_sg + h * a.dk_sh + (int64_t)key * a.dk_sr + 4 * hh, a.grad16, a.scale);
        store_grad<T>(dv, a.dv, g * a.dv_sg + h * a.dv_sh + (int64_t)key * a.dv_sr + 4 * hh, a.grad16, 1.0f);
    }
}

static int tattn_check(const TAttnArgs& a, int dtype) {
    if (a.G <= 0 || a.H <= 0 || a.Lq <= 0 || a.Lk <= 0) return CIR_EINVAL;
    if (dtype != CIR_BF16 && dtype != CIR_F16) return CIR_EDTYPE;
    if (a.p_drop < 0.f || a.p_drop >= 1.f) return CIR_EINVAL;
    const int64_t s16[] = {a.q_sg, a.q_sh, a.q_sr, a.k_sg, a.k_sh, a.k_sr, a.v_sg, a.v_sh, a.v_sr, a.o_sg, a.o_sh, a.o_sr};
    for (int64_t s : s16)
        if (s % 8) return CIR_EALIGN;
    if (!cir_aligned16(a.q) || !cir_aligned16(a.k) || !cir_aligned16(a.v) || !cir_aligned16(a.o)) return CIR_EALIGN;
    if ((int64_t)a.G * a.H * a.Lq > 0xffffffffLL) return CIR_ESHAPE;                 // dropout rows are numbered in 32 bits
    if ((int64_t)a.G * a.H * ((a.Lq + 31) / 32) > 0x7fffffffLL || (int64_t)a.G * a.H * ((a.Lk + 31) / 32) > 0x7fffffffLL) return CIR_ESHAPE;
    return CIR_OK;
}

}  // namespace cir

extern "C" int cir_attention_train_fwd(const void* q, int64_t q_sg, int64_t q_sh, int64_t q_sr, const void* k, int64_t k_sg, int64_t k_sh,
                                       int64_t k_sr, const void* v, int64_t v_sg, int64_t v_sh, int64_t v_sr, const float* mask, void* out,
                                       int64_t o_sg, int64_t o_sh, int64_t o_sr, float* out32, float* lse, int G, int H, int Lq, int Lk, float scale,
                                       float p_drop, uint64_t seed, int dtype, void* stream) {
    using namespace cir;
    CIR_CHECK_PTR(q); CIR_CHECK_PTR(k); CIR_CHECK_PTR(v); CIR_CHECK_PTR(out); CIR_CHECK_PTR(lse);
    if (out32 != nullptr && !cir_aligned16(out32)) return CIR_EALIGN;
    TAttnArgs a = {};
    a.q = q; a.q_sg = q_sg; a.q_sh = q_sh; a.q_sr = q_sr;
    a.k = k; a.k_sg = k_sg; a.k_sh = k_sh; a.k_sr = k_sr;
    a.v = v; a.v_sg = v_sg; a.v_sh = v_sh; a.v_sr = v_sr;
    a.o = out; a.d_o = nullptr; a.o_sg = o_sg; a.o_sh = o_sh; a.o_sr = o_sr; a.o32 = out32;
    a.mask = mask; a.lse = lse;
    a.G = G; a.H = H; a.Lq = Lq; a.Lk = Lk; a.nqt = (Lq + 31) / 32; a.nkt = (Lk + 31) / 32;
    a.scale = scale; a.p_drop = p_drop; a.seed = seed;
    if (const int e = tattn_check(a, dtype)) return e;
    const int64_t units = (int64_t)G * H * a.nqt;
    dim3 grid((unsigned)((units + 3) / 4)), block(256);
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const bool mk = mask != nullptr;
    if (dtype == CIR_BF16) { if (mk) hipLaunchKernelGGL((tattn_fwd_kernel<__bf16, true>), grid, block, 0, s, a); else hipLaunchKernelGGL((tattn_fwd_kernel<__bf16, false>), grid, block, 0, s, a); }
    else { if (mk) hipLaunchKernelGGL((tattn_fwd_kernel<_Float16, true>), grid, block, 0, s, a); else hipLaunchKernelGGL((tattn_fwd_kernel<_Float16, false>), grid, block, 0, s, a); }
    CIR_LAUNCH_RESULT();
}

extern "C" int cir_attention_train_bwd(const void* q, int64_t q_sg, int64_t q_sh, int64_t q_sr, const void* k, int64_t k_sg, int64_t k_sh,
                                       int64_t k_sr, const void* v, int64_t v_sg, int64_t v_sh, int64_t v_sr, const float* mask, const void* out,
                                       const void* d_out, int64_t o_sg, int64_t o_sh, int64_t o_sr, const float* out32, const float* lse,
                                       float* dsum_scratch,
                                       void* dq, int64_t dq_sg, int64_t dq_sh, int64_t dq_sr, void* dk, int64_t dk_sg, int64_t dk_sh,
                                       int64_t dk_sr, void* dv, int64_t dv_sg, int64_t dv_sh, int64_t dv_sr, int grad_dtype, int G, int H, int Lq,
                                       int Lk, float scale, float p_drop, uint64_t seed, int dtype, void* stream) {
    using namespace cir;
    CIR_CHECK_PTR(q); CIR_CHECK_PTR(k); CIR_CHECK_PTR(v); CIR_CHECK_PTR(out); CIR_CHECK_PTR(d_out); CIR_CHECK_PTR(lse);
    CIR_CHECK_PTR(dsum_scratch); CIR_CHECK_PTR(dq); CIR_CHECK_PTR(dk); CIR_CHECK_PTR(dv);
    TAttnArgs a = {};
    a.q = q; a.q_sg = q_sg; a.q_sh = q_sh; a.q_sr = q_sr;
    a.k = k; a.k_sg = k_sg; a.k_sh = k_sh; a.k_sr = k_sr;
    a.v = v; a.v_sg = v_sg; a.v_sh = v_sh; a.v_sr = v_sr;
    a.o = out; a.d_o = d_out; a.o_sg = o_sg; a.o_sh = o_sh; a.o_sr = o_sr;
    a.o32 = const_cast<float*>(out32);
    if (out32 != nullptr && !cir_aligned16(out32)) return CIR_EALIGN;
    a.mask = mask; a.lse = const_cast<float*>(lse); a.dsum = dsum_scratch;
    a.dq = dq; a.dq_sg = dq_sg; a.dq_sh = dq_sh; a.dq_sr = dq_sr;
    a.dk = dk; a.dk_sg = dk_sg; a.dk_sh = dk_sh; a.dk_sr = dk_sr;
    a.dv = dv; a.dv_sg = dv_sg; a.dv_sh = dv_sh; a.dv_sr = dv_sr;
    if (grad_dtype != CIR_F32 && grad_dtype != dtype) return CIR_EDTYPE;
    a.grad16 = grad_dtype != CIR_F32;
    a.G = G; a.H = H; a.Lq = Lq; a.Lk = Lk; a.nqt = (Lq + 31) / 32; a.nkt = (Lk + 31) / 32;
    a.scale = scale; a.p_drop = p_drop; a.seed = seed;
    if (const int e = tattn_check(a, dtype)) return e;
    if (!cir_aligned16(d_out) || !cir_aligned16(dq) || !cir_aligned16(dk) || !cir_aligned16(dv)) return CIR_EALIGN;
    const int64_t s32[] = {dq_sg, dq_sh, dq_sr, dk_sg, dk_sh, dk_sr, dv_sg, dv_sh, dv_sr};
    for (int64_t st : s32)
        if (st % 4) return CIR_EALIGN;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const bool mk = mask != nullptr;
    dim3 block(256);
    dim3 gq((unsigned)(((int64_t)G * H * a.nqt + 3) / 4)), gk((unsigned)(((int64_t)G * H * a.nkt + 3) / 4));
    // dQ first: it also writes D = rowsum(dO * O), which the dK / dV kernel reads (same stream: ordered)
    const bool dr = p_drop > 0.f;
#define CIR_TATTN_BWD(TT, MK, DR) do { hipLaunchKernelGGL((tattn_bwd_dq_kernel<TT, MK, DR>), gq, block, 0, s, a); \
                                       hipLaunchKernelGGL((tattn_bwd_dkv_kernel<TT, MK, DR>), gk, block, 0, s, a); } while (0)
#define CIR_TATTN_BWD_T(TT) do { if (mk) { if (dr) CIR_TATTN_BWD(TT, true, true); else CIR_TATTN_BWD(TT, true, false); } \
                                 else { if (dr) CIR_TATTN_BWD(TT, false, true); else CIR_TATTN_BWD(TT, false, false); } } while (0)
    if (dtype == CIR_BF16) CIR_TATTN_BWD_T(__bf16); else CIR_TATTN_BWD_T(_Float16);
#undef CIR_TATTN_BWD_T
#undef CIR_TATTN_BWD
    CIR_LAUNCH_RESULT();
}
