// cir_attention with fp32 tensors - the "exact" precision mode (round 5): softmax(q k^T * scale + mask) v with fp32 operands in
// BOTH products, fp32 statistics, fp32 out; head dimension 64.  Bound: fp32 MFMA (v_mfma_f32_32x32x2_f32: 64 flop / clk / SIMD,
// 157.3 TFLOP/s peak - f32 in, f32 accumulate, a chain of IEEE fmaf's); 4 * Lq * Lk * 64 flop per (item, head).
//
// One wave per (item, head, 32 queries), nothing shared between waves, no LDS: at 1/16 of the 16-bit MFMA rate a 32-key tile
// is 64 MFMAs = 4096 matrix-pipe cycles, against 8 + 32 load instructions - the operand stream is not the problem here.
// Same formulation as the 16-bit kernels (attention.hip), with the fp32 MFMA's one-value-per-lane operands:
//   S^T[key][query] = K_tile Q^T : lane (r, hh) holds K[key0 + r][8c + 4hh + s] and Q[q0 + r][8c + 4hh + s] (16-byte loads, c < 8, s < 4);
//                                  MFMA (c, s) contracts d in {8c + s, 8c + 4 + s} - 32 MFMAs cover d < 64;
//   the score tile's register i is key  key0 + (i & 3) + 8 (i >> 2) + 4 hh  of query r: online softmax in registers + one lane^32 exchange;
//   O^T[d][query] += V_tile^T P^T : MFMA i takes P^T's register i as the B operand (accumulator-as-operand: no LDS trip for P) and
//                                  V[key(i, hh)][32 dt + r] as A - one 4-byte load per lane, 32 consecutive d per half-wave = a whole 128-byte line.
// The next tile's K fragments are requested behind the score product, its V behind the second product.
// Replaces BertSelfAttention.forward (nlvr_encoder.py:140-222, med.py:158-240) and Attention.forward's core (vit.py:73-83) when every
// tensor of the model is fp32, like the reference's (validate_stage2.py:140-141).

#include "attention_args.hpp"

namespace cir {

// SPLIT (round 6): the context rows leave as "split8" operand rows [D x fp16 | D x e4m3 lo | D x e4m3 hi] (common.hpp) - what the
// self-attention output projection of the text32 mode reads - instead of fp32: the two half-waves exchange 4-element groups
// (v_permlane32_swap) so that a lane holds 8 consecutive features and stores 16 + 8 + 8 bytes per group.
template <bool MASKED, bool SPLIT = false>
__global__ __launch_bounds__(256, 2) void attn_f32_kernel(const AttnArgs a) {
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
    const float* qp = reinterpret_cast<const float*>(a.q) + b1 * a.q_s1 + b0 * a.q_s0 + (int64_t)qrow * a.q_rs + h * 64 + 4 * hh;
    const int64_t kb1 = a.kv_index ? a.kv_index[b1] : b1;
    const float* kb = reinterpret_cast<const float*>(a.k) + kb1 * a.k_s1 + b0 * a.k_s0 + h * 64 + 4 * hh;
    const float* vb = reinterpret_cast<const float*>(a.v) + kb1 * a.v_s1 + b0 * a.v_s0 + h * 64 + r;
    const float* mp = MASKED ? a.mask + b1 * a.m_s1 + b0 * a.m_s0 : nullptr;

    f32x4 qf[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) qf[c] = *reinterpret_cast<const f32x4*>(qp + 8 * c);

    const int nkt = (a.Lk + 31) >> 5;
    auto load_k = [&](int kt, f32x4 (&kf)[8]) {
        const float* kp = kb + (int64_t)min(kt * 32 + r, a.Lk - 1) * a.k_rs;
#pragma unroll
        for (int c = 0; c < 8; ++c) kf[c] = *reinterpret_cast<const f32x4*>(kp + 8 * c);
    };
    auto load_v = [&](int kt, float (&vf)[2][16]) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int key = min(kt * 32 + (i & 3) + 8 * (i >> 2) + 4 * hh, a.Lk - 1);     // rows past Lk: probability 0, any finite value
            const float* vp = vb + (int64_t)key * a.v_rs;
            vf[0][i] = vp[0];
            vf[1][i] = vp[32];
        }
    };

    float m_run = -INFINITY, l_run = 0.f;
    f32x16 o[2];
#pragma unroll
    for (int i = 0; i < 16; ++i) { o[0][i] = 0.f; o[1][i] = 0.f; }
    const float sl = a.scale * kLog2e;

    f32x4 kf[8];
    float vf[2][16];
    load_k(0, kf);
    load_v(0, vf);
    for (int kt = 0; kt < nkt; ++kt) {
        const int key0 = kt * 32;
        f32x16 s;
#pragma unroll
        for (int i = 0; i < 16; ++i) s[i] = 0.f;
#pragma unroll
        for (int c = 0; c < 8; ++c)
#pragma unroll
            for (int e = 0; e < 4; ++e) s = __builtin_amdgcn_mfma_f32_32x32x2f32(kf[c][e], qf[c][e], s, 0, 0, 0);
        if (kt + 1 < nkt) load_k(kt + 1, kf);
        // ---- online softmax in the log2 domain: lane = query r, 16 of the tile's 32 keys ----
        float sv[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int key = key0 + (i & 3) + 8 * (i >> 2) + 4 * hh;
            float x = s[i] * sl;
            if constexpr (MASKED) x = fmaf(fmaxf(mp[min(key, a.Lk - 1)], -2.0e38f), kLog2e, x);    // finfo.min-style masks stay finite
            sv[i] = key < a.Lk ? x : -INFINITY;
        }
        float mx = sv[0];
#pragma unroll
        for (int i = 1; i < 16; ++i) mx = fmaxf(mx, sv[i]);
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        const float m_new = fmaxf(m_run, mx);             // finite: every tile holds at least one valid key
        const float alpha = exp2f(m_run - m_new);
        m_run = m_new;
        float psum = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            sv[i] = exp2f(sv[i] - m_new);
            psum += sv[i];
        }
        psum += __shfl_xor(psum, 32, 64);
        l_run = l_run * alpha + psum;
#pragma unroll
        for (int i = 0; i < 16; ++i) { o[0][i] *= alpha; o[1][i] *= alpha; }
        // ---- O^T += V_tile^T P^T ----
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            o[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(vf[0][i], sv[i], o[0], 0, 0, 0);
            o[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(vf[1][i], sv[i], o[1], 0, 0, 0);
        }
        if (kt + 1 < nkt) load_v(kt + 1, vf);
    }
    if constexpr (SPLIT) {
        const float inv = 1.0f / l_run;
        const int d_model = a.H * 64;
        char* row = reinterpret_cast<char*>(a.out) + b1 * a.o_s1 + b0 * a.o_s0 + (int64_t)min(q0 + r, a.Lq - 1) * a.o_rs;
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int p = 0; p < 2; ++p) {
                float lo4[4], hi4[4];          // own groups qd = 2p (features 8 qd + 4 hh + j) and qd = 2p + 1
#pragma unroll
                for (int j = 0; j < 4; ++j) { lo4[j] = o[dt][(2 * p) * 4 + j] * inv; hi4[j] = o[dt][(2 * p + 1) * 4 + j] * inv; }
                // swap: lanes 0-31 end with (own group 2p | partner's group 2p) = 8 consecutive features at 8 (2p); lanes 32-63 with
                // (partner's group 2p+1 | own group 2p+1) = 8 consecutive features at 8 (2p+1)
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
    } else
    if (q0 + r < a.Lq) {
        const float inv = 1.0f / l_run;
        float* op = reinterpret_cast<float*>(a.out) + b1 * a.o_s1 + b0 * a.o_s0 + (int64_t)(q0 + r) * a.o_rs + h * 64 + 4 * hh;
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int qd = 0; qd < 4; ++qd)
                *reinterpret_cast<float4*>(op + dt * 32 + 8 * qd) =
                    make_float4(o[dt][qd * 4 + 0] * inv, o[dt][qd * 4 + 1] * inv, o[dt][qd * 4 + 2] * inv, o[dt][qd * 4 + 3] * inv);
    }
}

int launch_attention_f32(const AttnArgs& a, hipStream_t s) {
    const int64_t nblk = (a.total + 3) / 4;
    if (nblk > 0x7fffffff) return CIR_ESHAPE;
    dim3 grid((unsigned)nblk), block(256);
    if (a.out_split) {
        if (a.mask) hipLaunchKernelGGL((attn_f32_kernel<true, true>), grid, block, 0, s, a);
        else hipLaunchKernelGGL((attn_f32_kernel<false, true>), grid, block, 0, s, a);
    } else if (a.mask) hipLaunchKernelGGL((attn_f32_kernel<true>), grid, block, 0, s, a);
    else hipLaunchKernelGGL((attn_f32_kernel<false>), grid, block, 0, s, a);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? CIR_OK : (int)e;
}

}  // namespace cir
