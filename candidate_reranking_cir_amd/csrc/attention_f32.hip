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

template <bool MASKED>
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
    if (a.mask) hipLaunchKernelGGL((attn_f32_kernel<true>), grid, block, 0, s, a);
    else hipLaunchKernelGGL((attn_f32_kernel<false>), grid, block, 0, s, a);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? CIR_OK : (int)e;
}

}  // namespace cir
