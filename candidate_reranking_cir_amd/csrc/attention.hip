// cir_attention: softmax(q k^T * scale + mask) v for head_dim 64, one wave per (item, head, 32 queries).
//
// Bound: MFMA in principle (4*Lq*Lk*64 flop per head), VALU (exp/rescale) in practice at these tiny
// extents; the path spends < 5 % of its flops here.
//
// Formulation ("keys on rows"): with the 32x32x16 MFMA the score tile is computed TRANSPOSED,
//   S^T[key][query] = K_tile * Q^T,
// so a lane owns one query column and 16 of the tile's 32 keys; the row max / sum of the online
// softmax is an in-register reduction plus ONE cross-lane exchange (lane ^ 32).  P^T then already
// has the register layout of the B operand of the second product
//   O^T[dh][query] += V_tile^T * P^T,
// (accumulator-as-operand, no LDS round trip for P).  V^T fragments come from a row-major LDS copy
// of the V tile through ds_read_b64_tr_b16 (hardware transpose).  Q and K fragments are 16-byte
// vectors straight from global memory (a head's row slice is one 128-byte line).
// Ragged extents: rows beyond Lq / Lk are clamped on load; scores of keys >= Lk are set to -inf.
#include "common.hpp"

namespace cir {

struct AttnArgs {
    const void* q; int64_t q_s1, q_s0, q_rs;
    const void* k; int64_t k_s1, k_s0, k_rs;
    const void* v; int64_t v_s1, v_s0, v_rs;
    const float* mask; int64_t m_s1, m_s0;
    void* out; int64_t o_s1, o_s0, o_rs;
    int B0, H, Lq, Lk, nqt;
    int64_t total;
    float scale;
};

typedef __attribute__((address_space(3))) s16x4* lds_s16x4_ptr;

template <typename T>
__global__ __launch_bounds__(256) void attn_kernel(const AttnArgs a) {
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
    const T* kb = reinterpret_cast<const T*>(a.k) + b1 * a.k_s1 + b0 * a.k_s0 + h * 64;
    const T* vb = reinterpret_cast<const T*>(a.v) + b1 * a.v_s1 + b0 * a.v_s0 + h * 64;
    const float* mp = a.mask ? a.mask + b1 * a.m_s1 + b0 * a.m_s0 : nullptr;

    X8 qf[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) qf[s] = *reinterpret_cast<const X8*>(qp + 16 * s);

    constexpr float kLog2e = 1.4426950408889634f;
    float m_run = -INFINITY, l_run = 0.f;
    f32x16 o[2];
#pragma unroll
    for (int i = 0; i < 16; ++i) { o[0][i] = 0.f; o[1][i] = 0.f; }

    char* vl = smem + wave * 4096;
    const int i16 = lane & 15;
    // transposed-read address of this lane inside a 4-key x 16-dh block: row (i16>>2), 4 columns at (i16&3)*4
    const int tr_lane_off = (4 * hh + (i16 >> 2)) * 128 + (16 * ((lane >> 4) & 1) + 4 * (i16 & 3)) * 2;

    const int nkt = (a.Lk + 31) >> 5;
    for (int kt = 0; kt < nkt; ++kt) {
        const int key0 = kt * 32;
        const int krow = min(key0 + r, a.Lk - 1);
        const T* kp = kb + (int64_t)krow * a.k_rs + 8 * hh;
        X8 kf[4];
#pragma unroll
        for (int s = 0; s < 4; ++s) kf[s] = *reinterpret_cast<const X8*>(kp + 16 * s);
        X8 vreg[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int c = lane + 64 * i;
            const int vrow = min(key0 + (c >> 3), a.Lk - 1);  // clamped rows get probability 0
            vreg[i] = *reinterpret_cast<const X8*>(vb + (int64_t)vrow * a.v_rs + (c & 7) * 8);
        }

        f32x16 s;
#pragma unroll
        for (int i = 0; i < 16; ++i) s[i] = 0.f;
#pragma unroll
        for (int st = 0; st < 4; ++st) s = Elem<T>::mfma32(kf[st], qf[st], s);

        float sv[16];
        float mx = -INFINITY;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int key = key0 + (i & 3) + 8 * (i >> 2) + 4 * hh;
            float val = s[i] * a.scale;
            if (mp) val += mp[min(key, a.Lk - 1)];
            val = key < a.Lk ? val : -INFINITY;
            sv[i] = val;
            mx = fmaxf(mx, val);
        }
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        const float m_new = fmaxf(m_run, mx);  // finite: every tile holds at least one valid key
        const float alpha = __builtin_amdgcn_exp2f((m_run - m_new) * kLog2e);
        float psum = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const float p = __builtin_amdgcn_exp2f((sv[i] - m_new) * kLog2e);
            sv[i] = p;
            psum += p;
        }
        psum += __shfl_xor(psum, 32, 64);
        l_run = l_run * alpha + psum;
        m_run = m_new;
#pragma unroll
        for (int i = 0; i < 16; ++i) { o[0][i] *= alpha; o[1][i] *= alpha; }

        X8 pf[2];
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
            for (int j = 0; j < 8; ++j) pf[s2][j] = static_cast<T>(sv[8 * s2 + j]);

        // stage V row-major in this wave's LDS tile (lane-linear 16-byte writes)
#pragma unroll
        for (int i = 0; i < 4; ++i) *reinterpret_cast<X8*>(vl + (lane + 64 * i) * 16) = vreg[i];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();

#pragma unroll
        for (int dt = 0; dt < 2; ++dt) {
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                const char* base = vl + tr_lane_off + (16 * s2) * 128 + dt * 64;
                const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(base));
                const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(base + 8 * 128));
                s16x8 both;
                both.s0 = lo.x; both.s1 = lo.y; both.s2 = lo.z; both.s3 = lo.w;
                both.s4 = hi.x; both.s5 = hi.y; both.s6 = hi.z; both.s7 = hi.w;
                o[dt] = Elem<T>::mfma32(__builtin_bit_cast(X8, both), pf[s2], o[dt]);
            }
        }
        __builtin_amdgcn_wave_barrier();  // keep the next tile's LDS writes behind these reads
    }

    if (q0 + r < a.Lq) {
        const float inv = 1.0f / l_run;
        T* op = reinterpret_cast<T*>(a.out) + b1 * a.o_s1 + b0 * a.o_s0 + (int64_t)(q0 + r) * a.o_rs + h * 64 + 4 * hh;
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int qd = 0; qd < 4; ++qd) {
                u32x2 p;
                p.x = pack2<T>(o[dt][qd * 4 + 0] * inv, o[dt][qd * 4 + 1] * inv);
                p.y = pack2<T>(o[dt][qd * 4 + 2] * inv, o[dt][qd * 4 + 3] * inv);
                *reinterpret_cast<u32x2*>(op + dt * 32 + 8 * qd) = p;
            }
    }
}

}  // namespace cir

extern "C" int cir_attention(const void* q, int64_t q_s1, int64_t q_s0, int64_t q_rs, const void* k, int64_t k_s1,
                             int64_t k_s0, int64_t k_rs, const void* v, int64_t v_s1, int64_t v_s0, int64_t v_rs,
                             const float* mask, int64_t m_s1, int64_t m_s0, void* out, int64_t o_s1, int64_t o_s0,
                             int64_t o_rs, int B1, int B0, int H, int Lq, int Lk, float scale, int dtype, void* stream) {
    using namespace cir;
    CIR_CHECK_PTR(q); CIR_CHECK_PTR(k); CIR_CHECK_PTR(v); CIR_CHECK_PTR(out);
    if (B1 <= 0 || B0 <= 0 || H <= 0 || Lq <= 0 || Lk <= 0) return CIR_EINVAL;
    if (dtype != CIR_BF16 && dtype != CIR_F16) return CIR_EDTYPE;
    const int64_t strides[] = {q_s1, q_s0, q_rs, k_s1, k_s0, k_rs, v_s1, v_s0, v_rs};
    for (int64_t s : strides)
        if (s % 8) return CIR_EALIGN;
    if (o_s1 % 4 || o_s0 % 4 || o_rs % 4) return CIR_EALIGN;
    if (!cir_aligned16(q) || !cir_aligned16(k) || !cir_aligned16(v) || (reinterpret_cast<uintptr_t>(out) & 7)) return CIR_EALIGN;
    AttnArgs a;
    a.q = q; a.q_s1 = q_s1; a.q_s0 = q_s0; a.q_rs = q_rs;
    a.k = k; a.k_s1 = k_s1; a.k_s0 = k_s0; a.k_rs = k_rs;
    a.v = v; a.v_s1 = v_s1; a.v_s0 = v_s0; a.v_rs = v_rs;
    a.mask = mask; a.m_s1 = m_s1; a.m_s0 = m_s0;
    a.out = out; a.o_s1 = o_s1; a.o_s0 = o_s0; a.o_rs = o_rs;
    a.B0 = B0; a.H = H; a.Lq = Lq; a.Lk = Lk; a.nqt = (Lq + 31) / 32;
    a.total = (int64_t)B1 * B0 * H * a.nqt;
    a.scale = scale;
    const int64_t nblk = (a.total + 3) / 4;
    if (nblk > 0x7fffffff) return CIR_ESHAPE;
    dim3 grid((unsigned)nblk), block(256);
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (dtype == CIR_BF16) hipLaunchKernelGGL((attn_kernel<__bf16>), grid, block, 0, s, a);
    else hipLaunchKernelGGL((attn_kernel<_Float16>), grid, block, 0, s, a);
    CIR_LAUNCH_RESULT();
}
