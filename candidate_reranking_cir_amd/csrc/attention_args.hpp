// Argument block of cir_attention's kernels (attention.hip: 16-bit operands; attention_f32.hip: fp32 operands).
#pragma once
#include "common.hpp"

namespace cir {

struct AttnArgs {
    const void* q; int64_t q_s1, q_s0, q_rs;
    const void* k; int64_t k_s1, k_s0, k_rs;
    const void* v; int64_t v_s1, v_s0, v_rs;
    const float* mask; int64_t m_s1, m_s0;
    const int64_t* kv_index;   // optional: item b1 reads K/V of bank row kv_index[b1] (cross-query K/V cache)
    void* out; int64_t o_s1, o_s0, o_rs;
    int B0, H, Lq, Lk, nqt;
    int wide_store;            // attn_shared_kernel: one tile per wave, 16-byte-aligned output rows and room in LDS -> whole-row stores through LDS
    int64_t total;
    float scale;
    int out_split = 0;         // attn_f32_kernel: `out` receives "split8" rows (common.hpp; o_s1 / o_s0 / o_rs in BYTES) instead of fp32
};

constexpr float kLog2e = 1.4426950408889634f;

// one wave per (item, head, 32 queries) on v_mfma_f32_32x32x2_f32; returns a CIR_* / hipError_t code
int launch_attention_f32(const AttnArgs& a, hipStream_t s);

}  // namespace cir
