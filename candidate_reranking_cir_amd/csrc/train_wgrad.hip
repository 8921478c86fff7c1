// cir_wgrad: dW (N, K) += dy^T x - the weight gradient of a dense layer (nn.Linear's adjoint behind
// nlvr_encoder.py:150-168, 250-264, 383-409 in the training step) - on MFMA with both operands read AS STORED:
// dy (rows, N) and x (rows, K) are row-major 16-bit, the contraction runs over the ROWS, i.e. over the slow dimension of both.
//
// Bound: MFMA.  Algorithmic work 2 * rows * N * K flop.  Round 3 ran this product on cir_bmm (trans_a): operands through registers into
// LDS, 32 rows per barrier, 360 TFLOP/s on the step's shapes - 18 % of the step.  This kernel is the 128 x 128 GEMM's skeleton
// (gemm.hip: LDS-DMA staging, two LDS buffers, 64 contraction rows per barrier, XCD-aware tile order) turned for the transposed operands:
//   * a stage is 64 rows x 128 columns of each operand (a 256-byte row piece = 16 chunks of 16 bytes); one wave-instruction of
//     global_load_lds moves 4 such row pieces (1 KiB);
//   * an MFMA fragment needs 8 consecutive CONTRACTION indices of one output row - here 8 different LDS rows.  ds_read_b64_tr_b16 does that
//     transposition on the way out of LDS: a 16-lane group reads a 4-row x 16-column block and lane i receives column i's 4 values; two
//     of them (rows +0..3, +4..7) make one 16x16x32 operand.  The 16 pieces (4 rows x 32 bytes, for the 4 lane groups) of one such read
//     sit 256 bytes apart and would all start in the same bank: LDS row m therefore holds global chunk s ^ f(m),
//     f(m) = 2 (m & 3) + 8 ((m >> 3) & 1) (applied on the DMA's per-lane SOURCE address, the LDS image itself is lane-linear), which
//     spreads the 8 pieces of each half-wave over the 64 banks and keeps a 32-byte column pair adjacent;
//   * the rows are split over `splits` workgroups per output tile (a 768 x 768 weight has 36 tiles); every workgroup adds its partial tile
//     into dW with fp32 atomics - no partial tensor, no reduction launch.  Summation order is not fixed.
// Requirements: N % 128 == 0, K % 128 == 0, 16-byte aligned rows; rows beyond a multiple of 64 are left to the caller (cir_wgrad runs
// them through cir_bmm's kernel).
#include "gemm_args.hpp"

namespace cir {

struct WgradProblem {
    const void* dy; int64_t ldy;
    const void* x; int64_t ldx;
    float* dw; int64_t ldw;
    int tiles_k, splits, steps_per_split, total_steps;      // 64-row steps per split (the last split may run short)
    int unit_begin;                                         // first workgroup unit of this problem: units = tiles_n * tiles_k * splits
};
constexpr int kWgMaxProblems = 16;
struct WgradArgs {
    WgradProblem p[kWgMaxProblems];
    int count, units;
};

constexpr int kWgTile = 64 * 128 * 2;      // 16 KiB: 64 contraction rows x 128 columns of one operand
typedef __attribute__((address_space(3))) s16x4* wg_lds_s16x4_ptr;

template <typename T>
__global__ __launch_bounds__(256, 2) void wgrad_kernel(const WgradArgs a) {
    using X8 = typename Elem<T>::x8;
    __shared__ __attribute__((aligned(16))) char smem[4 * kWgTile];     // [buf][dy | x]
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave & 1, wn = wave >> 1;
    const int r15 = lane & 15, g = lane >> 4;

    // ---- which (problem, tile, split).  Problems are laid out longest units first (host).
    int id = xcd_remap(blockIdx.x, gridDim.x);
    int pi = 0;
#pragma unroll 1
    for (int i = 1; i < a.count; ++i) pi = id >= a.p[i].unit_begin ? i : pi;
    const WgradProblem& pr = a.p[pi];
    id -= pr.unit_begin;
    // Unit order inside a problem: splits slowest, and of the two tile extents the SMALLER runs fastest - the ~64 workgroups resident on
    // an XCD (each XCD gets a contiguous range of the remapped ids) then cover a near-square block of tiles of ONE split, so every dy / x row
    // piece they stream is shared by 6-10 of them in that L2.  Measured on one BertLayer's 13 products against split-fastest / K-extent-
    // fastest: 504 against 539 us (PMC of the latter: 1.82 GB fetched per launch for 0.63 GB of operands).
    const int tiles_pr = (int)((pi + 1 < a.count ? a.p[pi + 1].unit_begin : a.units) - pr.unit_begin) / pr.splits;
    const int split = id / tiles_pr;
    id -= split * tiles_pr;
    const int tiles_n_pr = tiles_pr / pr.tiles_k;
    int tile_n, tile_k;
    if (pr.tiles_k <= tiles_n_pr) { tile_n = id / pr.tiles_k; tile_k = id - tile_n * pr.tiles_k; }
    else { tile_k = id / tiles_n_pr; tile_n = id - tile_k * tiles_n_pr; }
    const int n0 = tile_n * 128, k0 = tile_k * 128;
    const int64_t step0 = (int64_t)split * pr.steps_per_split;
    const int nsteps = (int)(pr.total_steps - step0 < pr.steps_per_split ? pr.total_steps - step0 : pr.steps_per_split);
    if (nsteps <= 0) return;
    const int64_t ldy = pr.ldy, ldx = pr.ldx, ldw = pr.ldw;
    const bool atomic = pr.splits > 1;

    // ---- per-lane staging sources: 4 wave-instructions of 4 rows for each operand --------------------------------------------------
    const int srow = lane >> 4;                                   // row inside the 4-row piece
    const T* y_src[4];
    const T* x_src[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int m = (wave * 4 + j) * 4 + srow;                  // LDS row 0..63
        const int chunk = (lane & 15) ^ (((m & 3) << 1) | (((m >> 3) & 1) << 3));
        const int64_t gr = step0 * 64 + m;
        y_src[j] = reinterpret_cast<const T*>(pr.dy) + gr * ldy + n0 + chunk * 8;
        x_src[j] = reinterpret_cast<const T*>(pr.x) + gr * ldx + k0 + chunk * 8;
    }
    const int64_t ystep = 64 * ldy, xstep = 64 * ldx;
    auto stage = [&](int st, int buf) {
        char* base = smem + buf * 2 * kWgTile + wave * 4096;
#pragma unroll
        for (int j = 0; j < 4; ++j) __builtin_amdgcn_global_load_lds((gptr_t)(y_src[j] + st * ystep), (lptr_t)(base + j * 1024), 16, 0, 0);
#pragma unroll
        for (int j = 0; j < 4; ++j) __builtin_amdgcn_global_load_lds((gptr_t)(x_src[j] + st * xstep), (lptr_t)(base + kWgTile + j * 1024), 16, 0, 0);
    };

    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // fragment read offsets: this lane's 8 bytes of the 4-row x 16-column block of output-row block i (rows 8 g + q of a 32-row slab)
    const int q = r15 >> 2, p = r15 & 3;
    const int f = (q << 1) | ((g & 1) << 3);
    const int row_off = (8 * g + q) * 256 + (p & 1) * 8;
    int y_off[4], x_off[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        y_off[i] = row_off + (((wm * 8 + 2 * i + (p >> 1)) ^ f) << 4);
        x_off[i] = row_off + (((wn * 8 + 2 * i + (p >> 1)) ^ f) << 4);
    }
    auto frag = [&](const char* tile, int off) -> X8 {
        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((wg_lds_s16x4_ptr)(tile + off));
        const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((wg_lds_s16x4_ptr)(tile + off + 4 * 256));
        s16x8 both;
        both.s0 = lo.x; both.s1 = lo.y; both.s2 = lo.z; both.s3 = lo.w;
        both.s4 = hi.x; both.s5 = hi.y; both.s6 = hi.z; both.s7 = hi.w;
        return __builtin_bit_cast(X8, both);
    };

    stage(0, 0);
    for (int st = 0; st < nsteps; ++st) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // the DMA pieces of stage st (explicit: see gemm.hip)
        __syncthreads();                                          // ... landed everywhere; buffer (st+1)&1 is free again
        if (st + 1 < nsteps) stage(st + 1, (st + 1) & 1);
        const char* Ys = smem + (st & 1) * 2 * kWgTile;
        const char* Xs = Ys + kWgTile;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            X8 yf[4], xf[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                yf[i] = frag(Ys + ks * 32 * 256, y_off[i]);
                xf[i] = frag(Xs + ks * 32 * 256, x_off[i]);
            }
#pragma unroll
            for (int mi = 0; mi < 4; ++mi)
#pragma unroll
                for (int ni = 0; ni < 4; ++ni) acc[mi][ni] = Elem<T>::mfma16(yf[mi], xf[ni], acc[mi][ni]);
        }
    }
    // lane (r15, g) holds dW[n = 4 g + jj][k = r15] of each 16 x 16 tile: 16 lanes = 64 contiguous bytes per instruction.  A tile owned by
    // ONE workgroup (splits = 1) is added with a plain read-modify-write; split tiles with atomics (measured ~1.5 TB/s of atomic traffic:
    // every extra split of a 3072 x 768 weight costs 6 us - which is why cir_wgrad_grouped exists)
    float* dwp = pr.dw;
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) {
            float* cp = dwp + (int64_t)(n0 + wm * 64 + mi * 16 + g * 4) * ldw + k0 + wn * 64 + ni * 16 + r15;
            if (atomic) {
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) atomicAdd(cp + jj * ldw, acc[mi][ni][jj]);
            } else {
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) cp[jj * ldw] += acc[mi][ni][jj];
            }
        }
}

}  // namespace cir

extern "C" int cir_wgrad_grouped(const cir_wgrad_desc* d, int count, int in_dtype, void* stream) {
    using namespace cir;
    CIR_CHECK_PTR(d);
    if (count <= 0) return CIR_EINVAL;
    if (count > kWgMaxProblems) return CIR_ESHAPE;
    if (in_dtype != CIR_BF16 && in_dtype != CIR_F16) return CIR_EDTYPE;
    int64_t min_steps = 0, tiles_total = 0;
    for (int i = 0; i < count; ++i) {
        CIR_CHECK_PTR(d[i].dy); CIR_CHECK_PTR(d[i].x); CIR_CHECK_PTR(d[i].dw);
        if (d[i].rows <= 0 || d[i].N <= 0 || d[i].K <= 0 || d[i].splits < 0) return CIR_EINVAL;
        if (d[i].N % 128 != 0 || d[i].K % 128 != 0 || d[i].rows > 0x7fffffffLL) return CIR_ESHAPE;
        if (!cir_aligned16(d[i].dy) || !cir_aligned16(d[i].x) || d[i].ldy % 8 || d[i].ldx % 8 || (reinterpret_cast<uintptr_t>(d[i].dw) & 3u)) return CIR_EALIGN;
        const int64_t steps = d[i].rows / 64;
        if (steps > 0 && (min_steps == 0 || steps < min_steps)) min_steps = steps;
        if (steps > 0) tiles_total += (int64_t)(d[i].N / 128) * (d[i].K / 128);
    }
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (tiles_total > 0) {
        // Units of about equal length: a problem with k times the rows of the shortest one is split k ways (the FFN's 2R rows against
        // the attention projections' R); everything is split further until the group has ~9 units per CU (see `extra`).
        // Longest units first: the tail of the launch is then made of short ones.
        int order[kWgMaxProblems], n_act = 0;
        for (int i = 0; i < count; ++i) if (d[i].rows / 64 > 0) order[n_act++] = i;
        int64_t units_at_unit_len = 0;
        for (int j = 0; j < n_act; ++j) {
            const cir_wgrad_desc& q = d[order[j]];
            units_at_unit_len += (int64_t)(q.N / 128) * (q.K / 128) * ((q.rows / 64 + min_steps - 1) / min_steps);
        }
        // further split factor: about 2300 units per launch.  Measured on one BertLayer's 13 products (1224 units of 128 steps unsplit):
        // every unit halved (2448 units of 64 steps, every tile added by two workgroups' atomics) 546 us against 634 unsplit and 580 in
        // thirds - the tail of a 2.4-round launch costs more than the second adder's atomic traffic
        // (a launch that does not fill the chip on its own - a single product - keeps the smaller target: there the atomic traffic of
        // every extra split outweighs the balance, tools/bmm_bench.py: 36 tiles are best at 8 splits, 108 at 4)
        const int64_t target = units_at_unit_len >= 512 ? 2304 : 768;
        int64_t extra = (target + units_at_unit_len - 1) / units_at_unit_len;
        if (extra > min_steps / 4) extra = min_steps / 4;
        if (extra < 1) extra = 1;
        const int64_t unit_len = (min_steps + extra - 1) / extra;
        WgradArgs a;
        int64_t len[kWgMaxProblems];
        for (int j = 0; j < n_act; ++j) {
            const cir_wgrad_desc& q = d[order[j]];
            WgradProblem& p = a.p[j];
            const int64_t steps = q.rows / 64;
            int64_t sp = q.splits > 0 ? q.splits : (steps + unit_len / 2) / unit_len;     // nearest: 144 steps against 64-step units are 2 x 72, not 3 x 48
            if (sp < 1) sp = 1;
            if (sp > steps) sp = steps;
            p.dy = q.dy; p.ldy = q.ldy; p.x = q.x; p.ldx = q.ldx; p.dw = q.dw; p.ldw = q.ldw;
            p.tiles_k = q.K / 128; p.total_steps = (int)steps;
            p.steps_per_split = (int)((steps + sp - 1) / sp);
            p.splits = (int)((steps + p.steps_per_split - 1) / p.steps_per_split);
            len[j] = p.steps_per_split;
        }
        for (int j = 1; j < n_act; ++j)                                         // insertion sort by unit length, descending
            for (int k2 = j; k2 > 0 && len[k2] > len[k2 - 1]; --k2) {
                const WgradProblem tp = a.p[k2]; a.p[k2] = a.p[k2 - 1]; a.p[k2 - 1] = tp;
                const int64_t tl = len[k2]; len[k2] = len[k2 - 1]; len[k2 - 1] = tl;
                const int to = order[k2]; order[k2] = order[k2 - 1]; order[k2 - 1] = to;
            }
        int64_t units = 0;
        for (int j = 0; j < n_act; ++j) {
            a.p[j].unit_begin = (int)units;
            const cir_wgrad_desc& q = d[order[j]];
            units += (int64_t)(q.N / 128) * a.p[j].tiles_k * a.p[j].splits;
            if (units > 0x7fffffff) return CIR_ESHAPE;
        }
        a.count = n_act; a.units = (int)units;
        if (in_dtype == CIR_BF16) hipLaunchKernelGGL((wgrad_kernel<__bf16>), dim3((unsigned)units), dim3(256), 0, s, a);
        else hipLaunchKernelGGL((wgrad_kernel<_Float16>), dim3((unsigned)units), dim3(256), 0, s, a);
        const hipError_t e = hipGetLastError();
        if (e != hipSuccess) return (int)e;
    }
    for (int i = 0; i < count; ++i) {                             // ragged tails (< 64 rows): the general kernel, atomically into the same dW
        const int64_t main_rows = d[i].rows / 64 * 64;
        if (main_rows == d[i].rows) continue;
        const int e = cir_bmm(reinterpret_cast<const char*>(d[i].dy) + main_rows * d[i].ldy * 2, reinterpret_cast<const char*>(d[i].x) + main_rows * d[i].ldx * 2,
                              d[i].dw, d[i].N, d[i].K, (int)(d[i].rows - main_rows), d[i].ldy, d[i].ldx, d[i].ldw, 1, 0, 1, 1, 0, 0, 0, 0, 0, 0, 1.0f, 2,
                              in_dtype, CIR_F32, stream);
        if (e != CIR_OK) return e;
    }
    return CIR_OK;
}

extern "C" int cir_wgrad(const void* dy, int64_t ldy, const void* x, int64_t ldx, float* dw, int64_t ldw, int64_t rows, int N, int K, int splits,
                         int in_dtype, void* stream) {
    cir_wgrad_desc d;
    d.dy = dy; d.ldy = ldy; d.x = x; d.ldx = ldx; d.dw = dw; d.ldw = ldw; d.rows = rows; d.N = N; d.K = K; d.splits = splits;
    return cir_wgrad_grouped(&d, 1, in_dtype, stream);
}
