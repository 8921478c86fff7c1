// 256x256x64-tile MFMA GEMM for the large shapes of the path (M in the tens of thousands):
// 8 waves (2 along M x 4 along N), each wave owns 128 x 64 outputs = 128 fp32 accumulators per lane.
// PERSISTENT: one workgroup per CU walks tiles t, t + grid, ...; the K loop of this path is short
// (K = 768 -> 12 K-tiles), so the per-tile fixed cost matters as much as the main loop:
//   * the next tile's first K-tile is requested (LDS-DMA) BEFORE the current tile's epilogue, so its
//     HBM/L2 latency hides behind the epilogue's VALU work and store issue;
//   * the bias initialises the accumulators; the epilogue is activation, pack, an LDS transposition to whole output
//     rows, the fp32 residual (fetched row-contiguously two passes ahead) and 16-byte stores that drain during the
//     next main loop.
//
// Main-loop schedule (two-wave-per-SIMD ping-pong after the CDNA4 programming guide, re-derived with conservative
// hazards and re-cut by measurement; see DESIGN.md "GEMM"):
//   * LDS holds two K-tiles (dbuf 0/1), each as four 16-KiB half-tiles A0 A1 (activation rows 0-127 /
//     128-255 of the tile) and B0 B1 (weight rows 0-127 / 128-255); 128 KiB in total, one block per CU.
//   * a K-tile is consumed in TWO phases of 32 MFMAs per wave, one M-half (64 x 64 outputs per wave) each:
//       a = (A0 x B0,B1)  reads B0, A0, B1 (16 ds_read_b128)      b = (A1 x B0,B1)  reads A1 (8; weights kept)
//     phase = { reads + one counted wait } | barrier | { 32 in-place MFMAs, refill pieces between them } | barrier.
//   * the two wave groups (waves 0-3 / 4-7 = the two waves of each SIMD) run one barrier apart, so one group's MFMA
//     segment covers the other group's read segment.
//   * refills go by LDS-DMA (SADDR-form global_load_lds_dwordx4, two 1-KiB pieces per wave and half-tile), always into
//     a half-tile whose last read lies a full phase back (the other group reads one barrier later):
//       phase:   a0            b0               a1            b1
//       refill:  A1>d1 (2)     A0,B0,B1>d0 (6)  A1>d0 (2)     A0,B0,B1>d1 (6)        (pieces per wave)
//       K-tile:  t+1           t+2              t+2           t+3
//     Pieces are issued in the 12 idle issue cycles behind an MFMA: as a block between k-steps (with address selects)
//     an issue idled the matrix pipe ~100 cycles per phase.  The loop body is branch-free (refills past the end of the
//     tile fetch the next tile's first K-tiles, or re-fetch the last K-tile into a dead slot; the last K-tile pair is
//     peeled) and waits are counted: `vmcnt(2)` for the other buffer's A0/B0/B1 in the read segment of b, `vmcnt(6)`
//     for an A1 half-tile in that of a - each one phase (and one barrier) before the first read; every refill has two
//     phases to land.  A K-tile pair takes ~4400 cycles against 4096 cycles of MFMA issue.
// Operand layout, swizzle and the swapped MFMA orientation are those of gemm.hip.

#include <cstdlib>
#include <type_traits>

#include "gemm_args.hpp"

namespace cir {

#ifdef CIR_GEMM_STAMPS
// Diagnostic build only: s_memtime stamps of workgroup 0 / wave 0 (and wave 4) at tile-phase boundaries, written to a
// buffer of their own that no other code reads (MI355X guide: in-kernel stamps).  Never part of the shipped library.
__device__ unsigned long long g_stamps[2][64][8];
__device__ unsigned long long g_blk[512][4];   // per workgroup: s_memtime / s_memrealtime at entry and exit (wave 0)
#define STAMP_BLK(O)                                                                                        \
    if (wave == 0 && lane == 0 && blockIdx.x < 512) {                                                       \
        unsigned long long t_, r_;                                                                          \
        asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_), "=s"(r_) :: "memory"); \
        g_blk[blockIdx.x][O] = t_; g_blk[blockIdx.x][(O) + 1] = r_;                                         \
    }
#define STAMP(SLOT)                                                                                         \
    if (blockIdx.x == 0 && (wave == 0 || wave == 4) && lane == 0 && tile_no < 64) {                        \
        unsigned long long t_;                                                                              \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory");                    \
        g_stamps[wave >> 2][tile_no][SLOT] = t_;                                                            \
    }
__device__ unsigned long long g_pair[2][64][8];  // end of each K-tile pair of the main loop (workgroup 0, waves 0 / 4)
#define STAMP_PAIR(SLOT)                                                                                    \
    if (blockIdx.x == 0 && (wave == 0 || wave == 4) && lane == 0 && tile_no < 64 && (SLOT) < 8) {          \
        unsigned long long t_;                                                                              \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory");                    \
        g_pair[wave >> 2][tile_no][SLOT] = t_;                                                              \
    }
#else
#define STAMP(SLOT)
#define STAMP_BLK(O)
#define STAMP_PAIR(SLOT)
#endif

// (Measured and dropped, round 3: the epilogue of M-half 0 - final after the last K-tile's phase a - as four 16-row mini-passes
// BETWEEN the MFMAs of the tile's last phase (pack, 2 ds_write, 2 ds_read, 2 stores every 8 MFMAs; plain-pack variants, interior
// tiles): correct, the epilogue shrank 2.9 k -> 1.3 k cycles and the last K-tile pair grew 5.2 k -> 7.6-8.3 k: the work is not
// hidden under the matrix pipe, it stretches the phase; 75.2 against 74.8 ms of GEMM time per step.)
// Output stores are NON-TEMPORAL (`nt`): plain stores leave the tile's 128 KiB of C lines dirty in the XCD's L2, where they
// displace the weight / activation panels the next tiles re-read (every K-tile pair of the following tile 5.2-5.6 k cycles instead
// of 4.4 k); with `nt` only the first pair behind the stores is slow.  Measured on the wide GEMMs of a step, same box: plain
// 1171 / 1148 / 1022 TFLOP/s (K|V, QKV, fc1+GELU), nt 1240 / 1188 / 1057, sc1 1208 / 1184 / 1059, sc0 sc1 1213 / 1170 / 1052.
// (Inline asm: wait states close the statement - a store wider than 64 bits reads its data registers after issue and the
// hazard recogniser does not see inline asm.)
#define STORE_C(P, V) asm volatile("global_store_dwordx4 %0, %1, off nt\n\ts_nop 1" :: "v"(P), "v"(V) : "memory");

constexpr int T256 = 256;
constexpr int kHalf = 16384;       // one half-tile: 128 rows x 64 k x 2 B
constexpr int kDbuf = 4 * kHalf;   // A0 A1 B0 B1

// Epilogues: OUT_F32 = fp32 C (and fp32 R) through an fp32 staging tile (8 passes); otherwise the 4-pass 16-bit epilogue with
// C in the operand type T (ST = float: no residual) or in the 16-bit residual-stream type ST (_Float16; also from bf16
// operands), whose residual R (type ST) is added AFTER the transposition to whole rows: acc + bias (+ act) is rounded to ST,
// the residual is added to it in fp32 and the sum is rounded again.  (An fp32-staged variant with a single rounding was
// built first: 495 us instead of 432 us on the ViT proj shape - its 8 passes each wait on the LDS round trip.)
// ACT: the activation as a compile-time constant (CIR_ACT_NONE / CIR_ACT_GELU: the two the path runs at scale), or -1 = read
// a.act at run time (a branch and a register copy per 8 outputs; kept for the rarely used combinations).
// LNF ("LayerNorm folded", round 5): A holds the RAW residual-stream rows x (K = the whole row), W the weight with the LayerNorm gain
// folded in (W' = W diag(gamma)), bias b' = b + W beta, colsum s_n = sum_k W'[n,k]:
//     C = act( rstd_m * (x W'^T - mean_m * s) + b' )  =  act( LayerNorm(x) W^T + b )
// The row statistics come out of the main loop itself: the activation fragments a wave feeds to the MFMA pass through two
// v_dot2_f32_f16 per dword (sum and sum of squares in fp32; the four waves that hold the same 64 rows take one 16-row block each =
// 16 VALU ops per phase, in the issue gaps behind the MFMAs), are reduced over the four k-groups of lanes at the end of the tile and
// exchanged through LDS.  No LayerNorm pass, no normalised copy of x in HBM (vit.py:107-109: norm1 / norm2 of every block).
// MIX ("split8" operands, round 6; common.hpp): A and W rows are byte rows [K fp16 | K e4m3 | K e4m3]; the K loop walks the same 128-byte
// K-tiles through the same staging, reads and barriers, and only the MFMA of a K-tile pair changes: pairs of the leading segment run the
// 32 fp16 MFMAs per phase, pairs of the two correction segments 16 block-scaled fp8 MFMAs (v_mfma_scale_f32_16x16x128_f8f6f4: the lane's
// two 16-byte chunks of a row taken together as 32 e4m3 values of k - both operands in the same chunk order) of twice the duration each:
// a K-tile costs the same matrix-pipe time and carries twice the depth.  OSPL (with MIX): C is written as split8 rows of act(acc + bias)
// - the next GEMM's operand (fc1 -> fc2) - through the fp32 epilogue's 256-byte staging rows: 128 bytes of fp16 + 64 + 64 bytes of e4m3.
template <typename T, bool OUT_F32, bool HAS_RES, typename ST = float, int ACT = -1, bool LNF = false, bool MIX = false, bool OSPL = false>
__global__ __launch_bounds__(512, 2) void gemm256_kernel(const GemmArgs a) {
    constexpr bool FAST16 = !__is_same(ST, float);        // 16-bit residual-stream output (and residual)
    static_assert(!(FAST16 && OUT_F32), "the stream type is written by the 16-bit epilogue");
    static_assert(!LNF || (__is_same(T, _Float16) && !OUT_F32 && !HAS_RES && !FAST16), "LNF: fp16 stream rows in, operand-type C out");
    static_assert(!MIX || (__is_same(T, _Float16) && OUT_F32 && !LNF), "MIX: split8 operand rows in, fp32 accumulators out through the fp32-layout epilogue");
    static_assert(!OSPL || (MIX && !HAS_RES), "OSPL: split8 rows out of a MIX GEMM without a residual");
    using OT = typename std::conditional<FAST16, ST, T>::type;   // element type the 16-bit epilogue packs to
    using X8 = typename Elem<T>::x8;
    __shared__ __attribute__((aligned(16))) char smem[2 * kDbuf + 8 * 1024 + 4 * 4096];   // two K-tiles + a 1-KiB bias slot per wave + epilogue staging for waves 4-7

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;   // wr doubles as the stagger group (SIMD partners differ in wr)
    const int r15 = lane & 15, g = lane >> 4;
    const int srow = lane >> 3;
    const int schunk = (lane & 7) ^ srow;
    const int nk = a.K >> 6;
    const int per_batch = a.tiles_m * a.tiles_n;
    const int ntiles = per_batch * a.batch;
    char* const stage_base = smem + wave * 2048;

    // ---- per-tile state ----------------------------------------------------------------------------------------
    // Operand addresses = wave-uniform tile base (SGPR pair) + 32-bit per-lane byte offset RELATIVE to the tile.  For
    // interior tiles the lane offsets are tile-invariant (computed once); only tiles on the ragged M / N edge recompute
    // them with clamped rows.  setup() is then scalar work for almost every tile.
    int64_t m0 = 0;
    int n0 = 0, z = 0;
    const char* A_z = nullptr;
    const char* W_z = nullptr;
    unsigned a_off[2][2], w_off[2][2];
    bool rel_interior = false;
    auto lane_offsets = [&](bool interior) {
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int lr = (wave * 2 + j) * 8 + srow;             // LDS row inside the half-tile
                // half h of the weight tile holds, for each consumer wave wc, features wc*64 + h*32 + [0,32): a wave's
                // two halves are adjacent, so it owns 64 contiguous features = whole 128-byte lines of 16-bit output;
                // within a half, rows are permuted so that accumulator lane group g owns 8 consecutive features
                const int feat = (lr >> 5) * 64 + h * 32 + ((lr & 15) >> 2) * 8 + ((lr >> 4) & 1) * 4 + (lr & 3);
                int64_t rm = h * 128 + lr;
                int rn = feat;
                if (!interior) {
                    rm = (m0 + rm < a.M ? m0 + rm : a.M - 1) - m0;    // clamp the ragged edges (stores are predicated)
                    rn = (n0 + rn < a.N ? n0 + rn : a.N - 1) - n0;
                }
                a_off[h][j] = (unsigned)((rm * a.lda + schunk * 8) * 2);
                w_off[h][j] = (unsigned)(((int64_t)rn * a.ldw + schunk * 8) * 2);
            }
    };
    struct Coords { int64_t m0; int n0, z; const char* A; const char* W; bool interior; };
    auto coords = [&](int t) -> Coords {
        Coords c;
        int id = xcd_remap(t, ntiles);
        c.z = id / per_batch;
        id -= c.z * per_batch;
        // column groups of `group_w` weight panels: an XCD's 32 concurrent tiles then share few weight panels
        // (resident in its 4 MiB L2) and stream the activation panels, each used by group_w tiles at once.
        // (Measured and dropped, round 3: super-blocks of 8-32 m-tiles whose column groups are all walked before the next
        // super-block - second pass over the activation rows on-die - and a last-to-first m order that starts on the rows
        // the preceding kernel wrote last: no change of the benchmark step within 0.2 %; HBM reads are not what the loop waits for.)
        const int gsz = a.tiles_m * a.group_w;
        const int grp = id / gsz, rem = id - grp * gsz;
        const int first_n = grp * a.group_w;
        const int gw = min(a.group_w, a.tiles_n - first_n);
        const int tile_m = rem / gw, tile_n = first_n + (rem - tile_m * gw);
        c.m0 = (int64_t)tile_m * T256;
        c.n0 = tile_n * T256;
        c.A = reinterpret_cast<const char*>(reinterpret_cast<const T*>(a.A) + c.z * a.sA + c.m0 * a.lda);
        c.W = reinterpret_cast<const char*>(reinterpret_cast<const T*>(a.W) + c.z * a.sW + (int64_t)c.n0 * a.ldw);
        c.interior = c.m0 + T256 <= a.M && c.n0 + T256 <= a.N;
        return c;
    };
    auto adopt = [&](const Coords& c) {   // make `c` the current tile (refreshes lane offsets only on / after edge tiles)
        m0 = c.m0; n0 = c.n0; z = c.z; A_z = c.A; W_z = c.W;
        if (!c.interior || !rel_interior) lane_offsets(c.interior);
        rel_interior = c.interior;
    };
    // Operand stream across tile seams: when this and the next tile are interior (same lane offsets), the refill slots of
    // the LAST K-tile pair - which have nothing of this tile left to fetch - request the next tile's first K-tiles,
    // exactly the set a prologue would request, several phases before this tile even ends.
    const char* A_nx = nullptr;
    const char* W_nx = nullptr;
    bool stream = false;

// One half-tile = two 1-KiB pieces per wave (P selects the piece).  Refills are UNCONDITIONAL and branch-free: every
// K-tile pair issues the same 16 pieces per wave from wave-uniform base pointers computed outside the MFMA segments;
// in the (peeled) last pair those bases point at the next tile's first K-tiles when the seam streams and otherwise at
// this tile's last K-tile, re-fetched into a slot nobody reads again (a later prologue of the same wave overwrites it
// in order).  The counted waits therefore need no cases, and the pieces can sit between the MFMAs.
// piece P of half-tile H from the K-tile whose (wave-uniform) base pointer is KB
// (inline asm, SADDR form: wave-uniform 64-bit base in an SGPR pair + the lane's 32-bit offset; the builtin takes a
// 64-bit per-lane address, i.e. two VALU ops per piece and - hoisted out of the loop by the compiler - 32 more VGPRs)
#define LDS_DMA(SRC_BASE, VOFF, LDS_DST)                                                                                 \
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2"                                       \
                 :: "s"((unsigned)(size_t)(lptr_t)(LDS_DST)), "v"(VOFF), "s"(SRC_BASE) : "memory");   /* (M0 is reserved: the compiler re-loads it before each of its own uses) */
#define PIECE_A(H, DB, KB, P) LDS_DMA(KB, a_off[H][P], stage_base + (DB) * kDbuf + (H) * kHalf + (P) * 1024)
#define PIECE_B(H, DB, KB, P) LDS_DMA(KB, w_off[H][P], stage_base + (DB) * kDbuf + (2 + (H)) * kHalf + (P) * 1024)
// (prologue: K-tiles 0 and 1 of the adopted tile - the dispatcher guarantees at least two)
#define ISSUE_A1(H, DB, KT, P) { const char* kb_ = A_z + (KT) * 128; PIECE_A(H, DB, kb_, P) }
#define ISSUE_B1(H, DB, KT, P) { const char* kb_ = W_z + (KT) * 128; PIECE_B(H, DB, kb_, P) }
#define ISSUE_A(H, DB, KT) ISSUE_A1(H, DB, KT, 0) ISSUE_A1(H, DB, KT, 1)
#define ISSUE_B(H, DB, KT) ISSUE_B1(H, DB, KT, 0) ISSUE_B1(H, DB, KT, 1)
    // first K-tile complete in dbuf 0 plus A0, B0, B1 of K-tile 1 (what the last K-tile pair of a streaming seam issues)
#define ISSUE_PROLOGUE()                                                \
    ISSUE_A(0, 0, 0) ISSUE_A(1, 0, 0) ISSUE_B(0, 0, 0) ISSUE_B(1, 0, 0) \
    ISSUE_A(0, 1, 1) ISSUE_B(0, 1, 1) ISSUE_B(1, 1, 1)

    f32x4 acc[2][4][2][2];  // [m-half][m-tile][n-half][n-tile]
    const int swz = r15 & 7;
    const int c0 = ((0 + g) ^ swz) << 4, c1 = ((4 + g) ^ swz) << 4;   // 16-byte chunk offsets of the two k-steps
    const char* const a_rd = smem + (wr * 64 + r15) * 128;
    const char* const b_rd = smem + 2 * kHalf + (wc * 32 + r15) * 128;
    X8 af[4][2], wf[2][2][2];   // both weight halves stay in registers for the whole K-tile (B0 is used by q0 and q3)

#define READ_A(H, DB)                                                                                  \
    _Pragma("unroll") for (int mi = 0; mi < 4; ++mi) {                                                 \
        af[mi][0] = *reinterpret_cast<const X8*>(a_rd + (DB) * kDbuf + (H) * kHalf + mi * 2048 + c0);  \
        af[mi][1] = *reinterpret_cast<const X8*>(a_rd + (DB) * kDbuf + (H) * kHalf + mi * 2048 + c1);  \
    }
#define READ_B(H, DB)                                                                                  \
    _Pragma("unroll") for (int ni = 0; ni < 2; ++ni) {                                                 \
        wf[H][ni][0] = *reinterpret_cast<const X8*>(b_rd + (DB) * kDbuf + (H) * kHalf + ni * 2048 + c0);  \
        wf[H][ni][1] = *reinterpret_cast<const X8*>(b_rd + (DB) * kDbuf + (H) * kHalf + ni * 2048 + c1);  \
    }
#define SYNC()                                   \
    __builtin_amdgcn_sched_barrier(0);           \
    __builtin_amdgcn_s_barrier();                \
    __builtin_amdgcn_sched_barrier(0);
// MFMA segment of one phase: the 32 in-place MFMAs of one M-half (64 x 64 outputs per wave, two k-steps, both weight
// halves) in SOURCE order, with the phase's refill pieces placed between them.  An LDS-DMA issue (M0, address add, the
// VMEM issue itself) costs the wave some tens of cycles; as a block between the k-steps - with its address selects - it
// idled the matrix pipe for ~100 cycles per phase (measured: MFMA segments of 350-400 cycles for 256 cycles of MFMA
// work); behind an MFMA, whose issue leaves the wave 12 idle cycles, it is nearly free.
#define MM(MH, I) Elem<T>::mfma16_acc(acc[MH][((I) & 7) >> 1][(I) >> 4][(I) & 1], wf[(I) >> 4][(I) & 1][((I) >> 3) & 1], af[((I) & 7) >> 1][((I) >> 3) & 1]);
#define MM4(MH, I) MM(MH, I) MM(MH, (I) + 1) MM(MH, (I) + 2) MM(MH, (I) + 3)
#define PLACE(X) __builtin_amdgcn_sched_barrier(0); X __builtin_amdgcn_sched_barrier(0);
#define MFMAS_0(MH, P0, P1, P2, P3, P4, P5) MM4(MH, 0) MM4(MH, 4) MM4(MH, 8) MM4(MH, 12) MM4(MH, 16) MM4(MH, 20) MM4(MH, 24) MM4(MH, 28)
#define MFMAS_2(MH, P0, P1, P2, P3, P4, P5)                                                                              \
    MM4(MH, 0) MM4(MH, 4) PLACE(P0) MM4(MH, 8) MM4(MH, 12) MM4(MH, 16) PLACE(P1) MM4(MH, 20) MM4(MH, 24) MM4(MH, 28)
#define MFMAS_6(MH, P0, P1, P2, P3, P4, P5)                                                                              \
    MM4(MH, 0) PLACE(P0) MM4(MH, 4) MM(MH, 8) PLACE(P1) MM(MH, 9) MM(MH, 10) MM(MH, 11) MM4(MH, 12) PLACE(P2)            \
    MM4(MH, 16) MM(MH, 20) PLACE(P3) MM(MH, 21) MM(MH, 22) MM(MH, 23) MM(MH, 24) PLACE(P4) MM(MH, 25) MM(MH, 26)         \
    MM(MH, 27) MM(MH, 28) PLACE(P5) MM(MH, 29) MM(MH, 30) MM(MH, 31)
// MIX: the 16 block-scaled fp8 MFMAs of a phase (one per accumulator: the two chunks a lane read of a row are ONE 32-value operand);
// each lasts two fp16 MFMAs, so the refill pieces keep their places in time.  sc_w / sc_a: the E8M0 scale words of the pair's segment.
[[maybe_unused]] int sc_a = 0, sc_w = 0;
#define CAT8(X) __builtin_bit_cast(i32x8, __builtin_shufflevector((X)[0], (X)[1], 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15))
#define MF(MH, J) if constexpr (MIX) { mfma_f8_acc(acc[MH][((J) & 7) >> 1][(J) >> 3][(J) & 1], CAT8(wf[(J) >> 3][(J) & 1]), CAT8(af[((J) & 7) >> 1]), sc_w, sc_a); }
#define MF2(MH, J) MF(MH, J) MF(MH, (J) + 1)
#define MFMAS8_2(MH, P0, P1, P2, P3, P4, P5)                                                                             \
    MF2(MH, 0) MF2(MH, 2) PLACE(P0) MF2(MH, 4) MF2(MH, 6) MF2(MH, 8) PLACE(P1) MF2(MH, 10) MF2(MH, 12) MF2(MH, 14)
#define MFMAS8_6(MH, P0, P1, P2, P3, P4, P5)                                                                             \
    MF2(MH, 0) PLACE(P0) MF2(MH, 2) MF(MH, 4) PLACE(P1) MF(MH, 5) MF2(MH, 6) PLACE(P2) MF2(MH, 8) MF(MH, 10) PLACE(P3)   \
    MF(MH, 11) MF(MH, 12) PLACE(P4) MF(MH, 13) MF(MH, 14) PLACE(P5) MF(MH, 15)
// LNF: row statistics.  The four waves of a row group hold the same 64 rows; each takes one 16-row block (its column index wc) and
// reads that block's two fragments of the K-tile a SECOND time into registers of their own (sf) - a wave-dependent choice among the
// af registers is either a branch chain in the MFMA stream (measured: +29 % kernel time, and the compiler's merge of the arms lost
// terms) or 24 selects per phase; the two extra ds_read_b128 per phase (18 / 10 instead of 16 / 8) measured free.  Per phase 32
// v_fma_mix_f32 (fp16 sources, fp32 accumulators: sum += x * 1.0, sumsq += x * x per element) in the READ segment of the phase, behind
// the issue of the phase's LDS reads, where they run under the OTHER wave group's MFMA segment: ~6 % of the kernel's time (8 % with the
// GELU epilogue); the arrangement is what made them affordable at all - 16 v_dot2c_f32_f16 (two elements per op) cost +146 cycles per
// phase between this wave's own MFMAs and +177 in the read segment (dependent 2-pass ops), and four independent dot2c chains spill.
// (inline asm, volatile: left to the compiler the ops of four phases were sunk into one block of 64 behind copies of the fragments)
[[maybe_unused]] float ssum[2] = {0.f, 0.f}, ssq[2] = {0.f, 0.f};
[[maybe_unused]] X8 sf[2];
[[maybe_unused]] const char* const s_rd = a_rd + wc * 2048;
#define READ_S(H, DB)                                                                                  \
    if constexpr (LNF) {   /* asm: issued FIRST and waited for with a COUNTED lgkmcnt (NY younger reads may stay in flight) */ \
        const unsigned s0_ = (unsigned)(size_t)(lptr_t)(s_rd + (DB) * kDbuf + (H) * kHalf + c0);       \
        const unsigned s1_ = (unsigned)(size_t)(lptr_t)(s_rd + (DB) * kDbuf + (H) * kHalf + c1);       \
        __builtin_amdgcn_sched_barrier(0);                                                             \
        asm volatile("ds_read_b128 %0, %2\n\tds_read_b128 %1, %3" : "=&v"(sf[0]), "=&v"(sf[1]) : "v"(s0_), "v"(s1_)); \
        __builtin_amdgcn_sched_barrier(0);                                                             \
    }
#define ST1(MH, D)                                                                                     \
    { const unsigned d_ = __builtin_bit_cast(u32x4, sf[(D) >> 2])[(D) & 3];                            \
      asm volatile("v_fma_mix_f32 %0, %2, 1.0, %0 op_sel:[0,0,0] op_sel_hi:[1,0,0]\n\tv_fma_mix_f32 %1, %2, %2, %1 op_sel:[0,0,0] op_sel_hi:[1,1,0]\n\t" \
                   "v_fma_mix_f32 %0, %2, 1.0, %0 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\tv_fma_mix_f32 %1, %2, %2, %1 op_sel:[1,1,0] op_sel_hi:[1,1,0]" \
                   : "+v"(ssum[MH]), "+v"(ssq[MH]) : "v"(d_)); }
#define STATS(MH, NY)   /* NY = ds_read_b128 the compiler issued between READ_S and here */          \
    if constexpr (LNF) {                                                                               \
        __builtin_amdgcn_sched_barrier(0);                                                             \
        asm volatile("s_waitcnt lgkmcnt(%0)" :: "n"(NY) : "memory");                                   \
        ST1(MH, 0) ST1(MH, 1) ST1(MH, 2) ST1(MH, 3) ST1(MH, 4) ST1(MH, 5) ST1(MH, 6) ST1(MH, 7)        \
        __builtin_amdgcn_sched_barrier(0);                                                             \
    }
#define COMPUTE(KIND, MH, NDMA, P0, P1, P2, P3, P4, P5) \
    SYNC();                                      \
    __builtin_amdgcn_s_setprio(1);               \
    MFMAS##KIND##_##NDMA(MH, P0, P1, P2, P3, P4, P5)     \
    __builtin_amdgcn_s_setprio(0);               \
    SYNC();
// Counted waits, one per half of a K-tile, each in the read segment ONE phase before the half is first read (behind a
// barrier).  Issue order per K-tile pair: M(a0) A1>d1 (2 pieces) | M(b0) A0,B0,B1>d0 (6) | M(a1) A1>d0 (2) |
// M(b1) A0,B0,B1>d1 (6).  WAIT_ABB retires the other buffer's A0/B0/B1 (only the A1 issued after them may stay in
// flight: vmcnt(2)); WAIT_A1 retires an A1 half-tile (the 6 pieces issued after it may stay: vmcnt(6)).  Every
// half-tile has two phases (~2 x 32 MFMAs x 2 groups) to land.
// Vector memory retires in order, and the previous tile's output stores are OLDER than every refill of this tile: a
// counted wait cannot retire a refill before those stores have drained.  The first two waits of a tile only concern
// data the prologue delivered (already retired at the head of the tile), so a tile's first K-tile pair skips them
// (`skip`) instead of stalling on its predecessor's stores.
#define WAIT_N(N) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N) : "memory");
#define WAIT_PRO(N) if (!skip) { WAIT_N(N) }
#ifdef CIR_GEMM_STAMPS   // experiment (wrong results): drop the first pair's waits that stand behind the previous tile's stores
#define WAIT_ABB() if (!(skip && (a.dbg & 0x800))) { WAIT_N(2) }
#define WAIT_A1() if (!(skip && (a.dbg & 0x400))) { WAIT_N(6) }
#else
#define WAIT_ABB() WAIT_N(2)
#define WAIT_A1() WAIT_N(6)
#endif

    static_assert(!HAS_RES || OUT_F32 || FAST16, "the residual is added in a row layout of the epilogue");

    // bias of the tile arrives by LDS-DMA into a private 1-KiB slot per wave (no VGPR-destination loads on the
    // tile boundary: the compiler would answer those with vmcnt(0), which also waits for the previous tile's stores)
    char* const bias_slot = smem + 2 * kDbuf + wave * 1024;
// LNF: the slot holds the wave's OWN 64 features of b' (lanes 0-15 -> bytes 0-255) and of the column sums s (lanes 16-31 ->
// bytes 256-511); bytes 512-767 are the wave's page of the row-statistics exchange (lanes 32-63 stay masked off).  Both vectors are
// consumed at the HEAD of the epilogue (before the next tile's request overwrites them), not at the head of the tile.
#define ISSUE_BIAS()                                                                                                   \
    if constexpr (LNF) {                                                                                               \
        int ln_ = lane;                                                                                                \
        asm volatile("" : "+v"(ln_));   /* recomputed per tile: hoisted out of the tile loop these lane values spilled */ \
        int nm4_ = __builtin_amdgcn_readfirstlane(a.N - 4);                                                            \
        asm volatile("" : "+s"(nm4_));   /* (as a hoisted VGPR copy this scalar spilled: its reload is a vmcnt(0)) */     \
        if (ln_ < 32) {                                                                                                \
            const int bn = min(n0 + wc * 64 + (ln_ & 15) * 4, nm4_);                                                   \
            const float* src_ = (ln_ & 16) ? a.colsum : a.bias;                                                        \
            __builtin_amdgcn_global_load_lds((gptr_t)(src_ + bn), (lptr_t)(bias_slot), 16, 0, 0);                      \
        }                                                                                                              \
    } else if (a.bias != nullptr) {                                                                                    \
        int bn = n0 + lane * 4;                                                                                        \
        bn = bn + 4 <= a.N ? bn : a.N - 4;   /* clamped lanes feed features whose outputs are never stored */          \
        __builtin_amdgcn_global_load_lds((gptr_t)(a.bias + z * a.sBias + bn), (lptr_t)(bias_slot), 16, 0, 0);         \
    }

    int t = blockIdx.x;
#ifdef CIR_GEMM_STAMPS
    // experiment: de-phase the workgroups (their epilogue store bursts) by (blockIdx/8 mod 4) x (dbg & 0xff) sleeps of ~4 us
    for (int i = 0, n = (a.dbg & 0xff) * ((blockIdx.x >> 3) & 3); i < n; ++i) __builtin_amdgcn_s_sleep(127);
#endif
    STAMP_BLK(0)
    adopt(coords(t));
    ISSUE_PROLOGUE()
    ISSUE_BIAS()
    int pending_stores = 0;   // store instructions this wave issued after its newest loads (0 = unknown -> full drain)
    [[maybe_unused]] int tile_no = 0;

    for (;;) {
        const int64_t cm0 = m0;
        const int cn0 = n0, cz = z;
        const int tn = t + (int)gridDim.x;
        const bool more = tn < ntiles;
        Coords nx = coords(more ? tn : t);
        A_nx = nx.A; W_nx = nx.W;
        stream = false;   // the prologue of THIS tile must not be mistaken for a streamed one while waiting below
        STAMP(0)
        // ---- operands of the first K-tile(s) and the bias have landed; the previous tile's stores may still fly ------
        if (pending_stores == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else if (pending_stores == 16) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(32)" ::: "memory");
        STAMP(1)
        // ---- accumulators start at the bias ----------------------------------------------------------------------------
        // (Measured and dropped, round 3: the bias fragment as the C operand of each accumulator's first MFMA instead of this
        //  pass - 16 more registers live through the first K-tile: the initialisation shrank 470 -> 250 cycles, the seam
        //  around it grew by more; 72.9 against 72.5 ms of GEMM time per step.)
        {
            float bias[2][8];
            if constexpr (LNF) {   // b' joins in the epilogue, behind the row scale
#pragma unroll
                for (int q = 0; q < 8; ++q) { bias[0][q] = 0.f; bias[1][q] = 0.f; }
                ssum[0] = ssum[1] = ssq[0] = ssq[1] = 0.f;
            } else if (a.bias != nullptr) {
                // LDS reads hidden from the compiler's wait-count pass: it would put vmcnt(0) in front of a ds_read of
                // DMA-written LDS.  Loads and their lgkmcnt wait live in ONE statement (outputs early-clobber).
                f32x4 b00, b01, b10, b11;
                const unsigned baddr = (unsigned)(size_t)(lptr_t)(bias_slot) + (unsigned)((wc * 64 + g * 8) * 4);
                asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %4 offset:16\n\tds_read_b128 %2, %4 offset:128\n\t"
                             "ds_read_b128 %3, %4 offset:144\n\ts_waitcnt lgkmcnt(0)"
                             : "=&v"(b00), "=&v"(b01), "=&v"(b10), "=&v"(b11) : "v"(baddr) : "memory");
#pragma unroll
                for (int q = 0; q < 4; ++q) { bias[0][q] = b00[q]; bias[0][4 + q] = b01[q]; bias[1][q] = b10[q]; bias[1][4 + q] = b11[q]; }
            } else {
#pragma unroll
                for (int q = 0; q < 8; ++q) { bias[0][q] = 0.f; bias[1][q] = 0.f; }
            }
#pragma unroll
            for (int mh = 0; mh < 2; ++mh)
#pragma unroll
                for (int mi = 0; mi < 4; ++mi)
#pragma unroll
                    for (int nh = 0; nh < 2; ++nh) {
                        acc[mh][mi][nh][0] = f32x4{bias[nh][0], bias[nh][1], bias[nh][2], bias[nh][3]};
                        acc[mh][mi][nh][1] = f32x4{bias[nh][4], bias[nh][5], bias[nh][6], bias[nh][7]};
                    }
            // materialise every accumulator now (nothing of the initialisation may sink into the K loop)
#pragma unroll
            for (int mh = 0; mh < 2; ++mh)
#pragma unroll
                for (int mi = 0; mi < 4; ++mi)
#pragma unroll
                    for (int nh = 0; nh < 2; ++nh) {
                        asm volatile("" : "+v"(acc[mh][mi][nh][0]), "+v"(acc[mh][mi][nh][1]));
                    }
        }
        STAMP(2)
        SYNC();
        if (wr == 1) { SYNC(); }   // stagger: waves 4-7 run one barrier behind waves 0-3
        stream = more && nx.interior && rel_interior;
        STAMP(3)

        // K-tile T0 from dbuf 0 (phases 1-4) / T1 = T0+1 from dbuf 1 (phases 5-8); refills: see the schedule at the top.
        // PA1 / PA2,PW2 / PA3,PW3: wave-uniform base pointers of K-tiles T0+1, T0+2, T0+3 (computed outside: no
        // selects or branches between the MFMAs).
        // LNF: the row statistics are complete once the last phase's fragments have been added, i.e. in the READ segment of the tile's
        // last phase.  There - under the other wave group's MFMA segment - they are summed over the four k-groups of lanes and
        // (rstd, -rstd * mean) of this wave's two 16-row blocks go to its page of the exchange; the two barriers of that phase (and
        // the trailing one of the staggered group) order the writes before any wave's epilogue reads: no barrier of their own.
        // (A barrier + this block at the head of the epilogue: +1.7 k cycles per tile.)
        // Cross-lane sum over lanes l, l ^ 16, l ^ 32, l ^ 48 with the gfx950 row / half swaps: copy, swap the upper half (odd rows)
        // of one copy with the lower half (even rows) of the other, add - plain VALU, where __shfl_xor is two dependent round trips
        // through the LDS crossbar.  (asm: with both operands the same value the builtin's two results were folded into one -
        // v_add v1, v1, v1; wait states by hand, the hazard recogniser does not look inside)
#define PUBLISH_STATS()                                                                                                      \
        if constexpr (LNF) {                                                                                                 \
            const float inv_k = 1.0f / (float)a.K;                                                                           \
            int lp_ = lane;                                                                                                  \
            asm volatile("" : "+v"(lp_));                                                                                    \
            const unsigned xw = (unsigned)(size_t)(lptr_t)(bias_slot) + 512 + (lp_ & 15) * 8;                                \
            _Pragma("unroll") for (int mh = 0; mh < 2; ++mh) {                                                               \
                float sx = ssum[mh], sq = ssq[mh], t0, t1;                                                                   \
                asm volatile("v_mov_b32 %2, %0\n\tv_mov_b32 %3, %1\n\ts_nop 1\n\tv_permlane32_swap_b32 %0, %2\n\tv_permlane32_swap_b32 %1, %3\n\ts_nop 1\n\t" \
                             "v_add_f32 %0, %0, %2\n\tv_add_f32 %1, %1, %3\n\tv_mov_b32 %2, %0\n\tv_mov_b32 %3, %1\n\ts_nop 1\n\t"           \
                             "v_permlane16_swap_b32 %0, %2\n\tv_permlane16_swap_b32 %1, %3\n\ts_nop 1\n\tv_add_f32 %0, %0, %2\n\tv_add_f32 %1, %1, %3" \
                             : "+v"(sx), "+v"(sq), "=&v"(t0), "=&v"(t1));                                                    \
                const float mean = sx * inv_k;                                                                               \
                const float var = fmaxf(fmaf(sq, inv_k, -mean * mean), 0.f);                                                 \
                const float rstd = __builtin_amdgcn_rsqf(var + a.ln_eps);   /* v_rsq_f32: 1 ulp, far inside the fp16 output's rounding */ \
                const f32x2 st = {rstd, -rstd * mean};                                                                       \
                if (lp_ < 16) asm volatile("ds_write_b64 %0, %1" :: "v"(xw + mh * 128), "v"(st) : "memory");                 \
            }                                                                                                                \
            __builtin_amdgcn_sched_barrier(0);                                                                               \
        }
#define KTILE_D0(KIND, PA1, PA2, PW2)                                                                                   \
        READ_S(0, 0) READ_B(0, 0) READ_A(0, 0) STATS(0, 12) READ_B(1, 0) WAIT_PRO(6)                                                               \
            COMPUTE(KIND, 0, 2, PIECE_A(1, 1, PA1, 0), PIECE_A(1, 1, PA1, 1), , , , )                   /* phase a0 */   \
        READ_S(1, 0) READ_A(1, 0) STATS(1, 8) WAIT_PRO(2)                                                                                        \
            COMPUTE(KIND, 1, 6, PIECE_A(0, 0, PA2, 0), PIECE_B(0, 0, PW2, 0), PIECE_B(1, 0, PW2, 0),                     \
                    PIECE_A(0, 0, PA2, 1), PIECE_B(0, 0, PW2, 1), PIECE_B(1, 0, PW2, 1))                /* phase b0 */
#define KTILE_D1(KIND, PA2, PA3, PW3, HOOK)                                                                                   \
        READ_S(0, 1) READ_B(0, 1) READ_A(0, 1) STATS(0, 12) READ_B(1, 1) WAIT_A1()                                                                \
            COMPUTE(KIND, 0, 2, PIECE_A(1, 0, PA2, 0), PIECE_A(1, 0, PA2, 1), , , , )                   /* phase a1 */   \
        READ_S(1, 1) READ_A(1, 1) STATS(1, 8) HOOK WAIT_ABB()                                                                                         \
            COMPUTE(KIND, 1, 6, PIECE_A(0, 1, PA3, 0), PIECE_B(0, 1, PW3, 0), PIECE_B(1, 1, PW3, 0),                     \
                    PIECE_A(0, 1, PA3, 1), PIECE_B(0, 1, PW3, 1), PIECE_B(1, 1, PW3, 1))                /* phase b1 */
        // K-tiles past the end of this tile: the next tile's first two when the seam streams, else this tile's last one
        const char* const A_e0 = stream ? A_nx : A_z + (nk - 1) * 128;
        const char* const A_e1 = stream ? A_nx + 128 : A_e0;
        const char* const W_e0 = stream ? W_nx : W_z + (nk - 1) * 128;
        const char* const W_e1 = stream ? W_nx + 128 : W_e0;
        {
            const char* pa = A_z;
            const char* pw = W_z;
            bool skip = true;                            // first K-tile pair of the tile (see WAIT_PRO)
            // MIX: pairs 1 .. k16 / 2 are the fp16 segment, the next k16 / 4 pairs A_lo x W_hi8, the rest (the peeled last pair
            // included) A_hi8 x W_lo8 - a wave-uniform branch per pair around two copies of the same schedule
            [[maybe_unused]] const int p16 = a.k16 >> 1, p_seg2 = p16 + (a.k16 >> 2);
            // (two loops in sequence, not a branch inside one: with both schedules in one loop body the register allocator spilled 140
            //  registers - accumulators among them - although each schedule alone fits)
            int it = 1;
            for (; it < (MIX ? p16 + 1 : (nk >> 1)); ++it) {     // all K-tile pairs but the last: refills stay inside the tile
                KTILE_D0(, pa + 128, pa + 256, pw + 256)
                KTILE_D1(, pa + 256, pa + 384, pw + 384, )
                pa += 256;
                pw += 256;
                skip = false;
                STAMP_PAIR(it - 1)
            }
            if constexpr (MIX) {
                for (; it < (nk >> 1); ++it) {
                    sc_a = it <= p_seg2 ? a.sc_a1 : a.sc_a2;
                    sc_w = it <= p_seg2 ? a.sc_w1 : a.sc_w2;
                    KTILE_D0(8, pa + 128, pa + 256, pw + 256)
                    KTILE_D1(8, pa + 256, pa + 384, pw + 384, )
                    pa += 256;
                    pw += 256;
                }
            }
            // last pair (the dispatcher sends only K % 128 == 0 here): its refills are the K-tiles past the end
            if constexpr (MIX) {
                sc_a = a.sc_a2;
                sc_w = a.sc_w2;
                KTILE_D0(8, pa + 128, A_e0, W_e0)
                KTILE_D1(8, A_e0, A_e1, W_e1, )
            } else {
                KTILE_D0(, pa + 128, A_e0, W_e0)
                KTILE_D1(, A_e0, A_e1, W_e1, PUBLISH_STATS())
            }
        }
        STAMP(4)
        if (wr == 0) { SYNC(); }   // pair the trailing barrier of the staggered group: all LDS reads of this tile are done
        STAMP(5)

        // ---- epilogue geometry; the first residual passes are requested BEFORE the next tile's operands (vector memory
        //      returns in order: a residual load queued behind the prologue would wait for all of it) ------------------
        const bool full = cm0 + T256 <= a.M && cn0 + T256 <= a.N;
        constexpr int ROWS = OUT_F32 ? 16 : 32;                 // rows per pass
        constexpr int NPASS = 128 / ROWS;
        constexpr int ROWB = OUT_F32 ? 256 : 128;               // bytes of this wave's 64 features in one row
        constexpr int LPR = ROWB / 16;                          // lanes per row when reading back
        // (LNF: the epilogue's lane geometry is recomputed per tile - kept live across the K loop, next to the statistics registers, it spills)
        int lane_e = lane;
        if constexpr (LNF) asm volatile("" : "+v"(lane_e));
        [[maybe_unused]] const int r15e = lane_e & 15, ge = lane_e >> 4;
        const int rr = lane_e / LPR, sl = lane_e % LPR;
        const int ncol = cn0 + wc * 64 + sl * (OUT_F32 ? 4 : 8);
        // OSPL: slot sl of a 256-byte staging row is 16 bytes of the fp16 segment (sl < 8: features 8 sl ..), of the lo8 segment
        // (8 <= sl < 12: features 16 (sl - 8) ..) or of the hi8 segment (sl >= 12) of the wave's 64 features
        [[maybe_unused]] const int64_t spl_colb = sl < 8 ? 2 * (int64_t)(cn0 + wc * 64) + 16 * sl
                                                         : (int64_t)(sl < 12 ? 2 : 3) * a.n_logical + cn0 + wc * 64 + 16 * (sl & 3);
        [[maybe_unused]] const bool spl_col_ok = sl < 8 ? cn0 + wc * 64 + 8 * sl + 8 <= a.N : cn0 + wc * 64 + 16 * (sl & 3) + 16 <= a.N;
        // fp32 residual, fetched in the SAME row-contiguous layout the stores use (1 KiB = 4 rows x 256 B per
        // instruction, whole lines) RD passes ahead; loading it in the accumulator layout (16 rows x 32-byte pieces
        // per instruction) costs ~8 us of address processing per tile.
        constexpr int RD = 2;                                   // residual passes in flight
        // what one lane holds of a residual row: float4 in the fp32 layout, eight halves (u32x4) in the 16-bit layout
        using RV = typename std::conditional<FAST16, u32x4, float4>::type;
        constexpr int REL = FAST16 ? 8 : 4;                     // elements per lane and row
        [[maybe_unused]] RV rres[RD][4];
#define LOAD_RES(PS)                                                                                                         \
        if constexpr (HAS_RES) {                                                                                             \
            const int prow_ = ((PS) / (NPASS / 2)) * 128 + wr * 64 + ((PS) % (NPASS / 2)) * ROWS;                            \
            _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                                                  \
                const int64_t m = cm0 + prow_ + j * (64 / LPR) + rr;                                                         \
                rres[(PS) % RD][j] = (full || (m < a.M && ncol + REL <= a.N))                                                \
                    ? *reinterpret_cast<const RV*>(reinterpret_cast<const ST*>(a.R) + cz * a.sR + m * a.ldr + ncol) : RV{};  \
            }                                                                                                                \
        }
        _Pragma("unroll") for (int p = 0; p < RD; ++p) { LOAD_RES(p) }
        __builtin_amdgcn_sched_barrier(0);
        // ---- LNF: fetch b', s and the statistics of all eight row blocks this wave stores (published in the tile's last read segment)
        [[maybe_unused]] float lb[2][8], ls[2][8], lal[2][4], lnm[2][4];
        if constexpr (LNF) {
            const unsigned baddr = (unsigned)(size_t)(lptr_t)(bias_slot) + (unsigned)(ge * 32);
            const unsigned xr = (unsigned)(size_t)(lptr_t)(smem + 2 * kDbuf + wr * 4096 + 512) + r15e * 8;
            f32x4 b00, b01, b10, b11, s00, s01, s10, s11;
            f32x2 t00, t01, t02, t03, t10, t11, t12, t13;
            asm volatile("ds_read_b128 %0, %8\n\tds_read_b128 %1, %8 offset:16\n\tds_read_b128 %2, %8 offset:128\n\t"
                         "ds_read_b128 %3, %8 offset:144\n\tds_read_b128 %4, %8 offset:256\n\tds_read_b128 %5, %8 offset:272\n\t"
                         "ds_read_b128 %6, %8 offset:384\n\tds_read_b128 %7, %8 offset:400\n\ts_waitcnt lgkmcnt(0)"
                         : "=&v"(b00), "=&v"(b01), "=&v"(b10), "=&v"(b11), "=&v"(s00), "=&v"(s01), "=&v"(s10), "=&v"(s11)
                         : "v"(baddr) : "memory");
            asm volatile("ds_read_b64 %0, %8\n\tds_read_b64 %1, %8 offset:1024\n\tds_read_b64 %2, %8 offset:2048\n\t"
                         "ds_read_b64 %3, %8 offset:3072\n\tds_read_b64 %4, %8 offset:128\n\tds_read_b64 %5, %8 offset:1152\n\t"
                         "ds_read_b64 %6, %8 offset:2176\n\tds_read_b64 %7, %8 offset:3200\n\ts_waitcnt lgkmcnt(0)"
                         : "=&v"(t00), "=&v"(t01), "=&v"(t02), "=&v"(t03), "=&v"(t10), "=&v"(t11), "=&v"(t12), "=&v"(t13)
                         : "v"(xr) : "memory");
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                lb[0][q] = b00[q]; lb[0][4 + q] = b01[q]; lb[1][q] = b10[q]; lb[1][4 + q] = b11[q];
                ls[0][q] = s00[q]; ls[0][4 + q] = s01[q]; ls[1][q] = s10[q]; ls[1][4 + q] = s11[q];
            }
            lal[0][0] = t00.x; lnm[0][0] = t00.y; lal[0][1] = t01.x; lnm[0][1] = t01.y; lal[0][2] = t02.x; lnm[0][2] = t02.y; lal[0][3] = t03.x; lnm[0][3] = t03.y;
            lal[1][0] = t10.x; lnm[1][0] = t10.y; lal[1][1] = t11.x; lnm[1][1] = t11.y; lal[1][2] = t12.x; lnm[1][2] = t12.y; lal[1][3] = t13.x; lnm[1][3] = t13.y;
        }
        __builtin_amdgcn_sched_barrier(0);

        // ---- request the next tile's first K-tile (and bias) before this tile's epilogue ------------------------------
        if (more) {
            const bool streamed = stream;
            stream = false;
            adopt(nx);
            if (!streamed) { ISSUE_PROLOGUE() }   // otherwise already in flight since the last iteration
            ISSUE_BIAS()
        }
        __builtin_amdgcn_sched_barrier(0);
        // ---- LNF: row scale, mean correction and b' applied to all 128 accumulators in one pass of their own (under the next tile's
        //      operand requests): inside the store passes the 48 extra live values spilled, and a spill reload there is a vmcnt(0)
        if constexpr (LNF) {
#pragma unroll
            for (int mh = 0; mh < 2; ++mh)
#pragma unroll
                for (int mi = 0; mi < 4; ++mi)
#pragma unroll
                    for (int nh = 0; nh < 2; ++nh)
#pragma unroll
                        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
                            for (int jp = 0; jp < 2; ++jp) {   // two outputs per v_pk_fma_f32
                                const int q = ni * 4 + jp * 2;
                                const f32x2 al2 = {lal[mh][mi], lal[mh][mi]}, nm2 = {lnm[mh][mi], lnm[mh][mi]};
                                const f32x2 t2 = __builtin_elementwise_fma(nm2, f32x2{ls[nh][q], ls[nh][q + 1]}, f32x2{lb[nh][q], lb[nh][q + 1]});
                                const f32x2 v2 = __builtin_elementwise_fma(al2, f32x2{acc[mh][mi][nh][ni][jp * 2], acc[mh][mi][nh][ni][jp * 2 + 1]}, t2);
                                acc[mh][mi][nh][ni][jp * 2] = v2.x;
                                acc[mh][mi][nh][ni][jp * 2 + 1] = v2.y;
                            }
#pragma unroll
            for (int mh = 0; mh < 2; ++mh)
#pragma unroll
                for (int mi = 0; mi < 4; ++mi)
#pragma unroll
                    for (int nh = 0; nh < 2; ++nh) { asm volatile("" : "+v"(acc[mh][mi][nh][0]), "+v"(acc[mh][mi][nh][1])); }
            __builtin_amdgcn_sched_barrier(0);
        }
        STAMP(6)

        // ---- epilogue: activation, pack, then ROW-CONTIGUOUS stores through a private LDS staging tile ---------------
        // In the accumulator a lane owns 8 features of 16 different rows, so a direct store instruction would touch 16
        // rows x 64 B (store issue, not HBM, then bounds the tile: ~13 B/clk/CU measured).  Each wave instead transposes
        // 32 rows x 128 B (16-bit out) or 16 rows x 256 B (fp32 out) at a time through 4 KiB of LDS (half-tile A1 of dbuf 1,
        // which the next tile's prologue does not touch, and 16 KiB of spare LDS) and stores 1 KiB per instruction as 8 whole 128-byte lines.
        // The LDS ops are inline asm: the compiler's wait-count pass would drain the prologue DMA in front of them.
        // Interior tiles issue a FIXED number of store instructions (16 or 32): the next tile waits with a counted vmcnt.
        {
            char* const stg = wave < 4 ? smem + kDbuf + kHalf + wave * 4096 : smem + 2 * kDbuf + 8 * 1024 + (wave - 4) * 4096;
            const unsigned stg_addr = (unsigned)(size_t)(lptr_t)(stg);
            // 16-bit epilogue: tile-invariant lane addresses of the staging tile (write: row r15 (+16 by immediate), slot nh*4 + g;
            // read: row rr (+8 j by immediate), slot sl - both XOR-swizzled with the row) and of the output row (bytes)
            [[maybe_unused]] const unsigned st_w0 = stg_addr + r15e * 128 + (((0 + ge) ^ (r15e & 7)) << 4);
            [[maybe_unused]] const unsigned st_w1 = stg_addr + r15e * 128 + (((4 + ge) ^ (r15e & 7)) << 4);
            [[maybe_unused]] const unsigned st_rd = stg_addr + rr * 128 + ((sl ^ (rr & 7)) << 4);
            [[maybe_unused]] const unsigned st_voff = (unsigned)((rr * a.ldc + wc * 64 + sl * 8) * (int64_t)sizeof(OT));
            [[maybe_unused]] const char* const c_tile = reinterpret_cast<const char*>(a.C) + (cz * a.sC + (cm0 + wr * 64) * a.ldc + cn0) * (int64_t)sizeof(OT);
            [[maybe_unused]] const int64_t c_row8 = a.ldc * 8 * (int64_t)sizeof(OT);
            u32x4 wd[4];                                            // one pass of packed outputs: [sub][nh] (16-bit) / [nh][half] (fp32)
            // activation + pack of pass `ps` into wd (pure VALU: overlaps the LDS round trip of the previous pass)
#define PRODUCE(PS)                                                                                                          \
            _Pragma("unroll") for (int sub = 0; sub < ROWS / 16; ++sub) {                                                    \
                const int mi = ((PS) % (NPASS / 2)) * (ROWS / 16) + sub;                                                     \
                _Pragma("unroll") for (int nh = 0; nh < 2; ++nh) {                                                           \
                    float v[8];                                                                                              \
                    _Pragma("unroll") for (int ni = 0; ni < 2; ++ni)                                                         \
                        _Pragma("unroll") for (int jj = 0; jj < 4; ++jj) v[ni * 4 + jj] = acc[(PS) / (NPASS / 2)][mi][nh][ni][jj]; \
                    const int act_ = ACT >= 0 ? ACT : a.act;                                                                 \
                    if (act_ == CIR_ACT_GELU) {                                                                              \
                        if constexpr (MIX) gelu_erf_as8(v);   /* erf to 1.5e-7: the split8 path keeps ~16 bits */                 \
                        else gelu_erf8(v);   /* four interleaved packed chains */                                            \
                    } else if (act_ == CIR_ACT_RELU) {                                                                       \
                        _Pragma("unroll") for (int q = 0; q < 8; ++q) v[q] = fmaxf(v[q], 0.f);                               \
                    }                                                                                                        \
                    if constexpr (OSPL) {   /* [nh][0] = 8 halves, [nh][1] = {lo8 x 8, hi8 x 8} */                           \
                        const float qa_[4] = {v[0], v[1], v[2], v[3]}, qb_[4] = {v[4], v[5], v[6], v[7]};                     \
                        const Split4 sa_ = split8_x4(qa_), sb_ = split8_x4(qb_);                                             \
                        wd[nh * 2 + 0] = u32x4{sa_.h01, sa_.h23, sb_.h01, sb_.h23};                                          \
                        wd[nh * 2 + 1] = u32x4{sa_.lo8, sb_.lo8, sa_.hi8, sb_.hi8};                                          \
                    } else if constexpr (OUT_F32) {                                                                          \
                        wd[nh * 2 + 0] = __builtin_bit_cast(u32x4, f32x4{v[0], v[1], v[2], v[3]});                           \
                        wd[nh * 2 + 1] = __builtin_bit_cast(u32x4, f32x4{v[4], v[5], v[6], v[7]});                           \
                    } else {                                                                                                 \
                        u32x4 o;                                                                                             \
                        o.x = pack2<OT>(v[0], v[1]); o.y = pack2<OT>(v[2], v[3]); o.z = pack2<OT>(v[4], v[5]); o.w = pack2<OT>(v[6], v[7]); \
                        wd[sub * 2 + nh] = o;                                                                                \
                    }                                                                                                        \
                }                                                                                                            \
            }
            // registers -> LDS staging tile: slot = 16-byte chunk of the row, XOR-swizzled with the row
#define WRITE_STAGE()                                                                                                        \
            if constexpr (OSPL) {   /* row r15: fp16 features at bytes 0-127 (slot nh*4 + g), lo8 at 128-191, hi8 at 192-255 (8-byte pieces) */ \
                _Pragma("unroll") for (int nh = 0; nh < 2; ++nh) {                                                           \
                    const unsigned rb_ = stg_addr + r15 * ROWB;                                                              \
                    const unsigned ah_ = rb_ + (((nh * 4 + g) ^ (r15 & 7)) << 4);                                            \
                    const unsigned al_ = rb_ + (((8 + nh * 2 + (g >> 1)) ^ (r15 & 7)) << 4) + (g & 1) * 8;                   \
                    const unsigned a8_ = rb_ + (((12 + nh * 2 + (g >> 1)) ^ (r15 & 7)) << 4) + (g & 1) * 8;                  \
                    const u32x2 lo_ = {wd[nh * 2 + 1].x, wd[nh * 2 + 1].y}, hi_ = {wd[nh * 2 + 1].z, wd[nh * 2 + 1].w};      \
                    asm volatile("ds_write_b128 %0, %3\n\tds_write_b64 %1, %4\n\tds_write_b64 %2, %5"                        \
                                 :: "v"(ah_), "v"(al_), "v"(a8_), "v"(wd[nh * 2 + 0]), "v"(lo_), "v"(hi_) : "memory");       \
                }                                                                                                            \
            } else if constexpr (OUT_F32) {                                                                                  \
                _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                                              \
                    const int slot = (i >> 1) * 8 + g * 2 + (i & 1);                                                         \
                    const unsigned ad = stg_addr + r15 * ROWB + ((slot ^ (r15 & 7)) << 4);                                   \
                    asm volatile("ds_write_b128 %0, %1" :: "v"(ad), "v"(wd[i]) : "memory");                                  \
                }                                                                                                            \
            } else {   /* 16-bit rows: two lane addresses (weight half nh), the 16-row block `sub` is an immediate */        \
                asm volatile("ds_write_b128 %0, %2\n\tds_write_b128 %1, %3\n\tds_write_b128 %0, %4 offset:2048\n\t"        \
                             "ds_write_b128 %1, %5 offset:2048"                                                              \
                             :: "v"(st_w0), "v"(st_w1), "v"(wd[0]), "v"(wd[1]), "v"(wd[2]), "v"(wd[3]) : "memory");          \
            }
            PRODUCE(0)
            WRITE_STAGE()
#pragma unroll
            for (int ps = 0; ps < NPASS; ++ps) {
                // LDS -> registers, lanes along the row: instruction j covers 64/LPR rows x ROWB bytes = 1 KiB
                u32x4 d0, d1, d2, d3;
                if constexpr (OUT_F32) {
                    unsigned ra[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int lrow = j * (64 / LPR) + rr;
                        ra[j] = stg_addr + lrow * ROWB + ((sl ^ (lrow & 7)) << 4);
                    }
                    asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %5\n\tds_read_b128 %2, %6\n\tds_read_b128 %3, %7"
                                 : "=&v"(d0), "=&v"(d1), "=&v"(d2), "=&v"(d3)
                                 : "v"(ra[0]), "v"(ra[1]), "v"(ra[2]), "v"(ra[3]) : "memory");
                } else {   // 8 lanes per 128-byte row: row j*8 + rr, so (row & 7) == rr and j is an immediate
                    asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %4 offset:1024\n\tds_read_b128 %2, %4 offset:2048\n\t"
                                 "ds_read_b128 %3, %4 offset:3072"
                                 : "=&v"(d0), "=&v"(d1), "=&v"(d2), "=&v"(d3) : "v"(st_rd) : "memory");
                }
                if (ps + 1 < NPASS) { PRODUCE(ps + 1) }             // next pass's VALU work under the LDS latency
                asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) :: "memory");
                const int prow = (ps / (NPASS / 2)) * 128 + wr * 64 + (ps % (NPASS / 2)) * ROWS;   // first row of this pass in the tile
                u32x4 dd[4] = {d0, d1, d2, d3};
                if constexpr (HAS_RES && FAST16) {
                    typedef __attribute__((ext_vector_type(8))) ST st8;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const st8 cv = __builtin_bit_cast(st8, dd[j]), rv = __builtin_bit_cast(st8, rres[ps % RD][j]);
                        u32x4 o;
                        o[0] = pack2<ST>((float)cv[0] + (float)rv[0], (float)cv[1] + (float)rv[1]);
                        o[1] = pack2<ST>((float)cv[2] + (float)rv[2], (float)cv[3] + (float)rv[3]);
                        o[2] = pack2<ST>((float)cv[4] + (float)rv[4], (float)cv[5] + (float)rv[5]);
                        o[3] = pack2<ST>((float)cv[6] + (float)rv[6], (float)cv[7] + (float)rv[7]);
                        dd[j] = o;
                    }
                } else if constexpr (HAS_RES) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        f32x4 o = __builtin_bit_cast(f32x4, dd[j]);
                        const float4 r = rres[ps % RD][j];
                        o += f32x4{r.x, r.y, r.z, r.w};
                        dd[j] = __builtin_bit_cast(u32x4, o);
                    }
                }
                if (!OUT_F32 && full) {   // (also with a residual: +1-2.5 % on proj / fc2; fetching the residual the same way - scalar row base + lane offset - measured 3-8 % slower than the per-lane addresses of LOAD_RES and was dropped)
                    // interior tile: wave-uniform row base (SGPR pair, scalar adds) + the lane's tile-invariant 32-bit byte offset
                    // (one statement, closed by wait states: a store wider than 64 bits reads its data registers up to two cycles after
                    //  issue, and the hazard recogniser does not see inline asm - the next VALU write could land in dd first)
                    const char* const sb0 = c_tile + ((ps / (NPASS / 2)) * 16 + (ps % (NPASS / 2)) * (ROWS / 8)) * c_row8;
                    const char* const sb1 = sb0 + c_row8;
                    const char* const sb2 = sb1 + c_row8;
                    const char* const sb3 = sb2 + c_row8;
                    asm volatile("global_store_dwordx4 %0, %1, %5 nt\n\tglobal_store_dwordx4 %0, %2, %6 nt\n\t"
                                 "global_store_dwordx4 %0, %3, %7 nt\n\tglobal_store_dwordx4 %0, %4, %8 nt\n\ts_nop 1"
                                 :: "v"(st_voff), "v"(dd[0]), "v"(dd[1]), "v"(dd[2]), "v"(dd[3]), "s"(sb0), "s"(sb1), "s"(sb2), "s"(sb3) : "memory");
                } else
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    int64_t m = cm0 + prow + j * (64 / LPR) + rr;
#ifdef CIR_GEMM_STAMPS
                    if (a.dbg & 0x200) m = blockIdx.x * 16 + (m & 15);   // experiment: every store of a workgroup into the same 16 rows (no HBM write stream)
                    if ((a.dbg & 0x100) && dd[j][0] != 0x12345u) continue;   // experiment: no global stores (LDS + VALU part of the epilogue alone)
#endif
                    if constexpr (OSPL) {
                        if (full || (m < a.M && spl_col_ok)) {
                            u32x4* cp = reinterpret_cast<u32x4*>(reinterpret_cast<char*>(a.C) + cz * a.sC + m * a.ldc + spl_colb);   // ldc / sC in bytes
                            STORE_C(cp, dd[j])
                        }
                    } else
                    if (full || (m < a.M && ncol + (OUT_F32 ? 4 : 8) <= a.N)) {
                        u32x4* cp;
                        if constexpr (OUT_F32) cp = reinterpret_cast<u32x4*>(reinterpret_cast<float*>(a.C) + cz * a.sC + m * a.ldc + ncol);
                        else cp = reinterpret_cast<u32x4*>(reinterpret_cast<OT*>(a.C) + cz * a.sC + m * a.ldc + ncol);
                        STORE_C(cp, dd[j])
                    }
                }
                if (ps + RD < NPASS) { LOAD_RES(ps + RD) }
                if (ps + 1 < NPASS) { WRITE_STAGE() }               // the reads of this pass have retired
            }
#undef PRODUCE
#undef LOAD_RES
#undef WRITE_STAGE
        }
        pending_stores = full ? (OUT_F32 ? 32 : 16) : 0;   // (residual loads also follow the prologue: waiting for fewer is safe)
        STAMP(7)
        ++tile_no;
        if (!more) break;
        t = tn;
    }
    // the last tile's dead-slot refills are LDS-DMA writes: retire them before the workgroup gives its LDS back
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    STAMP_BLK(2)
#undef ISSUE_BIAS
#undef ISSUE_A
#undef ISSUE_B
#undef ISSUE_A1
#undef LDS_DMA
#undef PIECE_A
#undef PIECE_B
#undef ISSUE_B1
#undef ISSUE_PROLOGUE
#undef READ_A
#undef READ_B
#undef MM
#undef MM4
#undef PLACE
#undef SYNC
#undef COMPUTE
#undef WAIT_ABB
#undef WAIT_A1
#undef WAIT_N
#undef WAIT_PRO
}

#ifdef CIR_GEMM_STAMPS
}  // namespace cir
extern "C" int cir_debug_read_stamps(unsigned long long* host) {
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(cir::g_stamps), sizeof(cir::g_stamps));
}
extern "C" int cir_debug_read_pairs(unsigned long long* host) {
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(cir::g_pair), sizeof(cir::g_pair));
}
extern "C" int cir_debug_read_blocks(unsigned long long* host) {
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(cir::g_blk), sizeof(cir::g_blk));
}
namespace cir {
#endif

static int persistent_grid() {
    static int cus = 0;   // lazily read device constant (number of CUs); the only state this library keeps
    if (cus == 0) {
        int dev = 0, n = 0;
        if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n > 0) cus = n;
        else cus = 256;
    }
    return cus;
}

void launch_gemm256(const GemmArgs& a_in, int in_dtype, int out_kind, hipStream_t s) {
    // out_kind: 0 = 16-bit C in the operand type, 1 = fp32 C (and R), 2 = fp16 residual-stream C (and R);
    // 3 = LayerNorm folded in (fp16 rows in, fp16 C; a.colsum / a.ln_eps set; act NONE or GELU)
    GemmArgs a = a_in;
    a.dbg = 0;
#ifdef CIR_GEMM_STAMPS
    if (const char* e = getenv("CIR_DBG")) a.dbg = (int)strtol(e, nullptr, 0);
#endif
    a.tiles_m = (int)((a.M + T256 - 1) / T256);
    a.tiles_n = (a.N + T256 - 1) / T256;
    const int64_t ntiles = (int64_t)a.tiles_m * a.tiles_n * a.batch;
    // Raster: an XCD's 32 CUs sweep K together on 32 tiles laid out as gw columns x 32/gw rows, so per K-step its L2
    // pulls gw weight slices + 32/gw activation slices from the fabric for 32 x 2 delivered: pick the gw that minimises
    // gw + 32/gw (~6), preferring a divisor of tiles_n (ragged last groups cost more than they save).  Measured with the
    // operand stream alone (tools/dma_probe.hip, 318464 x 3072 x 768): 50 / 55 / 68 / 61 GB/s per CU at gw 1 / 4 / 6 / 12;
    // in the GEMM: 1022 -> 1080 TF/s from gw 4 to 6.
    int gw = 1;
    float best = 1e30f;
    for (int d = 1; d <= a.tiles_n && d <= 12; ++d) {
        const float cost = d + 32.0f / d + (a.tiles_n % d ? 1.5f : 0.f);
        if (cost < best) { best = cost; gw = d; }
    }
    if (const int v = g_tune[CIR_TUNE_GEMM_GROUP_W]; v > 0) gw = v < a.tiles_n ? v : a.tiles_n;
    a.group_w = gw;
    const int64_t g = persistent_grid();
    dim3 grid((unsigned)(ntiles < g ? ntiles : g)), block(512);
    const bool res = a.R != nullptr;
#define CIR_LAUNCH256(...) hipLaunchKernelGGL((gemm256_kernel<__VA_ARGS__>), grid, block, 0, s, a)
    if (out_kind == 3) {
        if (a.act == CIR_ACT_GELU) CIR_LAUNCH256(_Float16, false, false, float, CIR_ACT_GELU, true);
        else CIR_LAUNCH256(_Float16, false, false, float, CIR_ACT_NONE, true);
        return;
    }
    // the activation is a compile-time constant where the path runs at scale: operand-type C with none / GELU (QKV, K|V,
    // cross-Q / fc1), residual-stream C with none (proj, fc2, merge); every other combination reads a.act at run time
#define CIR_DISPATCH256(T)                                                                                                     \
    if (out_kind == 1) {                                                                                                       \
        if (a.act == CIR_ACT_NONE) { if (res) CIR_LAUNCH256(T, true, true, float, CIR_ACT_NONE); else CIR_LAUNCH256(T, true, false, float, CIR_ACT_NONE); } \
        else { if (res) CIR_LAUNCH256(T, true, true); else CIR_LAUNCH256(T, true, false); }                                    \
    }                                                                                                                          \
    else if (out_kind == 2) {                                                                                                  \
        if (a.act == CIR_ACT_NONE) { if (res) CIR_LAUNCH256(T, false, true, _Float16, CIR_ACT_NONE); else CIR_LAUNCH256(T, false, false, _Float16, CIR_ACT_NONE); } \
        else { if (res) CIR_LAUNCH256(T, false, true, _Float16); else CIR_LAUNCH256(T, false, false, _Float16); }              \
    } else if (a.act == CIR_ACT_NONE) CIR_LAUNCH256(T, false, false, float, CIR_ACT_NONE);                                    \
    else if (a.act == CIR_ACT_GELU) CIR_LAUNCH256(T, false, false, float, CIR_ACT_GELU);                                      \
    else CIR_LAUNCH256(T, false, false);
    if (in_dtype == CIR_BF16) { CIR_DISPATCH256(__bf16) } else { CIR_DISPATCH256(_Float16) }
#undef CIR_DISPATCH256
#undef CIR_LAUNCH256
}

void launch_gemm256_split8(const GemmArgs& a_in, int out_split, hipStream_t s) {
    GemmArgs a = a_in;
    a.dbg = 0;
    a.tiles_m = (int)((a.M + T256 - 1) / T256);
    a.tiles_n = (a.N + T256 - 1) / T256;
    const int64_t ntiles = (int64_t)a.tiles_m * a.tiles_n * a.batch;
    int gw = 1;
    float best = 1e30f;
    for (int d = 1; d <= a.tiles_n && d <= 12; ++d) {
        const float cost = d + 32.0f / d + (a.tiles_n % d ? 1.5f : 0.f);
        if (cost < best) { best = cost; gw = d; }
    }
    if (const int v = g_tune[CIR_TUNE_GEMM_GROUP_W]; v > 0) gw = v < a.tiles_n ? v : a.tiles_n;
    a.group_w = gw;
    const int64_t g = persistent_grid();
    dim3 grid((unsigned)(ntiles < g ? ntiles : g)), block(512);
#define CIR_LAUNCH256(...) hipLaunchKernelGGL((gemm256_kernel<__VA_ARGS__>), grid, block, 0, s, a)
    if (out_split) {
        if (a.act == CIR_ACT_GELU) CIR_LAUNCH256(_Float16, true, false, float, CIR_ACT_GELU, false, true, true);
        else CIR_LAUNCH256(_Float16, true, false, float, -1, false, true, true);
    } else if (a.R != nullptr) {
        CIR_LAUNCH256(_Float16, true, true, float, CIR_ACT_NONE, false, true, false);      // (the dispatcher admits a residual only without an activation)
    } else if (a.act == CIR_ACT_NONE) {
        CIR_LAUNCH256(_Float16, true, false, float, CIR_ACT_NONE, false, true, false);
    } else {
        CIR_LAUNCH256(_Float16, true, false, float, -1, false, true, false);
    }
#undef CIR_LAUNCH256
}

}  // namespace cir
