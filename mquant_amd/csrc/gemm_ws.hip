// gemm_ws.hip -- wave-specialised W4A8 GEMM for the shapes whose M x N output cannot feed 256 CUs
// with 256 x 256 tiles (every Linear of the prefill except gate_up / down_proj).
//
// Measurements that shaped it (tools/probes/l2_read_rate.hip, l2_row_stride.hip):
//   * a CU pulls 40-50 B/clk out of its XCD's L2 when every wave instruction fetches 1 KiB of
//     CONTIGUOUS bytes, but only ~14.5 B/clk when the instruction gathers rows of a row-major
//     matrix (8 x 128 B, 16 x 64 B, 4 x 256 B alike).  The activation operand therefore arrives
//     in the TILED layout ([M/16][K/64] pieces of 1 KiB in MFMA-fragment order, written that way
//     by the quantizer kernels): every LDS-DMA of this kernel is one contiguous KiB;
//   * with one wave per SIMD nothing overlaps: LDS-DMA issue (60-180 cycles each), ds_read
//     latency and the MFMA chain of a k-step add up (0.45 us per 128-byte k-step for a 96 x 128
//     tile whose MFMAs need 0.16 us).  Here the roles are split: NL LOADER waves issue every
//     LDS-DMA and own the vmcnt bookkeeping, the MATH waves only read fragments and issue MFMAs,
//     with the fragments of the next half k-step already on their way while the current ones
//     feed the matrix core.  A SIMD hosts one wave of each kind;
//   * ~4 us of every launch used to be epilogue latency (parameter fetches in front of the
//     stores).  The math waves fetch the per-channel / per-row parameters into LDS while the
//     loaders fill the first stage; after the k-loop ALL waves dequantise and store.
//
// Synchronisation: one s_barrier per k-step, B(0), B(1), ... B(nk):
//   B(0):     stage 0 has landed (loaders waited with a counted vmcnt); parameters are in LDS
//   B(it+1):  stage it+1 has landed; every math wave has finished step it-1, so the slot of stage
//             it-1 is free and the loaders refill it with stage it-1+S
//   math, step it:  B(it+1) | read (it, kt1) | MFMA (it, kt0) | read (it+1, kt0) | MFMA (it, kt1)
#include <type_traits>

#include "gemm_common.h"

namespace mq {

// MF: matrix instruction of the math waves -- 0: V_MFMA_I32_32X32X32_I8 (rounds 2-4), 1: V_MFMA_I32_16X16X64_I8 (round 5: under the
// package power limit the 16x16x64 form sustains 15 % more int8 ops per watt on random operands, 3.82 against 3.32 POP/s in the
// register-only burn of tools/probes/clock_recon.hip; its accumulator tile is a quarter of the 32x32 one, half the accumulator
// register traffic per MAC).
// WG (MF == 1 only), a bit set: 1 = group-wise WEIGHT scales (--w_groupsize g; GemmArgs::sw_groups, reference
// gptq/gptq_utils.py:263-273), 2 = group-wise ACTIVATION scales (--a_groupsize g; GemmArgs::sx_groups, quant_utils.py:181-203), 3 = both
// with the same g (64 or a multiple of 128).  After the k-tiles of a group the math waves fold
// (float(acc) * s_xg[row][group]) * s_wg[group][channel] into fp32 accumulators (ascending groups, one rounding per product and per
// sum -- oracle orc_gemm_wgroup / the restated arithmetic of tests/test_gpu_groupwise.py) and the epilogue receives the float bits
// and applies what is left (row scale, or s_w[n]).  The scales of a group are requested two groups ahead.
#ifndef MQ_WS_WG_ABL
#define MQ_WS_WG_ABL 0   // timing-only ablations of the group fold (wrong results): 4 no fold intervals (every interval plain), 8 no scale DMAs
#endif

template <int BM, int BN, int MW_M, int MW_N, int NL, int S, int W_BITS, int EPI, int MF = 0, int WG = 0, bool ACT = false, bool DIRECT = false>
__global__ __launch_bounds__((MW_M * MW_N + NL) * 64) void gemm_ws_kernel(GemmArgs p)
{
    // ACT: the instantiation with the consumer's activation in the store (GemmArgs::act; dispatch_ws_act)
    // DIRECT (round 6, ids 50-54): the plain 16-bit epilogue WITHOUT the LDS slab -- every math wave dequantises its own accumulators in
    // the MFMA D layout (a lane holds 4 consecutive channels of one row per 16 x 16 tile), packs them to 16 bits, V_PERMLANE16_SWAP
    // pairs the quads of two adjacent channel tiles so that a lane owns 8 consecutive channels, and stores 16 bytes: no barrier and no
    // LDS round trip behind the k-loop (the slab path: two barriers, park, re-read -- 1.6-4 us of every launch,
    // profiles/r5_ws_fixed_cost_timeline.txt).  Per-channel / per-row parameters wait in registers from before the k-loop.
    static_assert(!DIRECT || (MF == 1 && WG == 0 && !ACT && (EPI == EPI_F16 || EPI == EPI_BF16) && (BN / MW_N / 16) % 2 == 0), "direct epilogue: 16x16x64 tiles, 16-bit output, channel-tile pairs");
    static_assert(WG == 0 || MF == 1, "the weight-group fold lives in the 16x16x64 math loop");
    // Math waves: MW_M x MW_N wave tiles of (BM / MW_M) x (BN / MW_N), one or two per SIMD.  (A second group of
    // math waves working the K = 32 sub-steps of the other parity, and wide 96 x 64 / 64 x 64 wave tiles, were
    // built in round 2, exact, and no faster on any model shape -- DESIGN 4.1; they are not in the tree any more.)
    constexpr int NM = MW_M * MW_N;                 // math waves
    constexpr int NT = (NM + NL) * 64;              // threads
    constexpr int TM = BM / MW_M / 32;              // 32-row activation fragments per math wave
    constexpr int TN = BN / MW_N / 32;              // 32-channel weight fragments per math wave
    constexpr int A_PIECES = (BM / 16) * 2;         // 1 KiB pieces per stage (two k-tiles)
    constexpr int W_PIECES = (W_BITS == 4) ? (BN / 32) * 2 : (BN / 16) * 2;
    constexpr int PIECES = A_PIECES + W_PIECES;
    // Group scales (WG): every stage carries three 1 KiB scale blocks behind its pieces -- [128 weight scales | 128 activation scales]
    // of (A) the group that ended with the previous stage, (B) groups of 64: the group of this stage's first k-tile, (C) the group of
    // this stage's last k-tile -- one fp32 per lane and loader wave (NL = 4 quarters of a block), so the math waves find the scales of
    // a fold in the stage they are working on, as far ahead of their use as the operands themselves.
    constexpr int SC_BLOCKS = (WG && !(MQ_WS_WG_ABL & 8)) ? 3 : 0;
    constexpr int LPW = PIECES / NL + SC_BLOCKS;    // LDS-DMA instructions per loader wave per stage
    constexpr int A_BYTES = A_PIECES * 1024;
    constexpr int SC_OFF = PIECES * 1024;
    constexpr int STAGE = PIECES * 1024 + SC_BLOCKS * 1024;
    static_assert(WG == 0 || (NL == 4 && BN == 128 && BM <= 128), "scale blocks: four loader waves, 128 channels, at most 128 rows");
    constexpr int RING = S * STAGE;
    constexpr int PITCH = BN * 4 + 16;              // epilogue slab row pitch (bytes)
    static_assert(MF == 1 ? (BM % (MW_M * 16) == 0 && BN % (MW_N * 16) == 0) : (BM % (MW_M * 32) == 0 && BN % (MW_N * 32) == 0), "tile shape");
    static_assert(PIECES % NL == 0, "pieces must divide over the loader waves");
    static_assert(S >= 4 && S <= 8 && (S - 2) * LPW < 64, "ring depth (vmcnt is 6 bits)");
    constexpr int BODY = RING > BM * PITCH ? RING : BM * PITCH;   // the epilogue slab reuses the ring (and may exceed it)
    static_assert(NM * 64 >= BN && NM * 64 >= BM, "parameter prefetch: one thread per channel / row");

    extern __shared__ __attribute__((aligned(16))) char smem[];
    float *par_sw = reinterpret_cast<float *>(smem + BODY);      // [BN] weight scales
    float *par_bs = par_sw + BN;                                  // [BN] bias
    float *par_wz = par_bs + BN;                                  // [BN] w0 (split term)
    float *par_sx = par_wz + BN;                                  // [BM] activation scale of the row
    float *par_xz = par_sx + BM;                                  // [BM] x0 of the row
    float *par_w1 = par_xz + BM;                                  // [BN] w1 (second rank-1 term)
    float *par_x1 = par_w1 + BN;                                  // [BM] x1 of the row

    args_up_front(p);
    // Roles by wave: `wave` is the LOGICAL index (0 .. NM-1 math, NM .. NM+NL-1 loaders) = the hardware wave index.  The hardware
    // starts the waves of a workgroup one after the other (the first instruction of wave NM runs 0.45-0.5 us after wave 0's,
    // profiles/r5_ws_fixed_cost_timeline.txt); making the loaders the FIRST hardware waves (-DMQ_WS_LOADERS_FIRST) brings their
    // first LDS-DMA 0.2-0.3 us forward and was measured 0.6 % SLOWER over the bench's GEMM launches (6.51 against 6.47 ms,
    // profiles/r5_bench_ab_args_and_wave_order.txt): the math waves then start last and their parameter prefetch delays B(0).
    const int lane = threadIdx.x & 63;
#ifndef MQ_WS_LOADERS_FIRST
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
#else
    const int hwave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wave = hwave < NL ? NM + hwave : hwave - NL;
#endif
    const int tid = wave * 64 + lane;
#ifdef MQ_WS_TL
    // Timeline build (tools/gemm_timeline.py, profiles/r5_ws_fixed_cost_timeline.txt): 16 ints per workgroup in the
    // split-K workspace -- s_memtime at entry / first stage requested / first stage landed / B(0) / loop end / slab
    // parked / last store issued / stores drained, s_memrealtime (100 MHz, chip-wide) at entry and end, HW_ID, XCC_ID.
    int *tl = (p.partial && p.splits == 1) ? p.partial + (long)blockIdx.x * 16 : nullptr;
#define MQ_TL(who, i)                                                          \
    do {                                                                       \
        if (tl && tid == (who)) tl[i] = (int)__builtin_amdgcn_s_memtime();      \
    } while (0)
    if (tl && tid == 0) {
        unsigned hw, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        tl[0] = (int)__builtin_amdgcn_s_memtime();
        tl[1] = (int)__builtin_amdgcn_s_memrealtime();
        tl[2] = (int)hw;
        tl[3] = (int)xcc;
    }
#else
#define MQ_TL(who, i) do { } while (0)
#endif

    // ---- workgroup -> (split, bn, bm), XCD-aware and bijective (gemm_common.h) ---------------------
    int bm, bn, split, kb, nk;
    tile_of_block(p, bm, bn, split);
    k_range_of_split(p, split, kb, nk);
    const long m0 = (long)bm * BM;
    const long nt0 = (long)bn * (BN / 16);
    const long n0 = nt0 * 16;
    const long kps = p.K_pad >> 7;
    const long k_begin = kb;

    // Accumulators: with fewer than six 32x32 tiles per wave consecutive K = 32 sub-steps would form
    // a dependent MFMA chain only three or four instructions long and every VALU / LDS instruction
    // between two sub-steps would add to it (tools/probes/math_loop.hip: 520 -> 760 cycles per k-step);
    // even and odd sub-steps therefore accumulate into separate sets, added at the end (integers: exact).
    constexpr int NACC = (MF == 1) ? 1 : ((TM * TN < 6) ? 2 : 1);
    v16i acc[MF == 1 ? 1 : NACC][MF == 1 ? 1 : TN][MF == 1 ? 1 : TM];
    constexpr int TM16 = BM / MW_M / 16, TN16 = BN / MW_N / 16;    // MF == 1: 16-row / 16-channel fragments per math wave
    v4i acc16[MF == 1 ? TN16 : 1][MF == 1 ? TM16 : 1];

    if (wave >= NM) {
        // =========================== loader waves ===========================================
        const int lw = wave - NM;
#ifdef MQ_WS_TL
        const int tl_lentry = (int)__builtin_amdgcn_s_memtime();    // first instruction of a loader wave
#endif
        __builtin_amdgcn_s_setprio(2);               // a loader's few instructions go ahead of the math waves' streams
        const long KT = p.K_pad >> 6, MT = (p.M + 15) >> 4;
        // piece f of a stage: wave-uniform base (SGPRs) + lane * 16
        constexpr int LPP = PIECES / NL;             // pieces per loader wave and stage
        const char *src[LPP];
#pragma unroll
        for (int i = 0; i < LPP; ++i) {
            const int f = lw + i * NL;
            if (f < A_PIECES) {
                long mt = m0 / 16 + (f >> 1);
                if (mt >= MT) mt = MT - 1;
                src[i] = reinterpret_cast<const char *>(p.a) + (mt * KT + k_begin * 2 + (f & 1)) * 1024;
            } else if (W_BITS == 4) {
                const int g = f - A_PIECES;
                long ntp = nt0 / 2 + (g >> 1);
                if (ntp >= p.n_pairs) ntp = p.n_pairs - 1;
                if (ACT && p.act == MQ_ACT_SILU_MUL) {
                    // silu(gate) * up in the store (GemmArgs::act): slab columns [0, BN/2) = this n-block's BN / 2 gate channels,
                    // [BN/2, BN) = the same up channels (second half of the image)
                    constexpr int HS = BN / 64;                  // pair slots per half
                    const int sl = g >> 1;
                    const long hp = p.n_pairs >> 1;
                    long pi = (long)bn * HS + (sl % HS);
                    if (pi >= hp) pi = hp - 1;
                    ntp = sl >= HS ? hp + pi : pi;
                }
                src[i] = reinterpret_cast<const char *>(p.w) + ((ntp * kps + k_begin) * 2 + (g & 1)) * 1024;
            } else {
                const int g = f - A_PIECES;
                long nt = nt0 + (g >> 1);
                if (nt >= p.n_tiles) nt = p.n_tiles - 1;
                if (ACT && p.act == MQ_ACT_SILU_MUL) {
                    constexpr int HT = BN / 32;                  // 16-channel tile slots per half
                    const int sl = g >> 1;
                    const long ht = p.n_tiles >> 1;
                    long ti = (long)bn * HT + (sl % HT);
                    if (ti >= ht) ti = ht - 1;
                    nt = sl >= HT ? ht + ti : ti;
                }
                src[i] = reinterpret_cast<const char *>(p.w) + ((nt * kps + k_begin) * 2 + (g & 1)) * 1024;
            }
        }
        const unsigned lane_off = lane * 16;
        const unsigned lds0 = (unsigned)(size_t)(lds_void *)smem;
        // scale blocks: loader wave 0 / 1 = weight scales of channels n0 + lane (+ 64), wave 2 / 3 = activation scales of rows m0 + lane (+ 64)
        const char *sc_base = nullptr;
        long sc_stride = 0;                          // bytes from one group to the next
        unsigned sc_voff = 0;
        int sc_q = 0, sc_r = 0;                      // stage st = k-step st: group st / gsteps, k-step sc_r of it (groups of >= 128)
        const int sc_gsteps = (WG && p.group_k >= 128) ? (p.group_k >> 7) : 1;
        if (WG) {
            const bool use_w = lw < 2 ? (WG & 1) != 0 : (WG & 2) == 0;
            if (use_w) {                                          // weight scales (also the filler of waves 2 / 3 without activation groups)
                long nn = n0 + (lw & 1) * 64 + lane;
                if (nn >= p.N) nn = p.N - 1;
                sc_base = reinterpret_cast<const char *>(p.sw_groups);
                sc_stride = p.N * 4;
                sc_voff = (unsigned)(nn * 4);
            } else {                                              // activation scales (also the filler of waves 0 / 1 without weight groups)
                long row = m0 + (lw & 1) * 64 + lane;
                if (row >= p.M) row = p.M - 1;
                sc_base = reinterpret_cast<const char *>(p.sx_groups);
                sc_stride = 4;
                sc_voff = (unsigned)(row * p.n_groups * 4);
            }
        }
        auto issue = [&](int slot, int st) {
            const unsigned base = lds0 + slot * STAGE;
#pragma unroll
            for (int i = 0; i < LPP; ++i) dma16_s(src[i] + (long)st * 2048, lane_off, base + (lw + i * NL) * 1024);
            if (WG && SC_BLOCKS) {                   // stages are issued in ascending order: (sc_q, sc_r) follow st
                long ga, gb, gc;
                if (p.group_k == 64) {
                    ga = 2L * st - 1; gb = 2L * st; gc = 2L * st + 1;
                } else {
                    ga = sc_q - 1; gb = sc_q; gc = sc_q;
                    if (++sc_r == sc_gsteps) { sc_r = 0; ++sc_q; }
                }
                const long gl = p.n_groups - 1;
                ga = ga < 0 ? 0 : (ga > gl ? gl : ga);
                gb = gb > gl ? gl : gb;
                gc = gc > gl ? gl : gc;
                auto uni = [](const char *q) {      // wave-uniform by construction; tell the compiler (the DMA takes its base in SGPRs)
                    const unsigned long long v = (unsigned long long)q;
                    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
                    return reinterpret_cast<const char *>(((unsigned long long)hi << 32) | lo);
                };
                dma4_s(uni(sc_base + ga * sc_stride), sc_voff, base + SC_OFF + lw * 256);
                dma4_s(uni(sc_base + gb * sc_stride), sc_voff, base + SC_OFF + 1024 + lw * 256);
                dma4_s(uni(sc_base + gc * sc_stride), sc_voff, base + SC_OFF + 2048 + lw * 256);
            }
        };
        auto wait_younger = [&](int younger) {      // at most `younger` stages may still be in flight
#define MQ_WS_WAIT(k) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((k) * LPW < 64 ? (k) * LPW : 0) : "memory")
            switch (younger) {
            case 0: MQ_WS_WAIT(0); break;
            case 1: MQ_WS_WAIT(1); break;
            case 2: MQ_WS_WAIT(2); break;
            case 3: MQ_WS_WAIT(3); break;
            case 4: MQ_WS_WAIT(4); break;
            case 5: MQ_WS_WAIT(5); break;
            case 6: MQ_WS_WAIT(6); break;
            case 7: MQ_WS_WAIT(7); break;
            default: MQ_WS_WAIT(0); break;
            }
        };
        // Ring fill.  The first stages of a launch come from HBM and a CU keeps only so many misses in
        // flight: issuing the whole ring takes ~4000 cycles (profiles/r2_gemm_phase_stamps.txt), all of it in front of
        // the first MFMA if B(0) waits for it.  Only PRE0 stages go out before B(0); the rest of the ring
        // is filled at most two stages per k-step while the math waves already work (a stage is consumed
        // more slowly than it arrives, so the ring fills up behind them).
        constexpr int PRE0 = 3 < S ? 3 : S;            // measured against 1, 2 and the whole ring (profiles/r3_gemm_ring_prefill_depth.txt)
        const int pre0 = nk < PRE0 ? nk : PRE0;
#ifdef MQ_WS_TL
        asm volatile("" ::: "memory");
        const int tl_addr = (int)__builtin_amdgcn_s_memtime();      // piece addresses ready
        int tl_st0 = tl_addr;
#endif
#pragma unroll
        for (int s = 0; s < PRE0; ++s) {
            if (s < pre0) issue(s, s);
#ifdef MQ_WS_TL
            if (s == 0) tl_st0 = (int)__builtin_amdgcn_s_memtime();  // stage 0 requested
#endif
        }
#ifdef MQ_WS_TL
        const int tl_req = (int)__builtin_amdgcn_s_memtime();       // first stages requested (kept in registers: a store
#endif                                                              // here would sit in the counted vmcnt of the ring)
        wait_younger(pre0 - 1);                      // stage 0 landed
#ifdef MQ_WS_TL
        const int tl_land = (int)__builtin_amdgcn_s_memtime();
#endif
        __builtin_amdgcn_s_barrier();                // B(0)
        int last = pre0 - 1;                         // last stage issued
        int slot = pre0 == S ? 0 : pre0;             // slot of stage last + 1
        for (int it = 0; it < nk; ++it) {
            const int need = it + 1 < nk ? it + 1 : nk - 1;
            const int younger = last - need;
            if (younger == S - 3) MQ_WS_WAIT(S - 3);   // steady state: S-3 stages stay in flight across the barrier
            else wait_younger(younger > 0 ? younger : 0);
            __builtin_amdgcn_s_barrier();            // B(it+1): every math wave has finished step it-1
            // stages up to it-1+S fit the ring now
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                if (last + 1 < nk && last + 1 <= it - 1 + S) {
                    issue(slot, last + 1);
                    ++last;
                    if (++slot == S) slot = 0;
                }
            }
        }
        __builtin_amdgcn_s_setprio(0);
        if constexpr (DIRECT) return;                // (no slab epilogue to take part in)
#ifdef MQ_WS_TL
        if (tl && tid == NM * 64) {
            tl[4] = tl_req;
            tl[5] = tl_land;
            tl[14] = tl_addr;
            tl[15] = tl_st0;
            tl[3] = tl_lentry;
        }
#endif
    } else {
        // =========================== math waves =============================================
        const int wm = wave / MW_N, wn = wave % MW_N;
        // Epilogue parameters: straight-line loads at clamped indices now (no use before the k-loop ends,
        // so B(0) does not wait for these cold misses), selected and parked in LDS after the loop.
        float pr_sw = 0.0f, pr_bs = 0.0f, pr_wz = 0.0f, pr_sx = 0.0f, pr_xz = 0.0f, pr_w1 = 0.0f, pr_x1 = 0.0f;
        unsigned pr_rs = 0;
        if (EPI != EPI_I32 && !DIRECT) {
            const int tn = tid < BN ? tid : BN - 1, tm = tid < BM ? tid : BM - 1;
            long nc = n0 + tn < p.N ? n0 + tn : p.N - 1;
            if (ACT && p.act == MQ_ACT_SILU_MUL) {              // slab column tn: gate channel (first half) or the same up channel (second half)
                const long H = p.N >> 1, c = (long)bn * (BN / 2) + (tn & (BN / 2 - 1));
                nc = (c < H ? c : H - 1) + (tn >= BN / 2 ? H : 0);
            }
            const long mc = m0 + tm < p.M ? m0 + tm : p.M - 1;
            // (dummy: any readable fp32 array for the parameters that are absent -- with weight groups s_w itself is absent)
            const float *dm = (WG & 1) ? p.sw_groups : p.s_w;
            pr_sw = (WG & 1) ? 1.0f : p.s_w[nc];                         // weight groups: the scales were applied group by group, x 1.0 is exact
            pr_bs = (p.bias ? p.bias : dm)[p.bias ? nc : 0];
            pr_wz = (p.w0 ? p.w0 : dm)[p.w0 ? nc : 0];
            pr_sx = (p.sx_vec ? p.sx_vec : dm)[p.sx_vec ? mc : 0];
            pr_xz = (p.x0 ? p.x0 : dm)[p.x0 ? mc : 0];
            pr_w1 = (p.w1 ? p.w1 : dm)[p.w1 ? nc : 0];
            pr_x1 = (p.x1 ? p.x1 : dm)[p.x1 ? mc : 0];
            pr_rs = (p.row_sel ? p.row_sel : reinterpret_cast<const uint8_t *>(dm))[p.row_sel ? mc : 0];
        }
        // DIRECT: this lane's parameters in the D layout -- 4 consecutive channels per 16-channel tile (weight scale, bias, split-column
        // factor), one row per 16-row tile (activation scale, split-column value) -- requested here, consumed behind the k-loop
        constexpr int DN = DIRECT ? BN / MW_N / 16 : 1, DM = DIRECT ? BM / MW_M / 16 : 1;
        // (the 192 x 128 tile has no registers to spare for bias / split-column parameters during the k-loop: it fetches those behind it)
        constexpr bool D_EARLY = BM * BN < 192 * 128;
        v4f d_sw[DN], d_bs[DN], d_wz[DN];
        float d_sx[DM], d_xz[DM];
        auto direct_params = [&](bool scales, bool terms) {
            const float *bsp = p.bias ? p.bias : p.s_w, *wzp = p.w0 ? p.w0 : p.s_w;
#pragma unroll
            for (int i = 0; i < DN; ++i) {
                long nq = n0 + (wn * DN + i) * 16 + (lane >> 4) * 4;
                if (nq + 4 > p.N) nq = 0;                    // (N % 8 == 0, host-checked: a quad is inside or outside)
                if (scales) d_sw[i] = *reinterpret_cast<const v4f *>(p.s_w + nq);
                if (terms) {
                    d_bs[i] = *reinterpret_cast<const v4f *>(bsp + nq);
                    d_wz[i] = *reinterpret_cast<const v4f *>(wzp + nq);
                }
            }
#pragma unroll
            for (int j = 0; j < DM; ++j) {
                long mr = m0 + (wm * DM + j) * 16 + (lane & 15);
                if (mr >= p.M) mr = p.M - 1;
                if (scales) {
                    float sx = p.sx0;
                    if (p.sx_vec) sx = p.sx_vec[mr];
                    else if (p.row_sel && p.row_sel[mr]) sx = p.sx1;
                    d_sx[j] = (W_BITS == 4) ? sx * 0.0625f : sx;       // (int4 levels sit in the high nibble: exact power-of-two rescale)
                }
                if (terms) d_xz[j] = p.x0 ? p.x0[mr] : 0.0f;
            }
        };
        if constexpr (DIRECT) direct_params(true, D_EARLY);
        if constexpr (MF == 1) {
            // ---- V_MFMA_I32_16X16X64_I8: one interval per 64-wide k-tile t (two per stage, so every register-set index below
            // is a compile-time parity):   X[(t+1)&1] <- activations (t+1)   |  F[t&1] <- packed weights (t+2)
            //                               WU[(t+1)&1] <- unpack F[(t+1)&1] (= weights t+1, requested one interval ago)
            //                               TM16 x TN16 MFMAs on WU[t&1], X[t&1]
            // The LDS image is already in this instruction's fragment order (a 1 KiB piece = 16 rows x 64 k, lane l at byte 16 l).
#pragma unroll
            for (int i = 0; i < TN16; ++i)
#pragma unroll
                for (int j = 0; j < TM16; ++j) acc16[i][j] = v4i{0, 0, 0, 0};
            // Group scales: int32 -> fp32 WITHOUT v_cvt.  A group's first MFMA starts from C = 0x4B400000 (the bits of 1.5 * 2^23) in every
            // element; while |accumulator| <= 2^22 (groups of <= 256 k: 256 * 128 * 8 * 16 = 2^22) the accumulator's bits read as a float
            // are EXACTLY 12582912 + sum, and one packed subtraction per two elements returns float(sum) -- exact, like the conversion.
            // Larger groups (rare, one fold per >= 4 k-steps) convert with v_cvt_f32_i32 after the group's last k-step.
            // (W4: the levels sit in the high nibble, the accumulators are 16 x the sum -- the bias 1.5 * 2^19 has an ulp of 1/16, so the
            //  subtraction returns sum / 16 directly and the scales are used as they come)
            constexpr int MAGIC_BITS = (W_BITS == 4) ? 0x49400000 : 0x4B400000;
            constexpr float MAGIC_F = (W_BITS == 4) ? 786432.0f : 12582912.0f;
            const bool magic = WG && p.group_k <= 256;
            const int c0 = magic ? MAGIC_BITS : 0;
            const v4i cinit = v4i{c0, c0, c0, c0};
            if (WG) {
#pragma unroll
                for (int i = 0; i < TN16; ++i)
#pragma unroll
                    for (int j = 0; j < TM16; ++j) acc16[i][j] = cinit;
            }
            v4i X[2][TM16], WU[2][TN16], F[2][TN16];       // F: W4 uses the first two words
            auto rd_x = [&](int set, int slot, int kt) {
                const char *xs = smem + slot * STAGE + kt * 1024 + lane * 16;
#pragma unroll
                for (int j = 0; j < TM16; ++j) X[set][j] = *reinterpret_cast<const v4i *>(xs + (wm * TM16 + j) * 2048);
            };
            auto rd_w = [&](int set, int slot, int kt) {
                const char *ws = smem + slot * STAGE + A_BYTES + kt * 1024 + lane * 16;
#pragma unroll
                for (int i = 0; i < TN16; ++i) {
                    const int nt = wn * TN16 + i;
                    if (W_BITS == 4) {
                        const v2i pk = *reinterpret_cast<const v2i *>(ws + (nt >> 1) * 2048 + (nt & 1) * 8);
                        F[set][i][0] = pk[0];
                        F[set][i][1] = pk[1];
                    } else {
                        F[set][i] = *reinterpret_cast<const v4i *>(ws + nt * 2048);
                    }
                }
            };
            auto rd_x_part = [&](int set, int slot, int kt, int j) {
                const char *xs = smem + slot * STAGE + kt * 1024 + lane * 16;
                X[set][j] = *reinterpret_cast<const v4i *>(xs + (wm * TM16 + j) * 2048);
            };
            auto rd_w_part = [&](int set, int slot, int kt, int item) {      // W4: channel tiles 2 item, 2 item + 1 (one 16-byte read)
                const char *ws = smem + slot * STAGE + A_BYTES + kt * 1024 + lane * 16;
                if (W_BITS == 4) {
#pragma unroll
                    for (int i = 2 * item; i < 2 * item + 2 && i < TN16; ++i) {
                        const int nt = wn * TN16 + i;
                        const v2i pk = *reinterpret_cast<const v2i *>(ws + (nt >> 1) * 2048 + (nt & 1) * 8);
                        F[set][i][0] = pk[0];
                        F[set][i][1] = pk[1];
                    }
                } else {
                    F[set][item] = *reinterpret_cast<const v4i *>(ws + (wn * TN16 + item) * 2048);
                }
            };
            auto unpack16_part = [&](int set, int part) {                    // half a channel tile: 3 vector-ALU instructions
                const int i = part >> 1, h = part & 1;
                if (W_BITS == 4) {
                    const int w = F[set][i][h];
                    WU[set][i][2 * h] = (w << 4) & 0xF0F0F0F0;
                    WU[set][i][2 * h + 1] = w & 0xF0F0F0F0;
                } else if (h == 0) {
                    WU[set][i] = F[set][i];
                }
            };
            auto unpack16 = [&](int set) {
#pragma unroll
                for (int i = 0; i < TN16; ++i) {
                    if (W_BITS == 4) {
                        const int lo = F[set][i][0], hi = F[set][i][1];
                        WU[set][i][0] = (lo << 4) & 0xF0F0F0F0;
                        WU[set][i][1] = lo & 0xF0F0F0F0;
                        WU[set][i][2] = (hi << 4) & 0xF0F0F0F0;
                        WU[set][i][3] = hi & 0xF0F0F0F0;
                    } else {
                        WU[set][i] = F[set][i];
                    }
                }
            };
            // ---- group-wise scales (WG): fp32 accumulators; the scales of a fold come out of the stage's scale blocks ----
            constexpr bool WGW = (WG & 1) != 0, WGX = (WG & 2) != 0;
            v4f facc[WG ? TN16 : 1][WG ? TM16 : 1];
            v4f swc[WGW ? TN16 : 1];                                // weight scales of the group being folded, this lane's 4 channels per tile
            float sxc[WGX ? TM16 : 1];                              // activation scales of the group being folded, this lane's row per tile
            const int gsteps = WG ? (p.group_k >= 128 ? (p.group_k >> 7) : 1) : 1;   // k-steps per group (groups of 64: two groups per k-step)
            typedef __attribute__((address_space(3))) v4f lds_v4f;
            typedef __attribute__((address_space(3))) float lds_float;
            // per-lane offsets inside a scale block: 4 consecutive channels of the lane's quad / the lane's row
            const unsigned sc_lane_w = (unsigned)(size_t)(lds_void *)smem + SC_OFF + ((wn * TN16) * 16 + (lane >> 4) * 4) * 4;
            const unsigned sc_lane_x = (unsigned)(size_t)(lds_void *)smem + SC_OFF + 512 + ((wm * TM16) * 16 + (lane & 15)) * 4;
            unsigned sc_aw = 0, sc_ax = 0;
            auto sc_addr = [&](int slot, int block) {          // the addresses of a fold's reads, materialised BEFORE the interval's
                const unsigned u = slot * STAGE + block * 1024; // scheduling region (the reads then lead the interval's LDS queue)
                if (WGW) {
                    sc_aw = sc_lane_w + u;
                    asm volatile("" : "+v"(sc_aw));
                }
                if (WGX) {
                    sc_ax = sc_lane_x + u;
                    asm volatile("" : "+v"(sc_ax));
                }
            };
            auto rd_scales = [&]() {
                if (WGW) {
#pragma unroll
                    for (int i = 0; i < TN16; ++i) swc[i] = *(const lds_v4f *)(sc_aw + i * 64);
                }
                if (WGX) {
#pragma unroll
                    for (int j = 0; j < TM16; ++j) sxc[j] = *(const lds_float *)(sc_ax + j * 64);
                }
            };
            if (WG) {
#pragma unroll
                for (int i = 0; i < TN16; ++i)
#pragma unroll
                    for (int j = 0; j < TM16; ++j) facc[i][j] = v4f{0.f, 0.f, 0.f, 0.f};
            }
            // Issue order inside an interval: one fragment read and one share of the nibble unpack per MFMA gap.  The packed-weight
            // read goes out FIRST (LDS returns in order: the unpack of the NEXT interval then waits for the oldest read only, a
            // whole interval old), the activation reads behind it in the order the next interval's MFMAs consume them.
            constexpr int N_MFMA = TM16 * TN16;
            constexpr int N_DS = TM16 + ((W_BITS == 4) ? (TN16 + 1) / 2 : TN16);    // hipcc merges the two 8-byte reads of a channel-tile pair
            constexpr int N_VALU = (W_BITS == 4) ? 6 * TN16 : 0;
            constexpr int DS_PER_GAP = (N_DS + N_MFMA - 1) / N_MFMA, VALU_PER_GAP = (N_VALU + N_MFMA - 1) / N_MFMA;
            // fold of one 16 x 16 tile: 4 conversions, then packed fp32 multiplies / adds (one rounding per product and per sum)
            constexpr int FOLD_VALU = WG ? 2 + (WGX ? 2 : 0) + (WGW ? 2 : 0) + 2 : 0;
            // phase 1: the group sum as fp32 (needs no scales); phase 2: scale, add
            auto fold_sum = [&](auto magic_c, const v4i a, v2f_t &t0, v2f_t &t1) {
                if constexpr (decltype(magic_c)::value) {
                    const v2f_t mm = v2f_t{MAGIC_F, MAGIC_F};
                    t0 = v2f_t{__int_as_float(a[0]), __int_as_float(a[1])} - mm;
                    t1 = v2f_t{__int_as_float(a[2]), __int_as_float(a[3])} - mm;
                } else {
                    t0 = v2f_t{(float)a[0], (float)a[1]};
                    t1 = v2f_t{(float)a[2], (float)a[3]};
                    if (W_BITS == 4) {                   // exact: a power of two
                        t0 = t0 * v2f_t{0.0625f, 0.0625f};
                        t1 = t1 * v2f_t{0.0625f, 0.0625f};
                    }
                }
            };
            auto fold_add = [&](v2f_t t0, v2f_t t1, int i, int j) {
                if (WGX) {
                    const v2f_t sx2 = v2f_t{sxc[j], sxc[j]};
                    t0 = t0 * sx2;
                    t1 = t1 * sx2;
                }
                if (WGW) {
                    t0 = t0 * v2f_t{swc[i][0], swc[i][1]};
                    t1 = t1 * v2f_t{swc[i][2], swc[i][3]};
                }
                const v2f_t f0 = v2f_t{facc[i][j][0], facc[i][j][1]} + t0, f1 = v2f_t{facc[i][j][2], facc[i][j][3]} + t1;
                facc[i][j] = v4f{f0[0], f0[1], f1[0], f1[1]};
            };
            auto fold_tile = [&](auto magic_c, const v4i a, int i, int j) {
                v2f_t t0, t1;
                fold_sum(magic_c, a, t0, t1);
                fold_add(t0, t1, i, j);
            };
            // interval of k-tile (parity par): slotx / ktx = where activations t+1 live, slotw / ktw = where weights t+2 live.
            // BND (group scales, groups of <= 256): this k-tile is the FIRST of a group -- every tile's accumulators are folded with the
            // scales of the group that just ended, and the tile's first MFMA of the new group starts from the magic bias; the fold's
            // vector-ALU work sits in the MFMA gaps (6-8 packed instructions per 16 x 16 tile against the instruction's 8 passes)
            // instead of stopping the matrix pipe at every group boundary.
            // The scale reads go out first (LDS returns in order: the folds then wait for nothing younger) and a tile's phase 2 runs
            // LAG tiles behind its phase 1, so that the reads have two MFMAs' time to land.
            constexpr int LAG = N_MFMA > 2 ? 2 : 1;
            constexpr int N_SC = (WGW ? TN16 : 0) + (WGX ? TM16 : 0);
            auto interval = [&](auto bnd_c, int par, int slotx, int ktx, int slotw, int ktw, int slots = 0, int block = 0) {
                constexpr bool BND = decltype(bnd_c)::value;
                if (BND) sc_addr(slots, block);
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (!BND) {
                    rd_w(par, slotw, ktw);
                    rd_x(par ^ 1, slotx, ktx);
                    unpack16(par ^ 1);                       // weights t+1 (requested during the previous interval)
                }
                if constexpr (BND) {
                    // Hand-placed: one scheduling region per MFMA (the group-barrier pipeline does not survive this many vector-ALU
                    // instructions -- hipcc then issues the MFMAs back to back and the whole fold behind them).  Region m: one fragment
                    // read, one share of the nibble unpack, tile m's subtractions, its MFMA, and tile m - LAG's multiplies and adds.
                    v2f_t q0[N_MFMA], q1[N_MFMA];
                    rd_scales();
                    constexpr int NRW = (W_BITS == 4) ? (TN16 + 1) / 2 : TN16;      // weight-read items (W4: a channel-tile pair per item)
                    static_assert(NRW + TM16 <= N_MFMA && 2 * TN16 <= N_MFMA + 2, "one read and one unpack share per MFMA region");
#pragma unroll
                    for (int m = 0; m < N_MFMA + LAG; ++m) {
                        if (m < N_MFMA) {
                            if (m < NRW) rd_w_part(par, slotw, ktw, m);
                            else if (m - NRW < TM16) rd_x_part(par ^ 1, slotx, ktx, m - NRW);
                            if (m < 2 * TN16) unpack16_part(par ^ 1, m);
                            const int i = m / TM16, j = m % TM16;
                            fold_sum(std::true_type{}, acc16[i][j], q0[m], q1[m]);
                            acc16[i][j] = __builtin_amdgcn_mfma_i32_16x16x64_i8(WU[par][i], X[par][j], cinit, 0, 0, 0);
                        }
                        if (m >= LAG) fold_add(q0[m - LAG], q1[m - LAG], (m - LAG) / TM16, (m - LAG) % TM16);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                } else {
#pragma unroll
                    for (int i = 0; i < TN16; ++i)
#pragma unroll
                        for (int j = 0; j < TM16; ++j)
                            acc16[i][j] = __builtin_amdgcn_mfma_i32_16x16x64_i8(WU[par][i], X[par][j], acc16[i][j], 0, 0, 0);
#pragma unroll
                    for (int m = 0; m < N_MFMA; ++m) {
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                  // MFMA
                        if (N_VALU) __builtin_amdgcn_sched_group_barrier(0x002, VALU_PER_GAP, 0);
                        __builtin_amdgcn_sched_group_barrier(0x100, DS_PER_GAP, 0);        // DS reads
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            };
            int g_left = gsteps;                    // k-steps of the current group still to run
            const bool g64 = WG && p.group_k == 64;
            __builtin_amdgcn_s_barrier();            // B(0): stage 0 landed
            MQ_TL(0, 6);
            rd_w(0, 0, 0);
            rd_x(0, 0, 0);
            rd_w(1, 0, 1);
            unpack16(0);
            int cur = 0;
            constexpr std::false_type PLAIN{};
            constexpr std::true_type FIRST_OF_GROUP{};
            auto k_step = [&](int it) {
                __builtin_amdgcn_s_barrier();        // B(it+1): stage it+1 landed
                int nxt = cur + 1;
                if (nxt == S) nxt = 0;
                if (it + 1 >= nk) nxt = cur;         // last step: harmless re-reads of a live slot
                // k-tile 2 it: activations 2 it + 1 (this stage), weights 2 it + 2 (next stage)
                if (WG && magic && it > 0 && (g64 || g_left == gsteps) && !(MQ_WS_WG_ABL & 4)) {     // the previous k-tile completed a group
                    interval(FIRST_OF_GROUP, 0, cur, 1, nxt, 0, cur, 0);     // block A: the group that ended with the previous stage
                } else {
                    interval(PLAIN, 0, cur, 1, nxt, 0);
                }
                // k-tile 2 it + 1: activations 2 it + 2, weights 2 it + 3
                if (WG && g64 && !(MQ_WS_WG_ABL & 4)) {                     // groups of 64: one group per k-tile
                    interval(FIRST_OF_GROUP, 1, nxt, 0, nxt, 1, cur, 1);     // block B: the group of this stage's first k-tile
                } else {
                    interval(PLAIN, 1, nxt, 0, nxt, 1);
                }
                const int this_slot = cur;
                cur = nxt;
                if (WG && !g64 && --g_left == 0) {
                    g_left = gsteps;
                    if (!magic && it + 1 < nk) {     // groups of > 256: convert and fold here, the matrix pipe waits (one fold per >= 4 k-steps)
                        sc_addr(this_slot, 2);       // block C: the group of this stage's last k-tile
                        rd_scales();
#pragma unroll
                        for (int i = 0; i < TN16; ++i)
#pragma unroll
                            for (int j = 0; j < TM16; ++j) {
                                fold_tile(std::false_type{}, acc16[i][j], i, j);
                                acc16[i][j] = v4i{0, 0, 0, 0};
                            }
                    }
                }
            };
            if constexpr (WG != 0) {
#pragma clang loop unroll(disable)               // (no peeled first step: four interval variants are code enough)
                for (int it = 0; it < nk; ++it) k_step(it);
            } else {
                for (int it = 0; it < nk; ++it) k_step(it);
            }
            if (WG) {
                // the last group (a zero-padded last k-tile belongs to no group: its accumulators are 0, the clamped fold adds 0)
                sc_addr(cur, 2);                     // block C of the last stage
                rd_scales();
#pragma unroll
                for (int i = 0; i < TN16; ++i)
#pragma unroll
                    for (int j = 0; j < TM16; ++j) {
                        if (magic) fold_tile(std::true_type{}, acc16[i][j], i, j);
                        else fold_tile(std::false_type{}, acc16[i][j], i, j);
#pragma unroll
                        for (int r = 0; r < 4; ++r) acc16[i][j][r] = __float_as_int(facc[i][j][r]);
                    }
            }
            MQ_TL(0, 7);                             // k-loop done
            if constexpr (DIRECT) {
                const bool has_bias = p.bias != nullptr, has_x0 = p.x0 != nullptr, has_res = p.residual != nullptr;
                const bool gelu = p.act == MQ_ACT_QUICK_GELU;
                if constexpr (!D_EARLY) direct_params(false, true);
                const int g = lane >> 4;
                unsigned short *outp = reinterpret_cast<unsigned short *>(p.out);
                // residual (hidden + linear(x), torch rounds the Linear's output first): the 16 bytes this lane will store over, all
                // requested up front
                v4i res[TM16][TN16 / 2];
                if (has_res) {
                    const unsigned short *rp = reinterpret_cast<const unsigned short *>(p.residual);
#pragma unroll
                    for (int j = 0; j < TM16; ++j)
#pragma unroll
                        for (int ip = 0; ip < TN16 / 2; ++ip) {
                            long m = m0 + (wm * TM16 + j) * 16 + (lane & 15);
                            long n = n0 + (wn * TN16 + 2 * ip + (g & 1)) * 16 + (g >> 1) * 8;
                            if (m >= p.M) m = p.M - 1;
                            if (n + 8 > p.N) n = 0;
                            res[j][ip] = *reinterpret_cast<const v4i *>(rp + m * p.ldr + n);
                        }
                }
#pragma unroll
                for (int j = 0; j < TM16; ++j) {
                    const long m = m0 + (wm * TM16 + j) * 16 + (lane & 15);
                    const v2f_t sx2 = v2f_t{d_sx[j], d_sx[j]}, xz2 = v2f_t{d_xz[j], d_xz[j]};
#pragma unroll
                    for (int ip = 0; ip < TN16 / 2; ++ip) {
                        unsigned pk[2][2];
#pragma unroll
                        for (int h = 0; h < 2; ++h) {
                            const int i = 2 * ip + h;
                            const v4i a = acc16[i][j];
                            v2f_t v0 = v2f_t{(float)a[0], (float)a[1]}, v1 = v2f_t{(float)a[2], (float)a[3]};
                            v0 = v0 * sx2;
                            v1 = v1 * sx2;
                            v0 = v0 * v2f_t{d_sw[i][0], d_sw[i][1]};
                            v1 = v1 * v2f_t{d_sw[i][2], d_sw[i][3]};
                            if (has_bias) {
                                v0 = v0 + v2f_t{d_bs[i][0], d_bs[i][1]};
                                v1 = v1 + v2f_t{d_bs[i][2], d_bs[i][3]};
                            }
                            if (has_x0) {
                                const v2f_t p0 = xz2 * v2f_t{d_wz[i][0], d_wz[i][1]}, p1 = xz2 * v2f_t{d_wz[i][2], d_wz[i][3]};
                                v0 = v0 + p0;
                                v1 = v1 + p1;
                            }
                            if (gelu) {                  // GemmArgs::act == MQ_ACT_QUICK_GELU: the vision MLP's activation in fc1's store
                                constexpr int DT = (EPI == EPI_F16) ? MQ_F16 : MQ_BF16;
                                pk[h][0] = act_quick_gelu_pk<DT>(v0[0], v0[1]);
                                pk[h][1] = act_quick_gelu_pk<DT>(v1[0], v1[1]);
                            } else {
                                pk[h][0] = (EPI == EPI_F16) ? pack2_f16(v0[0], v0[1]) : pack2_bf16(v0[0], v0[1]);
                                pk[h][1] = (EPI == EPI_F16) ? pack2_f16(v1[0], v1[1]) : pack2_bf16(v1[0], v1[1]);
                            }
                        }
                        // odd rows of 16 lanes of tile i0's words <-> even rows of tile i1's: afterwards a lane holds 8 consecutive
                        // channels -- lane rows 0 / 2: channels 0..7 / 8..15 of tile i0, rows 1 / 3: the same of tile i1
                        const auto s0 = __builtin_amdgcn_permlane16_swap(pk[0][0], pk[1][0], false, false);
                        const auto s1 = __builtin_amdgcn_permlane16_swap(pk[0][1], pk[1][1], false, false);
                        v4i y = v4i{(int)s0[0], (int)s1[0], (int)s0[1], (int)s1[1]};
                        if (has_res) {
                            // cast(cast(y) + residual): the fp32 sum of two halves is exact unless one is below 2^-13 of the other, and then
                            // both forms return the larger one -- V_PK_ADD_F16 gives the same bits (as in the RoPE store); bf16 adds in fp32
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                if constexpr (EPI == EPI_F16) {
                                    typedef _Float16 h2 __attribute__((ext_vector_type(2)));
                                    h2 ya, rb;
                                    const int yi = y[e], ri = res[j][ip][e];
                                    __builtin_memcpy(&ya, &yi, 4);
                                    __builtin_memcpy(&rb, &ri, 4);
                                    const h2 sum = ya + rb;
                                    int si;
                                    __builtin_memcpy(&si, &sum, 4);
                                    y[e] = si;
                                } else {
                                    const unsigned yu = (unsigned)y[e], ru = (unsigned)res[j][ip][e];
                                    const float lo = __uint_as_float(yu << 16) + __uint_as_float(ru << 16);
                                    const float hi = __uint_as_float(yu & 0xffff0000u) + __uint_as_float(ru & 0xffff0000u);
                                    y[e] = (int)pack2_bf16(lo, hi);
                                }
                            }
                        }
                        const long n = n0 + (wn * TN16 + 2 * ip + (g & 1)) * 16 + (g >> 1) * 8;
                        if (m < p.M && n + 8 <= p.N) store_out(reinterpret_cast<v4i *>(outp + m * p.ldo + n), y);
                    }
                }
                return;
            }
        } else {
#pragma unroll
        for (int a = 0; a < NACC; ++a)
#pragma unroll
            for (int i = 0; i < TN; ++i)
#pragma unroll
                for (int j = 0; j < TM; ++j)
#pragma unroll
                    for (int e = 0; e < 16; ++e) acc[a][i][j][e] = 0;

        // V_MFMA_I32_32X32X32_I8: one wave per SIMD already runs it at ~85 % of the int8 peak, whereas
        // the 16x16x64 form needs two waves per SIMD and tops out at ~70 % (tools/probes/math_loop.hip).
        // Operand fragments out of the 16-row pieces: lane l holds row / channel (l & 31) = piece
        // (l >> 4) & 1, row l & 15 of it, and the 16 k-bytes of chunk 2 sub + (l >> 5) of the k-tile;
        // both operands use the same lane -> k map, so the (order-free) integer sum is exact.
        const int lane_x = ((lane >> 4) & 1) * 2048 + ((lane >> 5) * 16 + (lane & 15)) * 16;
        const int lane_w = (W_BITS == 4) ? ((lane >> 5) * 16 + (lane & 15)) * 16 + ((lane >> 4) & 1) * 8 : lane_x;
        // Register sets, indexed by the K = 32 sub-step modulo 4: activation fragments X and packed weights
        // WP are read TWO sub-steps ahead of their MFMAs (LDS latency under DMA traffic exceeds one
        // sub-step of 3-4 MFMAs), the nibbles are unpacked one sub-step ahead into WU (parity).
        v4i X[4][TM], WU[2][TN];
        v4i WP[4][TN];      // W4: only the first two words are used
        auto load_x = [&](int set, int slot, int sub) {          // sub = 0..3: K = 32 sub-steps of the stage
            const char *xs = smem + slot * STAGE + (sub >> 1) * 1024 + (sub & 1) * 512;
#pragma unroll
            for (int j = 0; j < TM; ++j)
                X[set][j] = *reinterpret_cast<const v4i *>(xs + (wm * TM + j) * 4096 + lane_x);
        };
        auto load_w = [&](int set, int slot, int sub) {
            const char *ws = smem + slot * STAGE + A_BYTES + (sub >> 1) * 1024 + (sub & 1) * 512;
#pragma unroll
            for (int i = 0; i < TN; ++i) {
                if (W_BITS == 4) {
                    const v2i pk = *reinterpret_cast<const v2i *>(ws + (wn * TN + i) * 2048 + lane_w);
                    WP[set][i][0] = pk[0];
                    WP[set][i][1] = pk[1];
                } else {
                    WP[set][i] = *reinterpret_cast<const v4i *>(ws + (wn * TN + i) * 4096 + lane_w);
                }
            }
        };
        auto unpack = [&](int from, int to) {
#pragma unroll
            for (int i = 0; i < TN; ++i) {
                if (W_BITS == 4) {
                    const int lo = WP[from][i][0], hi = WP[from][i][1];
                    WU[to][i][0] = (lo << 4) & 0xF0F0F0F0;
                    WU[to][i][1] = lo & 0xF0F0F0F0;
                    WU[to][i][2] = (hi << 4) & 0xF0F0F0F0;
                    WU[to][i][3] = hi & 0xF0F0F0F0;
                } else {
                    WU[to][i] = WP[from][i];
                }
            }
        };
        auto mfmas = [&](int sub) {
            const int a = NACC == 2 ? (sub & 1) : 0;
#pragma unroll
            for (int i = 0; i < TN; ++i)
#pragma unroll
                for (int j = 0; j < TM; ++j)
                    acc[a][i][j] = __builtin_amdgcn_mfma_i32_32x32x32_i8(WU[sub & 1][i], X[sub][j], acc[a][i][j], 0, 0, 0);
        };
        // One K = 32 sub-step: the MFMAs of sub-step s interleaved with the fragment reads of sub-step
        // s+2 and the nibble unpack of sub-step s+1, so that every gap between two MFMAs (32 cycles of
        // matrix-core time) carries its share of the LDS and VALU issue; grouped as
        // [reads][unpack][MFMAs] the in-order wave leaves a bubble per sub-step.
        constexpr int N_MFMA = TM * TN, N_DS = TM + TN, N_VALU = (W_BITS == 4) ? 6 * TN : 0;
        constexpr int DS_PER_GAP = (N_DS + N_MFMA - 1) / N_MFMA, VALU_PER_GAP = (N_VALU + N_MFMA - 1) / N_MFMA;
        auto substep = [&](int sub, int slot2, int sub2) {       // (slot2, sub2): where sub-step s+2 lives
            __builtin_amdgcn_sched_barrier(0);
#ifdef MQ_WS_XFIRST
            load_x((sub + 2) & 3, slot2, sub2);
            load_w((sub + 2) & 3, slot2, sub2);
#else
            load_w((sub + 2) & 3, slot2, sub2);          // the packed weights first: the unpack one sub-step later waits for THEM, and
            load_x((sub + 2) & 3, slot2, sub2);          // LDS returns in order -- its wait then leaves the activation reads in flight
#endif
            unpack((sub + 1) & 3, (sub + 1) & 1);
            mfmas(sub);
#pragma unroll
            for (int m = 0; m < N_MFMA; ++m) {   // a single wave issues back-to-back LDS reads slowly: spread them
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                  // MFMA
                __builtin_amdgcn_sched_group_barrier(0x100, DS_PER_GAP, 0);        // DS reads
                if (N_VALU) __builtin_amdgcn_sched_group_barrier(0x002, VALU_PER_GAP, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        };
        __builtin_amdgcn_s_barrier();                // B(0): stage 0 landed
        MQ_TL(0, 6);
#ifdef MQ_WS_STAMP
        const unsigned long long stamp0 = __builtin_readcyclecounter();
#endif
        load_x(0, 0, 0);
        load_w(0, 0, 0);
        load_x(1, 0, 1);
        load_w(1, 0, 1);
        unpack(0, 0);
        int cur = 0;
        for (int it = 0; it < nk; ++it) {
            __builtin_amdgcn_s_barrier();            // B(it+1): stage it+1 landed
            int nxt = cur + 1;
            if (nxt == S) nxt = 0;
            if (it + 1 >= nk) nxt = cur;             // last step: harmless re-reads of a live slot
            substep(0, cur, 2);
            substep(1, cur, 3);
            substep(2, nxt, 0);
            substep(3, nxt, 1);
            cur = nxt;
        }
        MQ_TL(0, 7);                                 // k-loop done
#ifdef MQ_WS_STAMP
        {
            const unsigned long long stamp1 = __builtin_readcyclecounter();
            if (tid == 0 && p.partial && p.splits == 1) {
                p.partial[blockIdx.x * 2] = (int)(stamp1 - stamp0);
                p.partial[blockIdx.x * 2 + 1] = nk;
            }
        }
#endif
        if (NACC == 2) {
#pragma unroll
            for (int i = 0; i < TN; ++i)
#pragma unroll
                for (int j = 0; j < TM; ++j) acc[0][i][j] += acc[1][i][j];
        }
        }
        if (EPI != EPI_I32) {                        // the parameter block sits behind the ring
            if (tid < BN) {
                const bool ok = (ACT && p.act == MQ_ACT_SILU_MUL) ? ((long)bn * (BN / 2) + (tid & (BN / 2 - 1)) < (p.N >> 1)) : (n0 + tid < p.N);
                par_sw[tid] = ok ? pr_sw : 0.0f;
                par_bs[tid] = (ok && p.bias) ? pr_bs : 0.0f;
                par_wz[tid] = (ok && p.w0) ? pr_wz : 0.0f;
                par_w1[tid] = (ok && p.w1) ? pr_w1 : 0.0f;
            }
            if (tid < BM) {
                const bool ok = m0 + tid < p.M;
                float sx = p.sx0;
                if (ok) {
                    if (p.sx_vec) sx = pr_sx;
                    else if (p.row_sel && pr_rs) sx = p.sx1;
                }
                par_sx[tid] = (WG == 2) ? 1.0f : sx;   // (activation groups alone: the row's scales went into the fold)
                par_xz[tid] = (ok && p.x0) ? pr_xz : 0.0f;
                par_x1[tid] = (ok && p.x1) ? pr_x1 : 0.0f;
            }
        }
    }

    // =========================== epilogue: all waves ==========================================
    __syncthreads();                                 // the ring is free: park the raw accumulators
    {
        // D layout of the 32x32 form: column (-> row m) = lane & 31, rows (-> channels) 8 q + 4 (lane >> 5) + e
        const int wm = wave / MW_N, wn = wave % MW_N;
        const int ml = lane & 31, nh = (lane >> 5) * 4;
        if (MF == 1) {
            // D layout of the 16x16 form: column (-> row m) = lane & 15, rows (-> channels) 4 (lane >> 4) + r
            if (wave < NM) {
#pragma unroll
                for (int j = 0; j < TM16; ++j)
#pragma unroll
                    for (int i = 0; i < TN16; ++i)
                        *reinterpret_cast<v4i *>(smem + ((wm * TM16 + j) * 16 + (lane & 15)) * PITCH + ((wn * TN16 + i) * 16 + (lane >> 4) * 4) * 4) =
                            acc16[i][j];
            }
        } else if (wave < NM) {
#pragma unroll
            for (int j = 0; j < TM; ++j)
#pragma unroll
                for (int i = 0; i < TN; ++i)
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        *reinterpret_cast<v4i *>(smem + ((wm * TM + j) * 32 + ml) * PITCH + ((wn * TN + i) * 32 + q * 8 + nh) * 4) =
                            v4i{acc[0][i][j][4 * q], acc[0][i][j][4 * q + 1], acc[0][i][j][4 * q + 2], acc[0][i][j][4 * q + 3]};
        }
    }
    __syncthreads();
    MQ_TL(0, 8);                                     // slab parked

    constexpr int LPR = BN / 8;                      // lanes per output row (8 channels per lane)
    constexpr int RPI = NT / LPR;                    // rows per iteration of the whole workgroup
    const int c8 = (tid % LPR) * 8;
    const long n = n0 + c8;
    const bool n_full = (n + 8 <= p.N) && p.vec_ok;
    const bool to_partial = p.splits > 1;
    float swv[8], bsv[8], wzv[8], w1v[8];
    if (EPI != EPI_I32 && !to_partial) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            swv[e] = par_sw[c8 + e];
            bsv[e] = par_bs[c8 + e];
            wzv[e] = par_wz[c8 + e];
            w1v[e] = par_w1[c8 + e];
        }
    }
    // Activation in the store (GemmArgs::act; preconditions host-checked in gemm_common): the Linear's output is formed and rounded
    // to the output dtype exactly as the fast path below stores it, then the activation like the torch ops on that tensor.
    // SILU_MUL: slab columns [0, BN/2) hold gate, [BN/2, BN) the same up channels (the loaders' pairing); a lane owns 8 OUTPUT channels.
    if constexpr (ACT) {
        if constexpr (EPI != EPI_I32 && WG == 0) {
            constexpr int DT = (EPI == EPI_F16) ? MQ_F16 : (EPI == EPI_BF16 ? MQ_BF16 : MQ_F32);
            const bool silu_rt = p.act == MQ_ACT_SILU_MUL;
            const bool bias_rt = p.bias != nullptr;
            constexpr int LPR_S = BN / 16;                                // lanes per row with half as many outputs
            const int rpi = silu_rt ? NT / LPR_S : RPI;
            const int arow = silu_rt ? tid / LPR_S : tid / LPR;
            const int ob = (silu_rt ? tid % LPR_S : tid % LPR) * 8;          // slab column of the (gate) operand; up: + BN / 2
            const long H = p.N >> 1;
            const long no = silu_rt ? (long)bn * (BN / 2) + ob : n0 + ob;    // output column
            const bool n_ok = silu_rt ? (no + 8 <= H) : (no + 8 <= p.N);
            float sg[8], su[8], bg[8], bu[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                sg[e] = par_sw[ob + e];
                bg[e] = par_bs[ob + e];
                su[e] = silu_rt ? par_sw[BN / 2 + ob + e] : 0.0f;
                bu[e] = silu_rt ? par_bs[BN / 2 + ob + e] : 0.0f;
            }
            auto rows = [&](auto silu_c, auto bias_c) {      // once per (activation, bias) combination: no uniform branch between two elements
                constexpr bool silu = decltype(silu_c)::value, has_bias = decltype(bias_c)::value;
#pragma unroll 1
                for (int r0 = 0; r0 < BM; r0 += rpi) {
                    const int row = r0 + arow;
                    const long m = m0 + row;
                    if (row >= BM || m >= p.M || !n_ok) continue;
                    const v4i g0 = *reinterpret_cast<const v4i *>(smem + row * PITCH + ob * 4);
                    const v4i g1 = *reinterpret_cast<const v4i *>(smem + row * PITCH + ob * 4 + 16);
                    v4i u0 = g0, u1 = g1;
                    if (silu) {
                        u0 = *reinterpret_cast<const v4i *>(smem + row * PITCH + (BN / 2 + ob) * 4);
                        u1 = *reinterpret_cast<const v4i *>(smem + row * PITCH + (BN / 2 + ob) * 4 + 16);
                    }
                    const float sxe = (W_BITS == 4) ? par_sx[row] * 0.0625f : par_sx[row];
                    const int ag[8] = {g0[0], g0[1], g0[2], g0[3], g1[0], g1[1], g1[2], g1[3]};
                    const int au[8] = {u0[0], u0[1], u0[2], u0[3], u1[0], u1[1], u1[2], u1[3]};
                    // the two Linear outputs in fp32 -- the plain launch's arithmetic, one rounding per operation -- then the activation
                    float gf[8], uf[8];
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        float g = (float)ag[e] * sxe;
                        g = g * sg[e];
                        if (has_bias) g = g + bg[e];
                        gf[e] = g;
                        float u = (float)au[e] * sxe;
                        u = u * su[e];
                        if (has_bias) u = u + bu[e];
                        uf[e] = u;
                    }
                
                    if constexpr (EPI == EPI_F32) {
                        float h[8];
#pragma unroll
                        for (int e = 0; e < 8; ++e) h[e] = silu ? act_silu_mul<DT>(gf[e], uf[e]) : act_quick_gelu<DT>(gf[e]);
                        float *o = reinterpret_cast<float *>(p.out) + m * p.ldo + no;
                        store_out(reinterpret_cast<v4f *>(o), v4f{h[0], h[1], h[2], h[3]});
                        store_out(reinterpret_cast<v4f *>(o + 4), v4f{h[4], h[5], h[6], h[7]});
                    } else {
                        v4i hw;
                        if (silu) {
#pragma unroll
                            for (int e = 0; e < 4; ++e) hw[e] = (int)act_silu_mul_pk<DT>(gf[2 * e], gf[2 * e + 1], uf[2 * e], uf[2 * e + 1]);
                        } else {
#pragma unroll
                            for (int e = 0; e < 4; ++e) hw[e] = (int)act_quick_gelu_pk<DT>(gf[2 * e], gf[2 * e + 1]);
                        }
                        store_out(reinterpret_cast<v4i *>(reinterpret_cast<unsigned short *>(p.out) + m * p.ldo + no), hw);
                    }
                }
            };
            if (silu_rt) { if (bias_rt) rows(std::true_type{}, std::true_type{}); else rows(std::true_type{}, std::false_type{}); }
            else { if (bias_rt) rows(std::false_type{}, std::true_type{}); else rows(std::false_type{}, std::false_type{}); }
        }
        return;
    }
    // Fast path (what every Linear of the prefill takes): whole 8-channel groups, fp16 / bf16 / fp32
    // output, no residual, no split-K.  Straight-line code: all slab reads first, then the arithmetic,
    // then the stores back to back -- this part runs once per workgroup with every workgroup of the
    // launch in it at the same time, so its latency chain is launch time ("fixed cost").
    if (EPI != EPI_I32 && !to_partial && !p.residual && __all(n_full)) {
        constexpr int ITERS = (BM + RPI - 1) / RPI;
        const int lrow = tid / LPR;
        v4i q0[ITERS], q1[ITERS];
        float sxr[ITERS], xzr[ITERS], x1r[ITERS];
        // RoPE in the store (GemmArgs::rope_cos): a 128-wide tile below rope_cols is one head; this lane's 8 channels d .. d+7
        // of the head and the rotate-half partner's (d +- 64) sit in lanes l and l ^ 8 of the same 16-lane row group
        constexpr bool ROPE_OK = (EPI == EPI_F16 || EPI == EPI_BF16) && BN == 128;
        const bool rope = ROPE_OK && p.rope_cos != nullptr && n0 < p.rope_cols;
        v4i rcs[ROPE_OK ? ITERS : 1], rsn[ROPE_OK ? ITERS : 1];
#pragma unroll
        for (int t = 0; t < ITERS; ++t) {
            const int row = (t * RPI + lrow < BM) ? t * RPI + lrow : BM - 1;
            q0[t] = *reinterpret_cast<const v4i *>(smem + row * PITCH + c8 * 4);
            q1[t] = *reinterpret_cast<const v4i *>(smem + row * PITCH + c8 * 4 + 16);
            sxr[t] = par_sx[row];
            xzr[t] = par_xz[row];
            x1r[t] = par_x1[row];
            if constexpr (ROPE_OK) {
                if (rope) {
                    long mr = m0 + row;
                    if (mr >= p.M) mr = p.M - 1;
                    rcs[t] = *reinterpret_cast<const v4i *>(reinterpret_cast<const unsigned short *>(p.rope_cos) + mr * 128 + c8);
                    rsn[t] = *reinterpret_cast<const v4i *>(reinterpret_cast<const unsigned short *>(p.rope_sin) + mr * 128 + c8);
                }
            }
        }
        // Packed fp32 arithmetic (v_pk_mul_f32 / v_pk_add_f32: two outputs per instruction, each lane
        // element rounded like the scalar form).  int4 weights: the accumulator carries a factor 16
        // (levels sit in the high nibble); float(16 a) * (s_x / 16) is the same real product as
        // float(a) * s_x with an exact power-of-two rescale on both sides, so one rounding, same bits,
        // and the shift is gone.
        typedef float v2f_t __attribute__((ext_vector_type(2)));
        const bool has_bias = p.bias != nullptr, has_x0 = p.x0 != nullptr, has_x1 = p.x1 != nullptr;
        v2f_t sw2[4], bs2[4], wz2[4], w12[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            sw2[e] = v2f_t{swv[2 * e], swv[2 * e + 1]};
            bs2[e] = v2f_t{bsv[2 * e], bsv[2 * e + 1]};
            wz2[e] = v2f_t{wzv[2 * e], wzv[2 * e + 1]};
            w12[e] = v2f_t{w1v[2 * e], w1v[2 * e + 1]};
        }
#pragma unroll
        for (int t = 0; t < ITERS; ++t) {
            const long m = m0 + t * RPI + lrow;
            const int a[8] = {q0[t][0], q0[t][1], q0[t][2], q0[t][3], q1[t][0], q1[t][1], q1[t][2], q1[t][3]};
            const float sxe = (W_BITS == 4 && !WG) ? sxr[t] * 0.0625f : sxr[t];      // (weight groups: the 1/16 went into the group scales)
            const v2f_t sx2 = v2f_t{sxe, sxe}, xz2 = v2f_t{xzr[t], xzr[t]}, x12 = v2f_t{x1r[t], x1r[t]};
            float y[8];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                v2f_t v = WG ? v2f_t{__int_as_float(a[2 * e]), __int_as_float(a[2 * e + 1])} : v2f_t{(float)a[2 * e], (float)a[2 * e + 1]};
                v = v * sx2;
                v = v * sw2[e];
                if (has_bias) v = v + bs2[e];
                if (has_x0) {
                    const v2f_t pr = xz2 * wz2[e];
                    v = v + pr;
                }
                if (has_x1) {
                    const v2f_t pr = x12 * w12[e];
                    v = v + pr;
                }
                y[2 * e] = v[0];
                y[2 * e + 1] = v[1];
            }
            if (m >= p.M || t * RPI + lrow >= BM) continue;
            if (EPI == EPI_F32) {
                float *o = reinterpret_cast<float *>(p.out) + m * p.ldo + n;
                store_out(reinterpret_cast<v4f *>(o), v4f{y[0], y[1], y[2], y[3]});
                store_out(reinterpret_cast<v4f *>(o + 4), v4f{y[4], y[5], y[6], y[7]});
            } else {
                v4i h;
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    h[e] = (int)((EPI == EPI_F16) ? pack2_f16(y[2 * e], y[2 * e + 1]) : pack2_bf16(y[2 * e], y[2 * e + 1]));
                if constexpr (ROPE_OK) {
                    if (rope) {                      // (uniform over the workgroup: every lane takes part in the exchange)
                        constexpr int DT = (EPI == EPI_F16) ? MQ_F16 : MQ_BF16;
                        const float sgn = (c8 < 64) ? -1.0f : 1.0f;       // rotate_half(x) = cat(-x2, x1)
                        if constexpr (EPI == EPI_F16) {
                            // fp16: packed half arithmetic gives the same bits as the fp32 form below -- a product of two halves is exact
                            // in fp32, so cast(a * c) IS the correctly rounded half product (v_pk_mul_f16); the sum of two halves is
                            // exact in fp32 unless one is below 2^-13 of the other, and then both forms return the larger one
                            // (v_pk_add_f16).  3 packed instructions per two outputs instead of ~20 (the fp32 form cost +4 us on the
                            // q|k|v launch, as much as the rotation's own launch: profiles/r5_full_prefill_kernel_split.txt).
                            typedef _Float16 h2 __attribute__((ext_vector_type(2)));
                            const int flip = (c8 < 64) ? (int)0x80008000 : 0;                 // -x2 in the low half of a head: sign bits
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                const int own_i = h[e];
                                const int oth_i = __builtin_amdgcn_update_dpp(0, own_i, 0x128 /* row_ror:8 */, 0xf, 0xf, false) ^ flip;
                                h2 own, oth, c2, s2;
                                __builtin_memcpy(&own, &own_i, 4);
                                __builtin_memcpy(&oth, &oth_i, 4);
                                const int cw = rcs[t][e], sw = rsn[t][e];
                                __builtin_memcpy(&c2, &cw, 4);
                                __builtin_memcpy(&s2, &sw, 4);
                                const h2 p1 = own * c2;
                                const h2 p2 = oth * s2;                           // (+-b) * sin
                                const h2 r = p1 + p2;
                                int r_i;
                                __builtin_memcpy(&r_i, &r, 4);
                                h[e] = r_i;
                            }
                        } else
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const unsigned own = (unsigned)h[e];
                            const unsigned oth = (unsigned)__builtin_amdgcn_update_dpp(0, h[e], 0x128 /* row_ror:8 */, 0xf, 0xf, false);
                            const unsigned cw = (unsigned)rcs[t][e], sw = (unsigned)rsn[t][e];
                            unsigned short ob[2];
#pragma unroll
                            for (int k = 0; k < 2; ++k) {
                                const float a = Elem<DT>::ld((unsigned short)(own >> (16 * k)));
                                const float b = Elem<DT>::ld((unsigned short)(oth >> (16 * k)));
                                const float c = Elem<DT>::ld((unsigned short)(cw >> (16 * k)));
                                const float sn = Elem<DT>::ld((unsigned short)(sw >> (16 * k)));
                                const float r = Elem<DT>::rnd(a * c) + Elem<DT>::rnd((sgn * b) * sn);
                                ob[k] = Elem<DT>::st(r);
                            }
                            h[e] = (int)((unsigned)ob[0] | ((unsigned)ob[1] << 16));
                        }
                    }
                }
                store_out(reinterpret_cast<v4i *>(reinterpret_cast<unsigned short *>(p.out) + m * p.ldo + n), h);
            }
        }
#ifdef MQ_WS_TL
        MQ_TL(0, 9);                                 // last store issued (wave 0)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        MQ_TL(0, 10);                                // wave 0's stores acknowledged
        __syncthreads();
        if (tl && tid == 0) {
            tl[11] = (int)__builtin_amdgcn_s_memtime();  // every wave's stores acknowledged
            tl[12] = (int)__builtin_amdgcn_s_memrealtime();
            tl[13] = nk;
        }
#endif
        return;
    }
#pragma unroll 1
    for (int r0 = 0; r0 < BM; r0 += RPI) {
        const int row = r0 + tid / LPR;
        const long m = m0 + row;
        if (row >= BM || m >= p.M || n >= p.N) continue;
        v4i q0 = *reinterpret_cast<const v4i *>(smem + row * PITCH + c8 * 4);
        v4i q1 = *reinterpret_cast<const v4i *>(smem + row * PITCH + c8 * 4 + 16);
        int a[8] = {q0[0], q0[1], q0[2], q0[3], q1[0], q1[1], q1[2], q1[3]};
        if (W_BITS == 4 && !WG) {                    // (weight groups: the registers hold float bits, the 1/16 is in the group scales)
#pragma unroll
            for (int e = 0; e < 8; ++e) a[e] >>= 4;
        }
        if (to_partial || EPI == EPI_I32) {
            int *o = to_partial ? p.partial + ((long)split * p.M + m) * p.N + n
                                : reinterpret_cast<int *>(p.out) + m * p.ldo + n;
            if (n_full) {
                *reinterpret_cast<v4i *>(o) = v4i{a[0], a[1], a[2], a[3]};
                *reinterpret_cast<v4i *>(o + 4) = v4i{a[4], a[5], a[6], a[7]};
            } else {
                for (int e = 0; e < 8; ++e)
                    if (n + e < p.N) o[e] = a[e];
            }
            continue;
        }
        const float sx = par_sx[row], xz = par_xz[row], x1s = par_x1[row];
        float res[8];
        if (p.residual) {
            if (EPI == EPI_F32) {
                const float *rp = reinterpret_cast<const float *>(p.residual) + m * p.ldr + n;
#pragma unroll
                for (int e = 0; e < 8; ++e) res[e] = (n + e < p.N) ? rp[e] : 0.0f;
            } else {
                const unsigned short *rp = reinterpret_cast<const unsigned short *>(p.residual) + m * p.ldr + n;
                if (n_full && p.res_vec) {
                    const v8us rv = *reinterpret_cast<const v8us *>(rp);
#pragma unroll
                    for (int e = 0; e < 8; ++e)
                        res[e] = (EPI == EPI_F16) ? f16_bits_to_f32(rv[e]) : bf16_bits_to_f32(rv[e]);
                } else {
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const unsigned short rb = (n + e < p.N) ? rp[e] : (unsigned short)0;
                        res[e] = (EPI == EPI_F16) ? f16_bits_to_f32(rb) : bf16_bits_to_f32(rb);
                    }
                }
            }
        }
        float y[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            float t = (WG ? __int_as_float(a[e]) : (float)a[e]) * sx;
            t = t * swv[e];
            if (p.bias) t = t + bsv[e];
            if (p.x0) {
                const float pr = xz * wzv[e];
                t = t + pr;
            }
            if (p.x1) {
                const float pr = x1s * w1v[e];
                t = t + pr;
            }
            if (p.residual) {   // torch: hidden + linear(x), the Linear's output rounded first
                if (EPI == EPI_F16) t = f16_bits_to_f32(f32_to_f16_bits(t));
                if (EPI == EPI_BF16) t = bf16_bits_to_f32(f32_to_bf16_bits(t));
                t = t + res[e];
            }
            y[e] = t;
        }
        if (EPI == EPI_F32) {
            float *o = reinterpret_cast<float *>(p.out) + m * p.ldo + n;
            if (n_full) {
                store_out(reinterpret_cast<v4f *>(o), v4f{y[0], y[1], y[2], y[3]});
                store_out(reinterpret_cast<v4f *>(o + 4), v4f{y[4], y[5], y[6], y[7]});
            } else {
                for (int e = 0; e < 8; ++e)
                    if (n + e < p.N) o[e] = y[e];
            }
        } else {
            v8us h;
#pragma unroll
            for (int e = 0; e < 8; ++e)
                h[e] = (EPI == EPI_F16) ? f32_to_f16_bits(y[e]) : f32_to_bf16_bits(y[e]);
            unsigned short *o = reinterpret_cast<unsigned short *>(p.out) + m * p.ldo + n;
            if (n_full) {
                store_out(reinterpret_cast<v8us *>(o), h);
            } else {
                for (int e = 0; e < 8; ++e)
                    if (n + e < p.N) o[e] = h[e];
            }
        }
    }
}

template <int BM, int BN, int MW_M, int MW_N, int NL, int S, int W_BITS, int EPI, int MF = 0, int WG = 0, bool ACT = false, bool DIRECT = false>
static int launch_ws(const GemmArgs &p, hipStream_t st)
{
    constexpr int PIECES = (BM / 16) * 2 + ((W_BITS == 4) ? (BN / 32) * 2 : (BN / 16) * 2);
    constexpr int RING = S * (PIECES + ((WG && !(MQ_WS_WG_ABL & 8)) ? 3 : 0)) * 1024, SLAB = BM * (BN * 4 + 16);   // (WG: three 1 KiB scale blocks per stage)
    constexpr int SMEM = (RING > SLAB ? RING : SLAB) + (4 * BN + 3 * BM) * 4;
    static_assert(SMEM <= 160 * 1024, "LDS budget");
    auto kern = gemm_ws_kernel<BM, BN, MW_M, MW_N, NL, S, W_BITS, EPI, MF, WG, ACT, DIRECT>;
    int rc = ensure_dynamic_lds((const void *)kern, SMEM);
    if (rc != MQ_OK) return rc;
    GemmArgs g = p;
    set_geometry(g, BM, BN, 128, W_BITS);
    if (!geometry_in_range(g)) return fail(MQ_EINVAL, "mq_gemm_w4a8: %u x %u x %d workgroups exceed the range of the launch-geometry arithmetic", g.m_blocks, g.n_blocks, g.splits);
    hipLaunchKernelGGL(kern, dim3(g.m_blocks * g.n_blocks * (unsigned)g.splits), dim3((MW_M * MW_N + NL) * 64), SMEM, st, g);
    return check_launch("gemm_ws");
}

// Activation in the store (GemmArgs::act): the 16x16x64 tiles (ids 44-48; the 32x32x32 ids 40-43 map onto their twins).
template <int W_BITS, int EPI>
static int dispatch_ws_act(const GemmArgs &p, int tile, hipStream_t st)
{
    if constexpr (EPI == EPI_I32) {
        return fail(MQ_EINVAL, "gemm_ws: an activation needs a floating-point output");
    } else {
        switch (tile) {
        case 40: case 44: case 50: return launch_ws<96, 128, 1, 4, 4, (W_BITS == 4 ? 7 : 5), W_BITS, EPI, 1, 0, true>(p, st);
        case 41: case 45: case 51: return launch_ws<128, 128, 2, 4, 4, (W_BITS == 4 ? 6 : 4), W_BITS, EPI, 1, 0, true>(p, st);
        case 42: case 46: case 52:
            if constexpr (W_BITS == 4) return launch_ws<192, 128, 2, 4, 4, 4, W_BITS, EPI, 1, 0, true>(p, st);
            else break;
        case 43: case 47: case 53: return launch_ws<64, 128, 1, 4, 4, (W_BITS == 4 ? 8 : 6), W_BITS, EPI, 1, 0, true>(p, st);
        case 48: case 54: return launch_ws<96, 128, 2, 4, 4, (W_BITS == 4 ? 7 : 5), W_BITS, EPI, 1, 0, true>(p, st);
        default: break;
        }
        return fail(MQ_EINVAL, "gemm_ws: tile %d has no activation epilogue", tile);
    }
}

template <int W_BITS, int EPI>
int dispatch_ws(const GemmArgs &p, int tile, hipStream_t st)
{
    // (QuickGELU is elementwise: the slab-free ids carry it themselves; silu(gate) * up needs the slab to bring the two halves together)
    const bool direct_gelu = p.act == MQ_ACT_QUICK_GELU && tile >= 50 && tile <= 54 && (EPI == EPI_F16 || EPI == EPI_BF16) && p.vec_ok && p.par_ok
                             && ((uintptr_t)p.s_w) % 16 == 0 && g_pp_act_slab == 0;
    if (p.act != MQ_ACT_NONE && !direct_gelu) return dispatch_ws_act<W_BITS, EPI>(p, tile, st);
    switch (tile) {
    // one math wave per SIMD (1 x 4 wave tiles) or two (2 x 4), four loader waves
    case 40: return launch_ws<96, 128, 1, 4, 4, (W_BITS == 4 ? 7 : 5), W_BITS, EPI>(p, st);
    case 41: return launch_ws<128, 128, 2, 4, 4, (W_BITS == 4 ? 6 : 4), W_BITS, EPI>(p, st);
    case 42:
        if constexpr (W_BITS == 4) return launch_ws<192, 128, 2, 4, 4, 4, W_BITS, EPI>(p, st);
        else break;
    case 43: return launch_ws<64, 128, 1, 4, 4, (W_BITS == 4 ? 8 : 6), W_BITS, EPI>(p, st);
    // the same tiles with V_MFMA_I32_16X16X64_I8 in the math waves (round 5)
    case 44: return launch_ws<96, 128, 1, 4, 4, (W_BITS == 4 ? 7 : 5), W_BITS, EPI, 1>(p, st);
    case 45: return launch_ws<128, 128, 2, 4, 4, (W_BITS == 4 ? 6 : 4), W_BITS, EPI, 1>(p, st);
    case 46:
        if constexpr (W_BITS == 4) return launch_ws<192, 128, 2, 4, 4, 4, W_BITS, EPI, 1>(p, st);
        else break;
    case 47: return launch_ws<64, 128, 1, 4, 4, (W_BITS == 4 ? 8 : 6), W_BITS, EPI, 1>(p, st);
    // 96 x 128 with TWO math waves per SIMD (48 x 32 per wave): the 16x16x64 form issues at full rate from two waves
    case 48: return launch_ws<96, 128, 2, 4, 4, (W_BITS == 4 ? 7 : 5), W_BITS, EPI, 1>(p, st);
    // ids 50-54: the 16x16x64 tiles 44-48 with the DIRECT epilogue (no LDS slab; see the kernel's head).  Plain 16-bit launches only
    // (no split-K, residual, second rank-1 term, RoPE; aligned parameters, N % 8 == 0): anything else takes the slab twin.
    case 50: case 51: case 52: case 53: case 54: {
        const bool direct_ok = (EPI == EPI_F16 || EPI == EPI_BF16) && p.splits == 1 && (!p.residual || p.res_vec) && !p.x1 && !p.rope_cos && p.vec_ok
                               && p.par_ok && ((uintptr_t)p.s_w) % 16 == 0;
        if (!direct_ok) return dispatch_ws<W_BITS, EPI>(p, tile - 6, st);
        if constexpr (EPI == EPI_F16 || EPI == EPI_BF16) {
            switch (tile) {
            case 50: return launch_ws<96, 128, 1, 4, 4, (W_BITS == 4 ? 7 : 5), W_BITS, EPI, 1, 0, false, true>(p, st);
            case 51: return launch_ws<128, 128, 2, 4, 4, (W_BITS == 4 ? 6 : 4), W_BITS, EPI, 1, 0, false, true>(p, st);
            case 52:
                if constexpr (W_BITS == 4) return launch_ws<192, 128, 2, 4, 4, 4, W_BITS, EPI, 1, 0, false, true>(p, st);
                else return dispatch_ws<W_BITS, EPI>(p, 45, st);
            case 53: return launch_ws<64, 128, 1, 4, 4, (W_BITS == 4 ? 8 : 6), W_BITS, EPI, 1, 0, false, true>(p, st);
            default: return launch_ws<96, 128, 2, 4, 4, (W_BITS == 4 ? 7 : 5), W_BITS, EPI, 1, 0, false, true>(p, st);
            }
        }
        break;
    }
    // (a 32 x 128 tile -- 320 workgroups for the ViT's 1024 x 1280 outputs instead of 160, two per CU -- was measured in round 5
    //  and is not faster: proj 9.68 against 9.73 us, fc2 14.8 against 13.4, profiles/r5_ws_tile_32x128.txt; not instantiated)
    default: break;
    }
    return fail(MQ_EINVAL, "gemm_ws: unknown tile %d", tile);
}

// Group-wise scales (GemmArgs::sw_groups and / or sx_groups, group_k 64 or a multiple of 128, no split-K): the 16x16x64 tiles with the
// fold.  WGM: 1 weight groups, 2 activation groups, 3 both.
template <int W_BITS, int EPI, int WGM>
static int dispatch_ws_group_mode(const GemmArgs &p, int tile, hipStream_t st)
{
    if (tile >= 50 && tile <= 54) tile -= 6;          // (the slab-free ids: their 16x16x64 twins carry the group fold)
    if (tile == 46) {
        // 192 x 128: twelve waves at <= 168 registers each cannot hold the extra fp32 accumulators (136 bytes of scratch).  Its shapes take
        // 128 x 128 or 96 x 128, whichever needs fewer row-rounds (gate|up: 7 x 128 against 10 x 96, measured 14-20 % ahead; ViT fc1:
        // 2 x 128 against 2 x 96, measured 15 % behind -- profiles/r5_group_gemm_ws_fold.txt)
        const long cus = device_cu_count(), nb = (p.N + 127) / 128;
        const long r128 = (((p.M + 127) / 128) * nb + cus - 1) / cus * 128, r96 = (((p.M + 95) / 96) * nb + cus - 1) / cus * 96;
        tile = r128 < r96 ? 45 : 48;
    }
    switch (tile) {
    // (ring depths: one stage fewer than the per-channel tiles where the scale blocks would not fit the 160 KiB)
    case 45: return launch_ws<128, 128, 2, 4, 4, (W_BITS == 4 ? 5 : 4), W_BITS, EPI, 1, WGM>(p, st);
    case 47: return launch_ws<64, 128, 1, 4, 4, (W_BITS == 4 ? 7 : 5), W_BITS, EPI, 1, WGM>(p, st);
    case 48: return launch_ws<96, 128, 2, 4, 4, (W_BITS == 4 ? 6 : 5), W_BITS, EPI, 1, WGM>(p, st);
    default: break;
    }
    return fail(MQ_EINVAL, "gemm_ws: tile %d has no group-scale variant", tile);
}

template <int W_BITS, int EPI>
int dispatch_ws_wgroup(const GemmArgs &p, int tile, hipStream_t st)
{
    if constexpr (EPI == EPI_I32) {
        return fail(MQ_EINVAL, "gemm_ws: group scales need a floating-point output");
    } else {
        if (p.sw_groups && p.sx_groups) return dispatch_ws_group_mode<W_BITS, EPI, 3>(p, tile, st);
        if (p.sw_groups) return dispatch_ws_group_mode<W_BITS, EPI, 1>(p, tile, st);
        return dispatch_ws_group_mode<W_BITS, EPI, 2>(p, tile, st);
    }
}
#define MQ_WSG_INST(B, E) template int dispatch_ws_wgroup<B, E>(const GemmArgs &, int, hipStream_t);
MQ_WSG_INST(4, EPI_F16) MQ_WSG_INST(4, EPI_BF16) MQ_WSG_INST(4, EPI_F32) MQ_WSG_INST(8, EPI_F16) MQ_WSG_INST(8, EPI_BF16) MQ_WSG_INST(8, EPI_F32)

// explicit instantiations used by gemm_w4a8.hip
#define MQ_WS_INST(B, E) template int dispatch_ws<B, E>(const GemmArgs &, int, hipStream_t);
MQ_WS_INST(4, EPI_F16) MQ_WS_INST(4, EPI_BF16) MQ_WS_INST(4, EPI_F32) MQ_WS_INST(4, EPI_I32)
MQ_WS_INST(8, EPI_F16) MQ_WS_INST(8, EPI_BF16) MQ_WS_INST(8, EPI_F32) MQ_WS_INST(8, EPI_I32)

}  // namespace mq
