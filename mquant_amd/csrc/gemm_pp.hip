// gemm_pp.hip -- "ping-pong" W4A8 GEMM (8 waves, int4 weights, activations in the tiled layout).  Same operands,
// LDS image, XCD map and epilogue as the kernels of gemm_w4a8.hip / gemm_ws.hip; what differs is WHO does what WHEN.
//
// Why (round 4, profiles/r4_pp_*.txt):
//   * the pipelined 256 x 256 kernel lets all eight waves run the same phase at the same time; the ISA hipcc makes
//     of it issues the ten fragment reads of the next k-tile right in front of the per-phase barrier and waits for
//     them right behind it: every phase exposes one LDS round trip with both waves of a SIMD idle;
//   * the wave-specialised kernels have ONE math wave per SIMD, and an in-order wave cannot overlap its own LDS
//     reads and nibble unpack with its own MFMAs (0.36-0.39 us per 128-deep k-step where the MFMAs need 0.19).
//
// Here the two waves of a SIMD alternate roles (the 8-phase idea of the CDNA4 guide, section 5): the workgroup's
// waves form two groups of four (one wave of each group per SIMD, group = half of the tile's rows); group 1 runs
// ONE barrier interval behind group 0.  In every interval one group issues nothing but MFMAs at raised priority
// while the other one does everything else for its next step: fragment reads, the LDS-DMA of the step RING-1
// ahead, the int4 -> int8 nibble unpack, the counted vmcnt.  The matrix pipe of a SIMD always has one wave with
// its operands in registers, and the reads of a step have the partner's whole MFMA interval to land.
//
//   step t of one group (KT k-tiles of 64; two intervals, the other group is shifted by one):
//     L(t): ds_read X (t), packed W (t); LDS-DMA step t+RING-1 -> slot of step t-1; unpack; vmcnt (step t+1)  | barrier
//     M(t): KT * TM * TN MFMA (V_MFMA_I32_16X16X64_I8)                                                         | barrier
//
// Slot safety: the slot of step t-1 is last read in L(t-1) of the lagging group; those reads are complete before
// that group's barrier -- an explicit s_waitcnt lgkmcnt(0) in front of it (it used to rest on hipcc keeping the X reads
// ahead of the W reads the unpack waits for; the MFMAs behind the barrier need every operand anyway, so the wait is free) --
// and the slot is refilled from L(t) of the leading group on -- the interval after that barrier.  Step t+1 is
// complete in LDS for everybody after the barrier that ends L(t) of the lagging group; its first reader is L(t+1)
// of the leading group, behind that barrier.
//
// Measured with cycle stamps (profiles/r4_pp_stamps.txt): the 256 x 256 loop spends 1182 cycles per k-tile where
// its MFMAs need 1024 (87 %), but the chip runs it at 1.60 GHz (MFMAs alone: 2.17 GHz, everything but the MFMAs:
// 2.36 GHz): with all CUs busy the wide GEMM is bound by the power budget, not by its schedule.
//
// Reference semantics: fake_quant/quant_utils.py:384 (F.linear on the quantize-dequantized tensors); the int32
// accumulators are exact, the epilogue is gemm_common.h's.
#include "gemm_common.h"

namespace mq {

#ifndef MQ_PP_ABL
#define MQ_PP_ABL 0   // timing-only ablations (wrong results): 1 no DMA in the loop, 2 no fragment reads, 4 no unpack, 8 no MFMA
#endif

template <int BM, int BN, int KT, int RING, int EPI, bool ACT = false>
__global__ __launch_bounds__(512) void gemm_w4a8_pp_kernel(GemmArgs p)
{
    kernarg_warm<sizeof(GemmArgs), true>();        // one scalar-load round trip instead of six (mq_common.h)
    constexpr int NWAVES = 8;
    constexpr int TM = BM / 32;                              // 16-row activation fragments per wave (two groups over M)
    constexpr int TN = BN / 64;                              // 16-channel weight fragments per wave (four waves over N)
    constexpr int XP = BM / 16, WP = BN / 32, PPK = XP + WP; // 1 KiB pieces per 64-wide k-tile
    constexpr int PIECES = KT * PPK;                         // ... per step
    constexpr int LPW = (PIECES + NWAVES - 1) / NWAVES;      // LDS-DMA instructions per wave and step (waves >= REM: one fewer)
    constexpr int REM = PIECES % NWAVES;
    constexpr int SLOT = PIECES * 1024;
    static_assert(BM % 32 == 0 && BN % 128 == 0 && TN % 2 == 0, "tile shape");
    static_assert(RING >= 3 && (RING - 2) * LPW < 64, "ring depth (vmcnt is 6 bits)");

    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave >> 2, wn = wave & 3;                // waves w and w + 4 share a SIMD

    // Persistent over the work ids: a launch with more tiles than CUs (gate|up: 444) starts one workgroup per CU (256: a multiple
    // of 8, so id % 8 stays the workgroup's XCD, tile_of_id) and each walks ids b, b + gridDim, ...: a CU's second tile starts behind one
    // barrier instead of a new workgroup's dispatch (5.3 k cycles between the two, profiles/r4_pp_cu_timeline.txt).
    const unsigned total_ids = p.m_blocks * p.n_blocks * (unsigned)p.splits;
    for (unsigned wid = blockIdx.x; wid < total_ids; wid += gridDim.x) {
    int bm, bn, split, sb, ns;
    tile_of_id(p, total_ids, wid, bm, bn, split);
    k_range_of_split(p, split, sb, ns);                       // in steps of KT k-tiles
    const long m0 = (long)bm * BM;
    const long nt0 = (long)bn * (BN / 16);
    const long kts = p.K_pad >> 6;

    // LDS-DMA sources: one contiguous KiB per piece, wave-uniform base (SGPRs) + lane * 16
    const char *src[LPW];
#pragma unroll
    for (int i = 0; i < LPW; ++i) {
        int f = wave + i * NWAVES;
        if (f >= PIECES) f = PIECES - 1;                      // never issued (REM), keeps the address valid
        const int kt = f / PPK, r = f % PPK;
        if (r < XP) {
            long mtg = m0 / 16 + r;
            const long MT = (p.M + 15) >> 4;
            if (mtg >= MT) mtg = MT - 1;
            src[i] = reinterpret_cast<const char *>(p.a) + (mtg * kts + (long)sb * KT + kt) * 1024;
        } else {
            long ntp = nt0 / 2 + (r - XP);
            if (ntp >= p.n_pairs) ntp = p.n_pairs - 1;
            if (ACT && p.act == MQ_ACT_SILU_MUL) {
                // silu(gate) * up in the store (GemmArgs::act): LDS pair slot 2 i = gate pair i, slot 2 i + 1 = up pair i of this
                // n-block's BN / 2 output channels -- wave wn (slots 2 wn, 2 wn + 1 for BN = 256) then holds 32 gate channels and
                // the SAME 32 up channels, and its private epilogue slab has both operands of an output
                const int g = r - XP;
                const long hp = p.n_pairs >> 1;                  // pairs per half (N / 2 is a multiple of 32: host-checked)
                long pi = (long)bn * (WP / 2) + (g >> 1);
                if (pi >= hp) pi = hp - 1;
                ntp = (g & 1) ? hp + pi : pi;
            }
            src[i] = reinterpret_cast<const char *>(p.w) + (ntp * kts + (long)sb * KT + kt) * 1024;
        }
    }
    const bool short_wave = REM != 0 && wave >= REM;          // this wave owns LPW - 1 pieces per step
    const unsigned lane_off = lane * 16;
    const unsigned lds0 = (unsigned)(size_t)(lds_void *)smem;
    auto issue = [&](int slot) {                              // this wave's pieces of the next step not yet requested
        const unsigned base = lds0 + slot * SLOT;
#pragma unroll
        for (int i = 0; i < LPW; ++i) {
            if (i == LPW - 1 && short_wave) break;
            dma16_s(src[i], lane_off, base + (wave + i * NWAVES) * 1024);
            src[i] += KT * 1024;
        }
    };
    // at most `younger` steps requested after the wanted one may still be in flight
#define MQ_PP_VM(n) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(n) : "memory")
#define MQ_PP_WAIT_CASE(k)                                                         \
    case k:                                                                        \
        if ((k) <= RING - 2) {                                                     \
            if (short_wave) MQ_PP_VM((k) * (LPW - 1) < 64 ? (k) * (LPW - 1) : 0);  \
            else MQ_PP_VM((k) * LPW < 64 ? (k) * LPW : 0);                          \
        }                                                                          \
        break;
    auto wait_younger = [&](int younger) {
        switch (younger) {
            MQ_PP_WAIT_CASE(0) MQ_PP_WAIT_CASE(1) MQ_PP_WAIT_CASE(2) MQ_PP_WAIT_CASE(3)
            MQ_PP_WAIT_CASE(4) MQ_PP_WAIT_CASE(5) MQ_PP_WAIT_CASE(6)
        default: MQ_PP_VM(0); break;
        }
    };

#ifdef MQ_PP_STAMP
    // stamp build (tools/gemm_kslope.py, tools/clock_recon.py): per work id 8 ints in the split-K workspace -- s_memtime ticks and
    // s_memrealtime ticks (100 MHz, chip-wide) over the k-loop, the k-steps, realtime at loop start / after the epilogue, HW_ID, XCC_ID
    const unsigned long long stamp0 = __builtin_readcyclecounter();
    const unsigned long long real0 = __builtin_amdgcn_s_memrealtime();
#endif
    v4i acc[TN][TM];
#pragma unroll
    for (int i = 0; i < TN; ++i)
#pragma unroll
        for (int j = 0; j < TM; ++j) acc[i][j] = v4i{0, 0, 0, 0};
    v4i X[KT][TM], Wp[KT][TN / 2], Wu[KT][TN];

    const char *x_lane = smem + (grp * TM) * 1024 + lane * 16;
    const char *w_lane = smem + (XP + wn * (TN / 2)) * 1024 + lane * 16;
    auto reads = [&](int slot) {
#pragma unroll
        for (int kt = 0; kt < KT; ++kt) {
            const char *xs = x_lane + slot * SLOT + kt * (PPK * 1024);
#pragma unroll
            for (int j = 0; j < TM; ++j) X[kt][j] = *reinterpret_cast<const v4i *>(xs + j * 1024);
        }
#pragma unroll
        for (int kt = 0; kt < KT; ++kt) {                    // W after X: the unpack's wait covers every read of the step
            const char *ws = w_lane + slot * SLOT + kt * (PPK * 1024);
#pragma unroll
            for (int i = 0; i < TN / 2; ++i) Wp[kt][i] = *reinterpret_cast<const v4i *>(ws + i * 1024);
        }
    };
    auto unpack = [&]() {                                     // nibbles into the HIGH half of int8 bytes (value x 16)
#pragma unroll
        for (int kt = 0; kt < KT; ++kt)
#pragma unroll
            for (int i = 0; i < TN; ++i) {
                const int lo = Wp[kt][i >> 1][(i & 1) * 2], hi = Wp[kt][i >> 1][(i & 1) * 2 + 1];
                Wu[kt][i][0] = (lo << 4) & 0xF0F0F0F0;
                Wu[kt][i][1] = lo & 0xF0F0F0F0;
                Wu[kt][i][2] = (hi << 4) & 0xF0F0F0F0;
                Wu[kt][i][3] = hi & 0xF0F0F0F0;
            }
    };
#define MQ_PP_BAR()                              \
    do {                                         \
        __builtin_amdgcn_sched_barrier(0);       \
        __builtin_amdgcn_s_barrier();            \
        __builtin_amdgcn_sched_barrier(0);       \
    } while (0)

    // ---- prologue: the first RING-1 steps requested, step 0 landed ------------------------------------------
    const int pre = ns < RING - 1 ? ns : RING - 1;
#pragma unroll
    for (int t = 0; t < RING - 1; ++t)
        if (t < pre) issue(t);
    wait_younger(pre - 1);
    MQ_PP_BAR();
    if (grp == 1) MQ_PP_BAR();                                // group 1 runs one interval behind group 0

    int s_cur = 0, s_fill = RING - 1;                         // slots of steps t and t-1 (= t+RING-1)
#define MQ_PP_STEP(STEADY)                                                                      \
    do {                                                                                        \
        if (!(MQ_PP_ABL & 2)) reads(s_cur);                                                     \
        if (!(MQ_PP_ABL & 1) && ((STEADY) || t + RING - 1 < ns)) issue(s_fill);                 \
        if (!(MQ_PP_ABL & 4)) unpack();                                                         \
        if (!(MQ_PP_ABL & 1)) {                                                                 \
            if (STEADY) {                                                                       \
                wait_younger(RING - 2);                                                         \
            } else if (t + 1 < ns) {                                                            \
                const int last = t + RING - 1 < ns - 1 ? t + RING - 1 : ns - 1;                 \
                wait_younger(last - (t + 1));                                                   \
            }                                                                                   \
        }                                                                                       \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); /* every fragment read of L(t) has returned */ \
        MQ_PP_BAR();                                                                            \
        if (!(MQ_PP_ABL & 8)) {                                                                 \
            __builtin_amdgcn_s_setprio(1);                                                      \
            _Pragma("unroll") for (int kt = 0; kt < KT; ++kt)                                   \
                _Pragma("unroll") for (int i = 0; i < TN; ++i)                                  \
                    _Pragma("unroll") for (int j = 0; j < TM; ++j)                              \
                        acc[i][j] = __builtin_amdgcn_mfma_i32_16x16x64_i8(Wu[kt][i], X[kt][j], acc[i][j], 0, 0, 0); \
            __builtin_amdgcn_s_setprio(0);                                                      \
        }                                                                                       \
        MQ_PP_BAR();                                                                            \
        if (++s_cur == RING) s_cur = 0;                                                         \
        if (++s_fill == RING) s_fill = 0;                                                       \
    } while (0)
    int t = 0;
    for (; t + RING - 1 < ns; ++t) MQ_PP_STEP(true);          // steady state: no conditionals between the barriers
    for (; t < ns; ++t) MQ_PP_STEP(false);                    // tail (and short reductions)
    if (grp == 0) MQ_PP_BAR();                                // group 0's share of the stagger

#ifdef MQ_PP_STAMP
    {
        const unsigned long long stamp1 = __builtin_readcyclecounter();
        const unsigned long long real1 = __builtin_amdgcn_s_memrealtime();
        if (tid == 0 && p.partial && p.splits == 1) {
            unsigned hw, xcc;
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
            int *o = p.partial + (long)wid * 8;
            o[0] = (int)(stamp1 - stamp0);
            o[1] = ns;
            o[2] = (int)(real1 - real0);
            o[3] = (int)real0;
            o[5] = (int)hw;
            o[6] = (int)xcc;
        }
    }
#endif
    // (Requesting one dword per 128-byte line of the NEXT tile's first RING - 1 weight steps here, in front of the epilogue, so that
    //  its prologue's LDS-DMAs hit the L2, was measured in round 5: 0.4 % SLOWER over the bench, profiles/r5_bench_ab_pp_next_tile_prefetch.txt)
    gemm_epilogue<TM, TN, NWAVES, RING * SLOT, 4, EPI, ACT>(p, acc, smem, wave, lane, grp, wn, m0, nt0, split);
#ifdef MQ_PP_STAMP
    if (tid == 0 && p.partial && p.splits == 1) p.partial[(long)wid * 8 + 4] = (int)__builtin_amdgcn_s_memrealtime();
#endif
    if (wid + gridDim.x < total_ids) __syncthreads();         // the slab of this tile is the ring of the next
    }
#undef MQ_PP_STEP
#undef MQ_PP_BAR
#undef MQ_PP_WAIT_CASE
#undef MQ_PP_VM
}

template <int BM, int BN, int KT, int RING, int EPI, bool ACT = false>
static int launch_pp(const GemmArgs &p, hipStream_t st)
{
    constexpr int SMEM = RING * KT * (BM / 16 + BN / 32) * 1024;
    static_assert(SMEM <= 160 * 1024, "LDS budget");
    auto kern = gemm_w4a8_pp_kernel<BM, BN, KT, RING, EPI, ACT>;
    const int rc = ensure_dynamic_lds((const void *)kern, SMEM);
    if (rc != MQ_OK) return rc;
    GemmArgs g = p;
    set_geometry(g, BM, BN, 64 * KT, 4);
    if (!geometry_in_range(g)) return fail(MQ_EINVAL, "mq_gemm_w4a8: %u x %u x %d workgroups exceed the range of the launch-geometry arithmetic", g.m_blocks, g.n_blocks, g.splits);
    unsigned ids = g.m_blocks * g.n_blocks * (unsigned)g.splits;
#ifndef MQ_PP_ONE_TILE_PER_WG
    const unsigned cus = (unsigned)device_cu_count();         // persistent: one workgroup per CU (150 KiB of LDS each)
    if (ids > cus) ids = cus;
#endif
    hipLaunchKernelGGL(kern, dim3(ids), dim3(512), SMEM, st, g);
    return check_launch("gemm_w4a8_pp");
}

// tile ids 14 (256 x 256), 15 (128 x 128), 16 (96 x 128), 17 (192 x 128), 18 (64 x 128), 19 (128 x 256)
template <int EPI>
int launch_gemm_pp(const GemmArgs &p, int tile, hipStream_t st)
{
    if (p.act != MQ_ACT_NONE) {
        // activation in the store (GemmArgs::act): the 256-wide tiles only -- a wave then holds four 16-channel fragments, i.e. a
        // gate pair AND its up pair (the plan sends every other shape to the wave-specialised kernels)
        if constexpr (EPI != EPI_I32) {
            if (tile == 14) return launch_pp<256, 256, 1, 6, EPI, true>(p, st);
            if (tile == 19) return launch_pp<128, 256, 1, 8, EPI, true>(p, st);
        }
        return fail(MQ_EINVAL, "gemm_pp: tile %d has no activation epilogue", tile);
    }
    switch (tile) {
    case 14: return launch_pp<256, 256, 1, 6, EPI>(p, st);
    case 15: return launch_pp<128, 128, 2, 6, EPI>(p, st);
    case 16: return launch_pp<96, 128, 2, 7, EPI>(p, st);
    case 17: return launch_pp<192, 128, 2, 5, EPI>(p, st);
    case 18: return launch_pp<64, 128, 2, 8, EPI>(p, st);
    case 19: return launch_pp<128, 256, 1, 8, EPI>(p, st);
    default: break;
    }
    return fail(MQ_EINVAL, "gemm_pp: unknown tile %d", tile);
}

template int launch_gemm_pp<EPI_F16>(const GemmArgs &, int, hipStream_t);
template int launch_gemm_pp<EPI_BF16>(const GemmArgs &, int, hipStream_t);
template int launch_gemm_pp<EPI_F32>(const GemmArgs &, int, hipStream_t);
template int launch_gemm_pp<EPI_I32>(const GemmArgs &, int, hipStream_t);

}  // namespace mq
