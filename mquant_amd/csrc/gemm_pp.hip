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

// DIRECT (round 6, tile id 20): the plain 16-bit epilogue without the LDS slab, as in gemm_ws.hip -- a wave dequantises its own
// accumulators in the MFMA D layout, V_PERMLANE16_SWAP pairs the quads of adjacent channel tiles, 16-byte stores.  The slab epilogue is
// 11 k of the ~82 k cycles a 256 x 256 tile takes (profiles/r4_pp_cu_timeline.txt).
template <int BM, int BN, int KT, int RING, int EPI, bool ACT = false, bool DIRECT = false>
__global__ __launch_bounds__(512) void gemm_w4a8_pp_kernel(GemmArgs p)
{
    static_assert(!DIRECT || (EPI == EPI_F16 || EPI == EPI_BF16), "direct epilogue: 16-bit outputs");
    static_assert(!(DIRECT && ACT) || BN == 256, "direct silu(gate) * up: a wave holds one gate pair and its up pair");
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
    const long kts = p.K_pad >> 6;
    // LDS-DMA sources: one contiguous KiB per piece, wave-uniform base (SGPRs) + lane * 16
    const char *src[LPW];
    auto set_src = [&](int bn, long m0, long nt0, int sb) {
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
    };
    // DIRECT: the epilogue leaves the ring alone, so a workgroup requests the first stages of its NEXT tile before it dequantises and
    // stores the current one (`primed`): the second tile of a CU finds its operands in LDS instead of waiting a cold prologue out
    bool primed = false;
    for (unsigned wid = blockIdx.x; wid < total_ids; wid += gridDim.x) {
    int bm, bn, split, sb, ns;
    tile_of_id(p, total_ids, wid, bm, bn, split);
    k_range_of_split(p, split, sb, ns);                       // in steps of KT k-tiles
    const long m0 = (long)bm * BM;
    const long nt0 = (long)bn * (BN / 16);
    if (!primed) set_src(bn, m0, nt0, sb);
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
    if (DIRECT && primed) {
        MQ_PP_VM(0);                                          // requested before the previous tile's epilogue: long landed
    } else {
#pragma unroll
        for (int t = 0; t < RING - 1; ++t)
            if (t < pre) issue(t);
        wait_younger(pre - 1);
    }
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
    if constexpr (DIRECT) {
        primed = false;
        if (wid + gridDim.x < total_ids) {                    // the next tile's first stages go out now (every wave is past its last
            int nbm, nbn, nsplit, nsb, nns;                   // fragment read: the re-aligning barrier above)
            tile_of_id(p, total_ids, wid + gridDim.x, nbm, nbn, nsplit);
            k_range_of_split(p, nsplit, nsb, nns);
            set_src(nbn, (long)nbm * BM, (long)nbn * (BN / 16), nsb);
            const int npre = nns < RING - 1 ? nns : RING - 1;
#pragma unroll
            for (int t = 0; t < RING - 1; ++t)
                if (t < npre) issue(t);
            primed = true;
        }
        // per-lane parameters in the D layout: 4 consecutive channels per 16-channel tile, one row per 16-row tile (fetched here: the
        // k-loop has no registers to spare; one exposed L2 round trip per tile against the slab's two barriers and LDS round trip)
        const bool has_bias = p.bias != nullptr, has_x0 = p.x0 != nullptr;
        const int g = lane >> 4;
        unsigned short *outp = reinterpret_cast<unsigned short *>(p.out);
        typedef float v2f __attribute__((ext_vector_type(2)));
        float d_sx[TM], d_xz[TM];
#pragma unroll
        for (int j = 0; j < TM; ++j) {
            long mr = m0 + (grp * TM + j) * 16 + (lane & 15);
            if (mr >= p.M) mr = p.M - 1;
            float sx = p.sx0;
            if (p.sx_vec) sx = p.sx_vec[mr];
            else if (p.row_sel && p.row_sel[mr]) sx = p.sx1;
            d_sx[j] = sx * 0.0625f;                          // int4 levels sit in the high nibble: exact power-of-two rescale
            d_xz[j] = (!ACT && has_x0) ? p.x0[mr] : 0.0f;
        }
        if constexpr (ACT) {
            // silu(gate) * up (GemmArgs::act; the loaders paired the halves): fragments 0, 1 of this wave are 32 gate channels, fragments
            // 2, 3 the same 32 up channels -- both operands of an output sit in ONE lane.  The two Linear outputs as the plain launch
            // forms and rounds them, then the activation like the torch ops (mq_common.h act_silu_mul_pk; fp32 cannot come here)
            constexpr int DT = (EPI == EPI_F16) ? MQ_F16 : MQ_BF16;
            const long H = p.N >> 1;
            const long cw = nt0 * 8 + wn * 32;                // first output channel of this wave
            v4f sg[2], su[2], bg[2], bu[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                long c = cw + i * 16 + g * 4;
                if (c + 4 > H) c = 0;
                sg[i] = *reinterpret_cast<const v4f *>(p.s_w + c);
                su[i] = *reinterpret_cast<const v4f *>(p.s_w + H + c);
                bg[i] = has_bias ? *reinterpret_cast<const v4f *>(p.bias + c) : v4f{0.f, 0.f, 0.f, 0.f};
                bu[i] = has_bias ? *reinterpret_cast<const v4f *>(p.bias + H + c) : v4f{0.f, 0.f, 0.f, 0.f};
            }
#pragma unroll
            for (int j = 0; j < TM; ++j) {
                const long m = m0 + (grp * TM + j) * 16 + (lane & 15);
                unsigned pk[2][2];
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const v4i ga = acc[i][j], ua = acc[i + 2][j];
                    float gf[4], uf[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        float gv = (float)ga[r] * d_sx[j];
                        gv = gv * sg[i][r];
                        if (has_bias) gv = gv + bg[i][r];
                        gf[r] = gv;
                        float uv = (float)ua[r] * d_sx[j];
                        uv = uv * su[i][r];
                        if (has_bias) uv = uv + bu[i][r];
                        uf[r] = uv;
                    }
                    pk[i][0] = act_silu_mul_pk<DT>(gf[0], gf[1], uf[0], uf[1]);
                    pk[i][1] = act_silu_mul_pk<DT>(gf[2], gf[3], uf[2], uf[3]);
                }
                const auto s0 = __builtin_amdgcn_permlane16_swap(pk[0][0], pk[1][0], false, false);
                const auto s1 = __builtin_amdgcn_permlane16_swap(pk[0][1], pk[1][1], false, false);
                const long n = cw + (g & 1) * 16 + (g >> 1) * 8;
                if (m < p.M && n + 8 <= H)
                    store_out(reinterpret_cast<v4i *>(outp + m * p.ldo + n), v4i{(int)s0[0], (int)s1[0], (int)s0[1], (int)s1[1]});
            }
        } else {
        const long nw = nt0 * 16 + wn * (TN * 16);
        const float *bsp = has_bias ? p.bias : p.s_w, *wzp = has_x0 ? p.w0 : p.s_w;
        v4f d_sw[TN], d_bs[TN], d_wz[TN];
#pragma unroll
        for (int i = 0; i < TN; ++i) {
            long nq = nw + i * 16 + g * 4;
            if (nq + 4 > p.N) nq = 0;
            d_sw[i] = *reinterpret_cast<const v4f *>(p.s_w + nq);
            d_bs[i] = *reinterpret_cast<const v4f *>(bsp + nq);
            d_wz[i] = *reinterpret_cast<const v4f *>(wzp + nq);
        }
#pragma unroll
        for (int j = 0; j < TM; ++j) {
            const long m = m0 + (grp * TM + j) * 16 + (lane & 15);
            const v2f sx2 = v2f{d_sx[j], d_sx[j]}, xz2 = v2f{d_xz[j], d_xz[j]};
#pragma unroll
            for (int ip = 0; ip < TN / 2; ++ip) {
                unsigned pk[2][2];
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int i = 2 * ip + h;
                    const v4i a = acc[i][j];
                    v2f v0 = v2f{(float)a[0], (float)a[1]}, v1 = v2f{(float)a[2], (float)a[3]};
                    v0 = v0 * sx2;
                    v1 = v1 * sx2;
                    v0 = v0 * v2f{d_sw[i][0], d_sw[i][1]};
                    v1 = v1 * v2f{d_sw[i][2], d_sw[i][3]};
                    if (has_bias) {
                        v0 = v0 + v2f{d_bs[i][0], d_bs[i][1]};
                        v1 = v1 + v2f{d_bs[i][2], d_bs[i][3]};
                    }
                    if (has_x0) {
                        const v2f p0 = xz2 * v2f{d_wz[i][0], d_wz[i][1]}, p1 = xz2 * v2f{d_wz[i][2], d_wz[i][3]};
                        v0 = v0 + p0;
                        v1 = v1 + p1;
                    }
                    pk[h][0] = (EPI == EPI_F16) ? pack2_f16(v0[0], v0[1]) : pack2_bf16(v0[0], v0[1]);
                    pk[h][1] = (EPI == EPI_F16) ? pack2_f16(v1[0], v1[1]) : pack2_bf16(v1[0], v1[1]);
                }
                const auto s0 = __builtin_amdgcn_permlane16_swap(pk[0][0], pk[1][0], false, false);
                const auto s1 = __builtin_amdgcn_permlane16_swap(pk[0][1], pk[1][1], false, false);
                const long n = nw + (2 * ip + (g & 1)) * 16 + (g >> 1) * 8;
                if (m < p.M && n + 8 <= p.N)
                    store_out(reinterpret_cast<v4i *>(outp + m * p.ldo + n), v4i{(int)s0[0], (int)s1[0], (int)s0[1], (int)s1[1]});
            }
        }
        }
    } else {
        gemm_epilogue<TM, TN, NWAVES, RING * SLOT, 4, EPI, ACT>(p, acc, smem, wave, lane, grp, wn, m0, nt0, split);
    }
#ifdef MQ_PP_STAMP
    if (tid == 0 && p.partial && p.splits == 1) p.partial[(long)wid * 8 + 4] = (int)__builtin_amdgcn_s_memrealtime();
#endif
    if (!DIRECT && wid + gridDim.x < total_ids) __syncthreads();   // the slab of this tile is the ring of the next
    }
#undef MQ_PP_STEP
#undef MQ_PP_BAR
#undef MQ_PP_WAIT_CASE
#undef MQ_PP_VM
}

template <int BM, int BN, int KT, int RING, int EPI, bool ACT = false, bool DIRECT = false>
static int launch_pp(const GemmArgs &p, hipStream_t st)
{
    constexpr int SMEM = RING * KT * (BM / 16 + BN / 32) * 1024;
    static_assert(SMEM <= 160 * 1024, "LDS budget");
    auto kern = gemm_w4a8_pp_kernel<BM, BN, KT, RING, EPI, ACT, DIRECT>;
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

// test hook (mq_gemm_debug_force bit 16 of `splits`): keep the slab form of the activation epilogue (A/B, tile sweeps)
thread_local int g_pp_act_slab = 0;

// tile ids 14 (256 x 256), 15 (128 x 128), 16 (96 x 128), 17 (192 x 128), 18 (64 x 128), 19 (128 x 256), 20 (256 x 256, slab-free epilogue)
template <int EPI>
int launch_gemm_pp(const GemmArgs &p, int tile, hipStream_t st)
{
    if (p.act != MQ_ACT_NONE) {
        // activation in the store (GemmArgs::act): the 256-wide tiles only -- a wave then holds four 16-channel fragments, i.e. a
        // gate pair AND its up pair (the plan sends every other shape to the wave-specialised kernels)
        if constexpr (EPI == EPI_F16 || EPI == EPI_BF16) {
            // silu(gate) * up without the slab (both operands of an output sit in one lane of the D layout), next tile primed
            if ((tile == 14 || tile == 20) && p.act == MQ_ACT_SILU_MUL && ((uintptr_t)p.s_w) % 16 == 0 && (!p.bias || ((uintptr_t)p.bias) % 16 == 0)
                && g_pp_act_slab == 0)
                return launch_pp<256, 256, 1, 6, EPI, true, true>(p, st);
        }
        if constexpr (EPI != EPI_I32) {
            if (tile == 14 || tile == 20) return launch_pp<256, 256, 1, 6, EPI, true>(p, st);
            if (tile == 19) return launch_pp<128, 256, 1, 8, EPI, true>(p, st);
        }
        return fail(MQ_EINVAL, "gemm_pp: tile %d has no activation epilogue", tile);
    }
    if (tile == 20) {
        // the 256 x 256 tile with the slab-free epilogue: plain 16-bit launches only, anything else takes tile 14
        const bool direct_ok = (EPI == EPI_F16 || EPI == EPI_BF16) && p.splits == 1 && !p.residual && !p.x1 && !p.acc_float && p.vec_ok && p.par_ok
                               && ((uintptr_t)p.s_w) % 16 == 0;
        if constexpr (EPI == EPI_F16 || EPI == EPI_BF16) {
            if (direct_ok) return launch_pp<256, 256, 1, 6, EPI, false, true>(p, st);
        }
        tile = 14;
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
