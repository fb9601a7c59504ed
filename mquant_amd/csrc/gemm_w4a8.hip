// gemm_w4a8.hip -- int8 activations x int4 (or int8) weights -> int32 -> fused dequant.
//
//   acc[m][n] = sum_k a[m][k] * w[n][k]        V_MFMA_I32_16X16X64_I8, exact
//   y[m][n]   = ((float(acc) * s_x[sel(m)]) * s_w[n]) + bias[n] + x0[m] * w0[n]
//
// Structure (one workgroup = BM x BN output tile, K walked in steps of 128 bytes):
//   * both operands arrive by LDS-DMA (global_load_lds, 16 B per lane) into a STAGES-deep
//     ring; loads stay in flight across the single per-step s_barrier (counted vmcnt,
//     never 0 in the steady state);
//   * the LDS image of every 16x64 fragment is lane-linear (lane l at byte 16*l), so
//     fragment reads are conflict-free ds_read_b128 / ds_read_b64 with no swizzle:
//       - weights are stored pre-tiled in HBM in exactly that order (weight_formats.hip);
//       - activations are row-major in HBM, the per-lane SOURCE address does the
//         re-tiling (row = m0 + (l & 15), bytes 16*(l >> 4) .. +16 of the 64-wide k-tile);
//   * int4 weights stay packed in LDS; each wave expands the nibbles it is about to feed
//     to the matrix core into the HIGH nibble of int8 bytes (three VALU ops per 8 weights,
//     no sign-extension); the x16 factor is removed by an exact arithmetic shift;
//   * the weight fragment is the MFMA "A" operand and the activation fragment the "B"
//     operand, so every lane ends up with 4 consecutive output channels of one row;
//   * workgroup -> tile map is XCD-aware (8 XCDs, private L2s): the m-blocks that share a
//     weight panel are consecutive on ONE XCD;
//   * optional split-K: integer partial sums go to a workspace and are combined by
//     splitk_reduce_kernel in a fixed order (integer addition: exact and order-free),
//     which also applies the dequant epilogue.
//
// Reference semantics: fake_quant/quant_utils.py:384 (F.linear on fake-quant tensors) and
// :367-376 (split: channel 0 through L1 in fp32).
#include "mq_common.h"
#include "gemm_common.h"

namespace mq {

template <int W_BITS, int EPI>
int dispatch_ws(const GemmArgs &p, int tile, hipStream_t st);   // gemm_ws.hip (tiled activations only)
template <int EPI>
int launch_gemm_pp(const GemmArgs &p, int tile, hipStream_t st);   // gemm_pp.hip (ping-pong kernels, W4, tiled activations)
template <int W_BITS, int EPI>
int dispatch_ws_wgroup(const GemmArgs &p, int tile, hipStream_t st);   // gemm_ws.hip (weight-group fold in the 16x16x64 math loop)
template <int EPI>
int launch_gemm_skinny(const GemmArgs &p, hipStream_t st);             // gemm_skinny.hip (M <= 64 rows, W4, tiled activations; tile id 60)
int skinny_slices(long M, long N, long K_pad, size_t ws_bytes);
int skinny_wg_slices(long M, long N, long K_pad, size_t ws_bytes);
template <int EPI>
int launch_gemm_skinny_wg(const GemmArgs &p, hipStream_t st);          // gemm_skinny.hip (M <= 16, K <= 4096: the K slices are a workgroup's waves; id 61)

// GROUPED (--a_groupsize): the int32 accumulators of one activation group (64 or a multiple of 128 k) are scaled by
// the group's activation scale of their row and added to fp32 accumulators in ascending group order; the epilogue
// gets the float bits (GemmArgs::acc_float).  Exact integers inside a group, one fp32 rounding per group and row.
template <int BM, int BN, int WARPS_M, int WARPS_N, int STAGES, int W_BITS, int EPI, int DMA_POS, bool GROUPED = false>
__global__ __launch_bounds__(WARPS_M *WARPS_N * 64) void gemm_w4a8_kernel(GemmArgs p)
{
    kernarg_warm<sizeof(GemmArgs), true>();
    constexpr int NWAVES = WARPS_M * WARPS_N;
    constexpr int TM = BM / WARPS_M / 16;          // activation fragments per wave
    constexpr int TN = BN / WARPS_N / 16;          // weight fragments per wave
    constexpr int X_FRAGS = (BM / 16) * 2;         // 1 KiB DMA pieces per stage (2 k-tiles)
    constexpr int W_PIECES = (W_BITS == 4) ? (BN / 16) : (BN / 16) * 2;
    constexpr int PIECES = X_FRAGS + W_PIECES;
    constexpr int LPW = PIECES / NWAVES;           // DMA instructions per wave per stage
    constexpr int X_BYTES = X_FRAGS * 1024;
    constexpr int STAGE_BYTES = PIECES * 1024;
    static_assert(BM % (WARPS_M * 16) == 0 && BN % (WARPS_N * 16) == 0, "tile shape");
    static_assert(PIECES % NWAVES == 0, "DMA pieces must divide evenly over the waves");
    static_assert(STAGES >= 2 && STAGES <= 8 && (STAGES - 2) * LPW < 64, "ring depth (vmcnt is 6 bits)");

    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WARPS_N, wn = wave % WARPS_N;

    // ---- workgroup -> (split, bn, bm), XCD-aware and bijective (gemm_common.h) ---------
    int bm, bn, split, kb, nk;
    tile_of_block(p, bm, bn, split);               // split-K: one XCD works on ONE k-slice, so its L2
    k_range_of_split(p, split, kb, nk);            // holds that slice of A once for all its tiles
    const long m0 = (long)bm * BM;
    const long nt0 = (long)bn * (BN / 16);
    const long kps = p.K_pad >> 7;                 // k-steps in the whole reduction
    const long k_begin = kb;

    // ---- per-lane DMA source addresses (piece f = wave + i*NWAVES) ----------------------
    const char *src[LPW];
    int step_bytes[LPW];
#pragma unroll
    for (int i = 0; i < LPW; ++i) {
        const int f = wave + i * NWAVES;
        if (f < X_FRAGS) {
            const int mt = f >> 1, kt = f & 1;
            long row = m0 + mt * 16 + (lane & 15);
            if (row >= p.M) row = p.M - 1;
            src[i] = reinterpret_cast<const char *>(p.a) + row * p.lda + kt * 64 + (lane >> 4) * 16 +
                     k_begin * 128;
            step_bytes[i] = 128;
            if (p.a_tiled) {   // one contiguous KiB per (16-row tile, k-tile): 3x the L2 rate of the row gather
                long mtg = m0 / 16 + mt;
                const long MT = (p.M + 15) >> 4;
                if (mtg >= MT) mtg = MT - 1;
                src[i] = reinterpret_cast<const char *>(p.a) + ((mtg * (p.K_pad >> 6) + k_begin * 2 + kt) * 64 + lane) * 16;
                step_bytes[i] = 2048;
            }
        } else if (W_BITS == 4) {
            // piece g = (n-tile pair, k-tile of the step): image [ntp][kt][lane][16 B]
            const int g = f - X_FRAGS;
            long ntp = nt0 / 2 + (g >> 1);
            if (ntp >= p.n_pairs) ntp = p.n_pairs - 1;
            src[i] = reinterpret_cast<const char *>(p.w) +
                     (((ntp * kps + k_begin) * 2 + (g & 1)) * 64 + lane) * 16;
            step_bytes[i] = 2048;
        } else {
            const int g = f - X_FRAGS;
            long nt = nt0 + (g >> 1);
            if (nt >= p.n_tiles) nt = p.n_tiles - 1;
            src[i] = reinterpret_cast<const char *>(p.w) +
                     (((nt * kps + k_begin) * 2 + (g & 1)) * 64 + lane) * 16;
            step_bytes[i] = 2048;
        }
    }

    auto issue_stage = [&](int stage, int it) {
        char *base = smem + stage * STAGE_BYTES;
#pragma unroll
        for (int i = 0; i < LPW; ++i) {
            const int f = wave + i * NWAVES;
            dma16(src[i] + (long)it * step_bytes[i], base + f * 1024);
        }
    };

    v4i acc[TN][TM];
    v4f facc[GROUPED ? TN : 1][GROUPED ? TM : 1];
#pragma unroll
    for (int i = 0; i < TN; ++i)
#pragma unroll
        for (int j = 0; j < TM; ++j) {
            acc[i][j] = v4i{0, 0, 0, 0};
            if (GROUPED) facc[i][j] = v4f{0.f, 0.f, 0.f, 0.f};
        }
    // D layout: column (lane & 15) = activation row, so a lane needs ONE group scale per row tile
    auto fold_group = [&](long gi) {
        // Zero-padded tail of K_pad (group_k = 64 and K % 128 == 64): the accumulators are 0 there, so the scaled sum adds
        // nothing -- but the asymmetric constant term is NOT an accumulator product and must not be added a second time
        // for the last real group (advisor finding r4: K = 192 -> K_pad = 256, 3 groups, 4 k-tiles).
        const bool tail = gi >= p.n_groups;
        if (tail) gi = p.n_groups - 1;
#pragma unroll
        for (int j = 0; j < TM; ++j) {
            long row = m0 + (wm * TM + j) * 16 + (lane & 15);
            if (row >= p.M) row = p.M - 1;
            float sg = p.sx_groups ? p.sx_groups[row * p.n_groups + gi] : 1.0f;   // (weight groups alone: the row scale waits in the epilogue)
            if (W_BITS == 4) sg = sg * 0.0625f;                // the int4 levels sit in the high nibble: exact rescale
            const float sh = (p.shift_groups && !tail) ? p.shift_groups[row * p.n_groups + gi] : 0.0f;
#pragma unroll
            for (int i = 0; i < TN; ++i) {
                // D layout: register r of the tile = channel 16 (nt0 + wn TN + i) + 4 (lane >> 4) + r
                const long nb = (nt0 + wn * TN + i) * 16 + (lane >> 4) * 4;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float t = (float)acc[i][j][r] * sg;
                    if (p.sw_groups) {                         // weight-group scale of (group, channel): one more rounding
                        const long nn = nb + r < p.N ? nb + r : p.N - 1;
                        t = t * p.sw_groups[gi * p.N + nn];
                    }
                    float f = facc[GROUPED ? i : 0][GROUPED ? j : 0][r] + t;
                    if (p.shift_groups && !tail) {             // asymmetric groups: the constant part of the group's levels
                        const long nn = nb + r < p.N ? nb + r : p.N - 1;
                        const float u = sh * p.wsum_groups[gi * p.N + nn];
                        f = f + u;
                    }
                    facc[GROUPED ? i : 0][GROUPED ? j : 0][r] = f;
                }
                acc[i][j] = v4i{0, 0, 0, 0};
            }
        }
    };

    // ---- main loop ------------------------------------------------------------------------
#pragma unroll
    for (int s = 0; s < STAGES - 1; ++s)
        if (s < nk) issue_stage(s, s);

    int cur = 0;
    for (int it = 0; it < nk; ++it) {
        // stage `it` has landed when at most the (STAGES-2) younger stages are outstanding
        const int younger = nk - 1 - it;
        if (younger >= STAGES - 2) {
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"((STAGES - 2) * LPW) : "memory");
        } else {                                    // drain: fewer stages behind this one
            switch (younger) {
            case 5: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(STAGES > 7 ? 5 * LPW : 0) : "memory"); break;
            case 4: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(STAGES > 6 ? 4 * LPW : 0) : "memory"); break;
            case 3: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(STAGES > 5 ? 3 * LPW : 0) : "memory"); break;
            case 2: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(STAGES > 4 ? 2 * LPW : 0) : "memory"); break;
            case 1: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(STAGES > 3 ? LPW : 0) : "memory"); break;
            default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
            }
        }
        __builtin_amdgcn_s_barrier();   // everyone's pieces landed; everyone left stage it-1
        const bool more = it + STAGES - 1 < nk;
        int nxt = cur + STAGES - 1;
        if (nxt >= STAGES) nxt -= STAGES;
        if (DMA_POS == 0 && more) issue_stage(nxt, it + STAGES - 1);

        const char *xs = smem + cur * STAGE_BYTES;
        const char *ws = xs + X_BYTES;
#pragma unroll
        for (int kt = 0; kt < 2; ++kt) {
            v4i xf[TM];
#pragma unroll
            for (int j = 0; j < TM; ++j) {
                const int mt = wm * TM + j;
                xf[j] = *reinterpret_cast<const v4i *>(xs + (mt * 2 + kt) * 1024 + lane * 16);
            }
            // the DMA of the stage after next is issued behind the first fragment reads, so its
            // issue slots overlap MFMA execution instead of delaying the first MFMA of the step
            if (DMA_POS == 1 && kt == 0 && more) issue_stage(nxt, it + STAGES - 1);
            if (DMA_POS == 2 && kt == 1 && more) issue_stage(nxt, it + STAGES - 1);
#pragma unroll
            for (int i = 0; i < TN; ++i) {
                const int nt = wn * TN + i;
                v4i wf;
                if (W_BITS == 4) {
                    const v2i pk = *reinterpret_cast<const v2i *>(ws + ((nt >> 1) * 2 + kt) * 1024 + lane * 16 + (nt & 1) * 8);
                    wf[0] = (pk[0] << 4) & 0xF0F0F0F0;
                    wf[1] = pk[0] & 0xF0F0F0F0;
                    wf[2] = (pk[1] << 4) & 0xF0F0F0F0;
                    wf[3] = pk[1] & 0xF0F0F0F0;
                } else {
                    wf = *reinterpret_cast<const v4i *>(ws + (nt * 2 + kt) * 1024 + lane * 16);
                }
#pragma unroll
                for (int j = 0; j < TM; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_i32_16x16x64_i8(wf, xf[j], acc[i][j], 0, 0, 0);
            }
            if (GROUPED && p.group_k == 64) fold_group((k_begin + it) * 2 + kt);
        }
        if (GROUPED && p.group_k >= 128) {
            const long kstep = k_begin + it + 1;                       // k-steps done, in units of 128
            const int per = p.group_k >> 7;
            if (kstep % per == 0 || it + 1 == nk) fold_group((kstep - 1) / per);
        }
        if (++cur == STAGES) cur = 0;
    }
    if (GROUPED) {
#pragma unroll
        for (int i = 0; i < TN; ++i)
#pragma unroll
            for (int j = 0; j < TM; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[i][j][r] = __float_as_int(facc[GROUPED ? i : 0][GROUPED ? j : 0][r]);
    }

    gemm_epilogue<TM, TN, NWAVES, STAGES * STAGE_BYTES, W_BITS, EPI>(p, acc, smem, wave, lane, wm, wn, m0, nt0, split);
}

// Combine split-K partials (fixed order s = 0..splits-1; integer sums are exact) and apply
// the same epilogue.  One thread per 4 consecutive output channels.
template <int EPI>
__global__ __launch_bounds__(256) void splitk_reduce_kernel(GemmArgs p)
{
    const long quads_per_row = ceil_div(p.N, 4);
    const long total = p.M * quads_per_row;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (long)gridDim.x * blockDim.x) {
        const long m = i / quads_per_row;
        const long n = (i - m * quads_per_row) * 4;
        v4i a = {0, 0, 0, 0};
        const bool vec = (n + 4 <= p.N) && (p.N % 4 == 0);
        const long sstride = p.M * p.N;
        const int *src0 = p.partial + m * p.N + n;
        int s = 0;
        if (vec) {
            // four partials in flight (a rolled loop waits for every load in turn: ~0.8 us per split; the skinny kernel's 14-28 K
            // slices made that the longest part of a decode GEMM); integer sums: exact in any order
            for (; s + 4 <= p.splits; s += 4) {
                const v4i t0 = *reinterpret_cast<const v4i *>(src0 + (long)s * sstride);
                const v4i t1 = *reinterpret_cast<const v4i *>(src0 + (long)(s + 1) * sstride);
                const v4i t2 = *reinterpret_cast<const v4i *>(src0 + (long)(s + 2) * sstride);
                const v4i t3 = *reinterpret_cast<const v4i *>(src0 + (long)(s + 3) * sstride);
                a += (t0 + t1) + (t2 + t3);
            }
        }
        for (; s < p.splits; ++s) {
            const int *src = src0 + (long)s * sstride;
            if (vec) {
                const v4i t = *reinterpret_cast<const v4i *>(src);
                a += t;
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (n + r < p.N) a[r] += src[r];
            }
        }
        float sx = p.sx0, xz = 0.0f, x1v = 0.0f;
        if (EPI != EPI_I32) {
            if (p.sx_vec) sx = p.sx_vec[m];
            else if (p.row_sel && p.row_sel[m]) sx = p.sx1;
            if (p.x0) xz = p.x0[m];
            if (p.x1) x1v = p.x1[m];
        }
        store_quad<EPI>(p, m, n, a, sx, xz, x1v);
    }
}

// =================================================================================================
// Software-pipelined variant for the large GEMMs (256 x 256 tile, 8 waves, int4 weights).
//
// The K loop advances in 64-wide k-tiles ("phases"); a 6-slot LDS ring (6 x 24 KiB) keeps four
// k-tiles of LDS-DMA in flight.  The MFMA operands are double buffered in registers: while the 32
// MFMAs of k-tile p issue, the ds_reads of k-tile p+1 fill the other register set, so no LDS
// latency is exposed after the per-phase barrier:
//
//   phase p:  s_waitcnt vmcnt(9)  (k-tile p+1 landed; p+2..p+4 stay in flight)
//             s_barrier           (p+1 visible to all; everybody has finished reading slot of p-1)
//             DMA k-tile p+5 -> slot (p+5) % 6 (== slot of p-1)
//             ds_read fragments of k-tile p+1 -> frag[(p+1) & 1]
//             32 x V_MFMA_I32_16X16X64_I8 on frag[p & 1]
// =================================================================================================
template <int EPI>
__global__ __launch_bounds__(512) void gemm_w4a8_pipe_kernel(GemmArgs p)
{
    kernarg_warm<sizeof(GemmArgs), true>();
    constexpr int BM = 256, BN = 256, WARPS_N = 4, NWAVES = 8, TM = 8, TN = 4;
    constexpr int RING = 6, X_PIECES = BM / 16, W_PIECES = BN / 32, PIECES = X_PIECES + W_PIECES;
    constexpr int LPW = PIECES / NWAVES;                    // 3 DMA instructions per wave per k-tile
    constexpr int SLOT_BYTES = PIECES * 1024, X_BYTES = X_PIECES * 1024;
    static_assert(PIECES % NWAVES == 0, "DMA pieces must divide evenly over the waves");

    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WARPS_N, wn = wave % WARPS_N;

    int bm, bn, split, ktb, nt;
    tile_of_block(p, bm, bn, split);
    k_range_of_split(p, split, ktb, nt);                     // in 64-wide k-tiles for this kernel
    const long m0 = (long)bm * BM;
    const long nt0 = (long)bn * (BN / 16);
    const long kts = p.K_pad >> 6;
    const long kt_begin = ktb;

    const char *src[LPW];
    int step_bytes[LPW];
#pragma unroll
    for (int i = 0; i < LPW; ++i) {
        const int f = wave + i * NWAVES;
        if (f < X_PIECES) {
            long row = m0 + f * 16 + (lane & 15);
            if (row >= p.M) row = p.M - 1;
            src[i] = reinterpret_cast<const char *>(p.a) + row * p.lda + (lane >> 4) * 16 + kt_begin * 64;
            step_bytes[i] = 64;
            if (p.a_tiled) {
                long mtg = m0 / 16 + f;
                const long MT = (p.M + 15) >> 4;
                if (mtg >= MT) mtg = MT - 1;
                src[i] = reinterpret_cast<const char *>(p.a) + ((mtg * kts + kt_begin) * 64 + lane) * 16;
                step_bytes[i] = 1024;
            }
        } else {
            long ntp = nt0 / 2 + (f - X_PIECES);
            if (ntp >= p.n_pairs) ntp = p.n_pairs - 1;
            src[i] = reinterpret_cast<const char *>(p.w) + ((ntp * kts + kt_begin) * 64 + lane) * 16;
            step_bytes[i] = 1024;
        }
    }
    auto issue = [&](int slot, int t) {
        char *base = smem + slot * SLOT_BYTES;
#pragma unroll
        for (int i = 0; i < LPW; ++i)
            dma16(src[i] + (long)t * step_bytes[i], base + (wave + i * NWAVES) * 1024);
    };
    // wait until at most `newer` k-tiles issued after the wanted one are still in flight
    auto wait_newer = [&](int newer) {
        switch (newer) {
        case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
        case 1: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LPW) : "memory"); break;
        case 2: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * LPW) : "memory"); break;
        case 3: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(3 * LPW) : "memory"); break;
        default: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(4 * LPW) : "memory"); break;
        }
    };

    struct Frag {
        v4i x[TM];      // activation fragments (MFMA B operand)
        v4i wpk[TN / 2];  // packed weights: two n-tiles per 16 bytes
    };
    auto load_frag = [&](Frag &fr, int slot) {
        const char *xs = smem + slot * SLOT_BYTES;
        const char *ws = xs + X_BYTES;
#pragma unroll
        for (int j = 0; j < TM; ++j)
            fr.x[j] = *reinterpret_cast<const v4i *>(xs + (wm * TM + j) * 1024 + lane * 16);
#pragma unroll
        for (int i = 0; i < TN / 2; ++i)
            fr.wpk[i] = *reinterpret_cast<const v4i *>(ws + (wn * (TN / 2) + i) * 1024 + lane * 16);
    };

    v4i acc[TN][TM];
#pragma unroll
    for (int i = 0; i < TN; ++i)
#pragma unroll
        for (int j = 0; j < TM; ++j) acc[i][j] = v4i{0, 0, 0, 0};

    auto mfmas = [&](const Frag &fr) {
#pragma unroll
        for (int i = 0; i < TN; ++i) {
            const int lo = fr.wpk[i >> 1][(i & 1) * 2], hi = fr.wpk[i >> 1][(i & 1) * 2 + 1];
            v4i wf;
            wf[0] = (lo << 4) & 0xF0F0F0F0;
            wf[1] = lo & 0xF0F0F0F0;
            wf[2] = (hi << 4) & 0xF0F0F0F0;
            wf[3] = hi & 0xF0F0F0F0;
#pragma unroll
            for (int j = 0; j < TM; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_i32_16x16x64_i8(wf, fr.x[j], acc[i][j], 0, 0, 0);
        }
    };

    // ---- prologue: k-tiles 0..4 in flight, k-tile 0 into registers --------------------------------
    const int pre = nt < RING - 1 ? nt : RING - 1;
    for (int t = 0; t < pre; ++t) issue(t, t);
    wait_newer(pre - 1);
    __builtin_amdgcn_s_barrier();
    Frag fa, fb;
    load_frag(fa, 0);

    int slot_next = 1;          // slot of k-tile p+1
    int slot_fill = RING - 1;   // slot of k-tile p+5
    auto advance = [&]() {
        if (++slot_next == RING) slot_next = 0;
        if (++slot_fill == RING) slot_fill = 0;
    };
    // steady state: no conditionals inside, so the compiler can emit COUNTED lgkmcnt waits (the
    // ds_reads of the next k-tile stay in flight under the MFMAs of the current one)
    auto phase_steady = [&](const Frag &cur, Frag &nxt, int pidx) {
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(3 * LPW) : "memory");   // k-tile p+1 landed
        __builtin_amdgcn_s_barrier();
        issue(slot_fill, pidx + RING - 1);
        load_frag(nxt, slot_next);
        mfmas(cur);
        advance();
    };
    auto phase_tail = [&](const Frag &cur, Frag &nxt, int pidx) {
        const bool has_next = pidx + 1 < nt;
        if (has_next) {
            const int last_issued = (pidx + RING - 2 < nt - 1) ? pidx + RING - 2 : nt - 1;
            wait_newer(last_issued - (pidx + 1));
        }
        __builtin_amdgcn_s_barrier();
        if (pidx + RING - 1 < nt) issue(slot_fill, pidx + RING - 1);
        if (has_next) load_frag(nxt, slot_next);
        mfmas(cur);
        advance();
    };
    int pidx = 0;
    for (; pidx + RING < nt; pidx += 2) {     // both phases of the pair are in steady state
        phase_steady(fa, fb, pidx);
        phase_steady(fb, fa, pidx + 1);
    }
    for (; pidx < nt; pidx += 2) {
        phase_tail(fa, fb, pidx);
        if (pidx + 1 < nt) phase_tail(fb, fa, pidx + 1);
    }

    gemm_epilogue<TM, TN, NWAVES, RING * SLOT_BYTES, 4, EPI>(p, acc, smem, wave, lane, wm, wn, m0, nt0, split);
}

template <int EPI>
static int launch_gemm_pipe(const GemmArgs &p, hipStream_t st)
{
    constexpr int SMEM = 6 * 24 * 1024;
    auto kern = gemm_w4a8_pipe_kernel<EPI>;
    {
        const int rc = ensure_dynamic_lds((const void *)kern, SMEM);
        if (rc != MQ_OK) return rc;
    }
    GemmArgs g = p;
    set_geometry(g, 256, 256, 64, 4);
    if (!geometry_in_range(g)) return fail(MQ_EINVAL, "mq_gemm_w4a8: %u x %u x %d workgroups exceed the range of the launch-geometry arithmetic", g.m_blocks, g.n_blocks, g.splits);
    hipLaunchKernelGGL(kern, dim3(g.m_blocks * g.n_blocks * (unsigned)g.splits), dim3(512), SMEM, st, g);
    int rc = check_launch("gemm_w4a8_pipe");
    if (rc != MQ_OK || p.splits == 1) return rc;
    long blocks = ceil_div(p.M * ceil_div(p.N, 4), 256);
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(splitk_reduce_kernel<EPI>, dim3((unsigned)blocks), dim3(256), 0, st, p);
    return check_launch("splitk_reduce");
}

template <int BM, int BN, int WARPS_M, int WARPS_N, int STAGES, int W_BITS, int EPI, int DMA_POS = 1, bool GROUPED = false>
static int launch_gemm(const GemmArgs &p, hipStream_t st)
{
    constexpr int PIECES = (BM / 16) * 2 + ((W_BITS == 4) ? (BN / 16) : (BN / 16) * 2);
    constexpr int SMEM = STAGES * PIECES * 1024;
    auto kern = gemm_w4a8_kernel<BM, BN, WARPS_M, WARPS_N, STAGES, W_BITS, EPI, DMA_POS, GROUPED>;
    {
        const int rc = ensure_dynamic_lds((const void *)kern, SMEM);
        if (rc != MQ_OK) return rc;
    }
    GemmArgs g = p;
    set_geometry(g, BM, BN, 128, W_BITS);
    if (!geometry_in_range(g)) return fail(MQ_EINVAL, "mq_gemm_w4a8: %u x %u x %d workgroups exceed the range of the launch-geometry arithmetic", g.m_blocks, g.n_blocks, g.splits);
    hipLaunchKernelGGL(kern, dim3(g.m_blocks * g.n_blocks * (unsigned)g.splits),
                       dim3(WARPS_M * WARPS_N * 64), SMEM, st, g);
    int rc = check_launch("gemm_w4a8");
    if (rc != MQ_OK || p.splits == 1) return rc;
    long blocks = ceil_div(p.M * ceil_div(p.N, 4), 256);
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(splitk_reduce_kernel<EPI>, dim3((unsigned)blocks), dim3(256), 0, st, p);
    return check_launch("splitk_reduce");
}

// Tile / split-K plan, from measurements on MI355X (profiles/ and DESIGN.md):
//   * a 256x256 tile (16 waves) moves ~24 B/clk/CU through L1 at full MFMA rate against ~48 for
//     128x128, and reaches ~2.0 POP/s when the output alone fills the chip (>= 192 tiles);
//   * long reductions with few output tiles (down_proj: 42 tiles, K = 19968) are split over K
//     so that ~250 workgroups run, integer partials are combined by splitk_reduce_kernel;
//   * everything else is latency bound (~8 us floor per launch + ~0.3 us per k-step): 64x128
//     tiles put two or three workgroups on every CU and measured 10-30 % faster than 128x128.
#ifndef MQ_PLAN_WIDE_TILE
#define MQ_PLAN_WIDE_TILE 20   // the ping-pong 256 x 256 tile with the slab-free epilogue and the next tile's stages requested ahead (round 6: -2.4 % on
                               // gate|up, profiles/r6_pp_direct_epilogue_ab.txt; launches it does not cover take tile 14 inside launch_gemm_pp).
                               // A/B builds: -DMQ_PLAN_WIDE_TILE=14 (round 4/5), =13 (the software-pipelined kernel of rounds 1-3)
#endif
struct Plan {
    int tile;    // index into dispatch_tile
    int splits;
};

// epi_only: 0 = any kernel; 1 = an epilogue that exists in the wave-specialised kernels only (RoPE in the store); 2 = an activation in
// the store (GemmArgs::act): the wave-specialised tiles or the wide ping-pong tile, no split-K, no weight-streaming kernels
static Plan make_plan(long M, long N, long K_pad, bool have_ws, size_t ws_bytes, int force_tile,
                      int force_splits, bool w4 = true, bool a_tiled = false, int epi_only = 0)
{
    const bool ws_only = epi_only == 1, act_mode = epi_only == 2;
    const long kps = K_pad / 128;
    const long t256 = ceil_div(M, 256) * ceil_div(N, 256);
    Plan pl = {10, 1};
    long best = -1, best_tiles = 0;
    if (a_tiled) {
        // wave-specialised kernels (gemm_ws.hip), every operand byte arrives as contiguous KiB pieces:
        // the bytes the busiest CU has to pull decide, ceil(tiles/256) x (BM + BN/2) per unit of K
        // (round 5: the V_MFMA_I32_16X16X64_I8 twins of the round-2 tiles, ids 47 / 48 / 45 / 46 for 64 / 96 / 128 / 192 x 128 -- bit-identical
        //  results, 1-7 % faster per shape on one box: the 16x16x64 form does the same MACs with half the accumulator register traffic and
        //  the launches run at the package power limit, profiles/r5_ws_tiles_ab.txt; -DMQ_PLAN_WS_32X32 restores ids 43 / 40 / 41 / 42)
#ifdef MQ_PLAN_WS_32X32
        static const int cand[][3] = {{43, 64, 128}, {40, 96, 128}, {41, 128, 128}, {42, 192, 128}};
        constexpr int WS192 = 42;
#else
        // round 6: ids 53 / 54 / 51 / 52 = the same kernels with the slab-free epilogue (gemm_ws.hip DIRECT: -0.6 ... -1.5 us per launch,
        // profiles/r6_ws_direct_epilogue_ab.txt); launches it does not cover (split-K, RoPE, int32 / fp32 outputs ...) fall back to the
        // slab twin inside dispatch_ws.  -DMQ_PLAN_WS_SLAB restores ids 47 / 48 / 45 / 46
#ifdef MQ_PLAN_WS_SLAB
        static const int cand[][3] = {{47, 64, 128}, {48, 96, 128}, {45, 128, 128}, {46, 192, 128}};
        constexpr int WS192 = 46;
#else
        static const int cand[][3] = {{53, 64, 128}, {54, 96, 128}, {51, 128, 128}, {52, 192, 128}};
        constexpr int WS192 = 52;
#endif
#endif
        for (const auto &c : cand) {
            if (c[0] == WS192 && !w4) continue;
            const long tiles = ceil_div(M, c[1]) * ceil_div(N, c[2]);
            const long cost = ceil_div(tiles, 256) * (c[1] + (w4 ? c[2] / 2 : c[2]));
            if (best < 0 || cost < best) { best = cost; pl.tile = c[0]; best_tiles = tiles; }
        }
    } else {
        // Row-major activations: small GEMMs are bound by the row gather (~14.5 B/clk/CU out of L2,
        // tools/probes/l2_row_stride.hip); same cost model over the symmetric kernels.
        static const int cand[][3] = {{10, 64, 128}, {31, 96, 128}, {26, 128, 128}, {35, 192, 128}, {2, 256, 128}};
        for (const auto &c : cand) {
            if (c[0] == 35 && !w4) continue;
            const long tiles = ceil_div(M, c[1]) * ceil_div(N, c[2]);
            const long cost = ceil_div(tiles, 256) * (c[1] + (w4 ? c[2] / 2 : c[2]));   // int8 weights: one byte each
            if (best < 0 || cost < best) { best = cost; pl.tile = c[0]; }
        }
    }
    if (ws_only) {
        // (an epilogue that exists in the wave-specialised kernels only -- RoPE in the store: the best of the tiles above, no split-K)
    } else if (!act_mode && a_tiled && w4 && ceil_div(N, 128) <= 65535 &&
               (M <= 16 || (M <= 32 && K_pad >= 512 && K_pad <= 4096 && N >= 2048) || (M <= 64 && kps >= 64 && ceil_div(N, 128) < 128))) {
        // A few rows (generation steps of the exam scripts): the weight stream is the work -- gemm_skinny.hip, id 60; `splits` = its
        // K slices, added up by splitk_reduce_kernel.  One row tile: every decoder shape gains (48 against 80 us per layer at M = 1);
        // two to four row tiles: only the long reductions over few channel tiles do (down_proj: 23 against 36 us at M = 32) --
        // profiles/r5_decode_gemm_bench.txt
        pl.tile = 60;
        pl.splits = have_ws ? skinny_slices(M, N, K_pad, ws_bytes) : 1;
        if (M <= 32 && K_pad >= 512 && K_pad <= 4096 && N >= 2048) {   // short reduction, enough pairs: the slices are the eight waves of a workgroup
            pl.tile = 61;
            pl.splits = 1;
        } else if (M <= 32 && K_pad > 4096 && have_ws) {               // long reduction: a few workgroup slices of eight waves each
            pl.tile = 61;
            pl.splits = skinny_wg_slices(M, N, K_pad, ws_bytes);
        }
    } else if (act_mode && !(w4 && a_tiled)) {
        // (int8 weights with an activation in the store: the wave-specialised tile chosen above; the 16-wave kernel has no act path)
    } else if (t256 >= 192) {
        // gate|up: with tiled activations the 8-wave ping-pong kernel (gemm_pp.hip, round 4: 93-101 us against 108-116
        // for the software-pipelined tile 13 and 110-112 for the 16-wave tile 3, profiles/r4_pp_ab.txt); with row-major
        // activations the 16-wave kernel
        pl.tile = (w4 && a_tiled) ? MQ_PLAN_WIDE_TILE : 3;
        // ... unless a handful of 256 x 256 tiles spill into one more round (Qwen-VL w1|w2: 258 tiles; InternVL2 wqkv at
        // batch 4: 288): a round of the 256^2 kernel is ~57 us whether 2 or 256 tiles run in it.  Then the 192 x 128
        // wave-specialised tile is compared on measured per-tile times (us: f0 + c per 128-deep k-step; a partial last
        // round costs 0.8 .. 1.0 of a full one; profiles/r3_gemm_plan_spill.txt): 101 -> 78 us and 108 -> 82 us there.
        const long full = t256 / 256, rem = t256 % 256;
        if (w4 && a_tiled && full >= 1 && rem > 0 && rem < 64) {
            auto rounds = [](long tiles) { const long f = tiles / 256, r = tiles % 256; return (float)f + (r ? 0.8f + 0.2f * (float)r / 256.0f : 0.0f); };
            const float t_pipe = rounds(t256) * (4.0f + 1.65f * (float)kps);
            const float t_192 = rounds(ceil_div(M, 192) * ceil_div(N, 128)) * (7.8f + 0.55f * (float)kps);
#ifdef MQ_PLAN_WS_32X32
            if (t_192 < t_pipe) pl.tile = 42;
#elif defined(MQ_PLAN_WS_SLAB)
            if (t_192 < t_pipe) pl.tile = 46;
#else
            if (t_192 < t_pipe) pl.tile = 52;
#endif
        }
    } else if (a_tiled && best >= 0 && ceil_div(M, 96) * ceil_div(N, 128) >= 128) {
        // enough 96..192 x 128 tiles for most CUs: the wave-specialised kernel walks the whole reduction
        // (down_proj, K = 19968: 64 us against 60 + 14 us for split-K partials plus the reduce kernel)
    } else if (have_ws && kps >= 64 && a_tiled && best_tiles > 0 && M <= 256) {
        // few tiles and a long reduction on tiled activations (down_proj of a short prompt): split K over the wave-specialised tile the
        // cost model chose, one round of workgroups (each fills a CU's LDS), at least eight k-steps each (M = 65 / 128 / 256: 36 / 38 / 44.5 us on the 16-wave tile
        // below -> 19.7 / 21.7 / 31.4 us: profiles/r5_midm_gemm_bench.txt)
        long s = (long)device_cu_count() / best_tiles;      // ONE round: a workgroup of these kernels fills a CU's LDS
        if (s > 16) s = 16;
        while (s > 1 && ((size_t)(s * M * N * 4) > ws_bytes || kps / s < 8)) --s;
        if (s > 1) pl.splits = (int)s;
    } else if (have_ws && kps >= 64 && t256 >= 8) {
        long s = (252 + t256 - 1) / t256;
        if (s > 8) s = 8;
        while (s > 1 && ((size_t)(s * M * N * 4) > ws_bytes || kps / s < 8)) --s;
        if (s > 1) { pl.tile = 3; pl.splits = (int)s; }
    }
    if (force_tile >= 0 && (!ws_only || (force_tile >= 40 && force_tile < 60))
        && (!act_mode || (force_tile >= 40 && force_tile < 60) || force_tile == 14 || force_tile == 19 || force_tile == 20)) {
        if ((pl.tile == 60 || pl.tile == 61) && force_tile != pl.tile) pl.splits = 1;   // (the slices were the skinny kernel's)
        pl.tile = force_tile;
    }
    if (force_splits > 0 && !ws_only && !act_mode) pl.splits = force_splits;
    if (pl.splits > kps) pl.splits = (int)kps;
    if (pl.splits < 1) pl.splits = 1;
    return pl;
}

// Shapes that were measured and dropped (no gain): LDS-DMA issue positions, 2x8 / 8x2
// wave layouts, rings of 4-8 stages, 64x256, 128x64, 128x256.
template <int W_BITS, int EPI>
static int dispatch_tile(const GemmArgs &p, int tile, hipStream_t st)
{
    if (tile >= 40 && tile < 60) {
        if (!p.a_tiled) return fail(MQ_EINVAL, "mq_gemm_w4a8: tile %d needs activations in the tiled layout (lda = MQ_LD_TILED)", tile);
        const int rc = dispatch_ws<W_BITS, EPI>(p, tile, st);
        if (rc != MQ_OK || p.splits == 1) return rc;
        long blocks = ceil_div(p.M * ceil_div(p.N, 4), 256);
        if (blocks > 2048) blocks = 2048;
        hipLaunchKernelGGL(splitk_reduce_kernel<EPI>, dim3((unsigned)blocks), dim3(256), 0, st, p);
        return check_launch("splitk_reduce");
    }
    switch (tile) {
    case 1:
        if constexpr (W_BITS == 4) return launch_gemm<256, 256, 2, 4, 3, W_BITS, EPI>(p, st);
        else return launch_gemm<256, 128, 4, 2, 3, W_BITS, EPI>(p, st);  // int8 weights: 3 x 64 KiB would not fit
    case 2: return launch_gemm<256, 128, 4, 2, 3, W_BITS, EPI>(p, st);
    case 3:
        if constexpr (W_BITS == 4) return launch_gemm<256, 256, 4, 4, 3, W_BITS, EPI>(p, st);
        else return launch_gemm<256, 128, 4, 2, 3, W_BITS, EPI>(p, st);
    case 4: return launch_gemm<128, 256, 2, 4, 3, W_BITS, EPI>(p, st);
    case 5: return launch_gemm<256, 128, 2, 4, 3, W_BITS, EPI>(p, st);
    case 13: if constexpr (W_BITS == 4) return launch_gemm_pipe<EPI>(p, st); else break;
    case 14: case 15: case 16: case 17: case 18: case 19: case 20:
        if constexpr (W_BITS == 4) {
            if (!p.a_tiled) return fail(MQ_EINVAL, "mq_gemm_w4a8: tile %d needs activations in the tiled layout (lda = MQ_LD_TILED)", tile);
            const int rc = launch_gemm_pp<EPI>(p, tile, st);
            if (rc != MQ_OK || p.splits == 1) return rc;
            long blocks = ceil_div(p.M * ceil_div(p.N, 4), 256);
            if (blocks > 2048) blocks = 2048;
            hipLaunchKernelGGL(splitk_reduce_kernel<EPI>, dim3((unsigned)blocks), dim3(256), 0, st, p);
            return check_launch("splitk_reduce");
        } else break;
    case 60:
        if constexpr (W_BITS == 4) {
            const int rc = launch_gemm_skinny<EPI>(p, st);
            if (rc != MQ_OK || p.splits == 1) return rc;
            long blocks = ceil_div(p.M * ceil_div(p.N, 4), 256);
            if (blocks > 2048) blocks = 2048;
            hipLaunchKernelGGL(splitk_reduce_kernel<EPI>, dim3((unsigned)blocks), dim3(256), 0, st, p);
            return check_launch("splitk_reduce");
        } else break;
    case 61:
        if constexpr (W_BITS == 4) {
            const int rc = launch_gemm_skinny_wg<EPI>(p, st);
            if (rc != MQ_OK || p.splits == 1) return rc;
            long blocks = ceil_div(p.M * ceil_div(p.N, 4), 256);
            if (blocks > 2048) blocks = 2048;
            hipLaunchKernelGGL(splitk_reduce_kernel<EPI>, dim3((unsigned)blocks), dim3(256), 0, st, p);
            return check_launch("splitk_reduce");
        } else break;
    case 10: return launch_gemm<64, 128, 2, 2, 3, W_BITS, EPI>(p, st);
    case 11: return launch_gemm<128, 64, 2, 2, 3, W_BITS, EPI>(p, st);
    case 12: return launch_gemm<128, 128, 4, 2, 3, W_BITS, EPI>(p, st);
    case 26: return launch_gemm<128, 128, 2, 4, 3, W_BITS, EPI>(p, st);
    case 31: return launch_gemm<96, 128, 1, 4, 3, W_BITS, EPI>(p, st);
    case 35: if constexpr (W_BITS == 4) return launch_gemm<192, 128, 2, 4, 3, W_BITS, EPI>(p, st); else break;
    default: break;
    }
    return launch_gemm<128, 128, 2, 2, 3, W_BITS, EPI>(p, st);
}

// test / tuning overrides of the plan (mq_gemm_debug_force): thread-local, so they change the dispatch of the
// calling thread only and the library stays re-entrant for everybody else
static thread_local int g_force_tile = -1, g_force_splits = 0;
thread_local int g_gemm_force_xm = 0;

static int gemm_common(const int8_t *a, long lda, const void *w, int w_bits, long M, long N,
                       long K_pad, float s_x0, float s_x1, const uint8_t *row_sel,
                       const float *s_w, const float *bias, const float *x0, const float *w0,
                       void *out, int epi, long ldo, void *workspace, size_t workspace_bytes,
                       void *stream, const float *sx_vec = nullptr, const void *residual = nullptr,
                       long ldr = 0, const float *sx_groups = nullptr, long n_groups = 0, int group_k = 0,
                       const float *x1 = nullptr, const float *w1 = nullptr, const float *shift_groups = nullptr,
                       const float *wsum_groups = nullptr, const float *sw_groups = nullptr,
                       const void *rope_cos = nullptr, const void *rope_sin = nullptr, long rope_cols = 0, int act = MQ_ACT_NONE)
{
    MQ_REQUIRE(M >= 0 && N >= 0 && K_pad >= 0, "mq_gemm_w4a8: negative shape");
    if (M == 0 || N == 0) return MQ_OK;
    MQ_REQUIRE(a && w && out, "mq_gemm_w4a8: null buffer");
    MQ_REQUIRE(K_pad > 0 && K_pad % 128 == 0, "mq_gemm_w4a8: K_pad=%ld must be a positive multiple of 128", K_pad);
    const bool a_tiled = (lda == MQ_LD_TILED);
    MQ_REQUIRE(((uintptr_t)a) % 16 == 0 && (a_tiled || (lda >= K_pad && lda % 16 == 0)),
               "mq_gemm_w4a8: activations must be 16-byte aligned with lda %% 16 == 0 and lda >= K_pad (or lda = MQ_LD_TILED)");
    MQ_REQUIRE(((uintptr_t)w) % 16 == 0, "mq_gemm_w4a8: weight image must be 16-byte aligned");
    MQ_REQUIRE(w_bits == 4 || w_bits == 8, "mq_gemm_w4a8: w_bits must be 4 or 8");
    MQ_REQUIRE(ldo >= (act == MQ_ACT_SILU_MUL ? N / 2 : N), "mq_gemm_w4a8: ldo < N");
    MQ_REQUIRE(epi == EPI_I32 || s_w || sw_groups, "mq_gemm_w4a8: s_w is required");
    MQ_REQUIRE((x0 == nullptr) == (w0 == nullptr), "mq_gemm_w4a8: x0 and w0 go together");
    MQ_REQUIRE((x1 == nullptr) == (w1 == nullptr) && (x1 == nullptr || epi != EPI_I32), "mq_gemm_w4a8: x1 and w1 go together (floating-point outputs)");
    // int32 headroom: |acc| <= K * 128 * 8 * 16 (int4 in the high nibble) or K * 128 * 128
    MQ_REQUIRE(K_pad <= 131072L, "mq_gemm_w4a8: K too large for int32 accumulation");
    MQ_REQUIRE(!workspace || ((uintptr_t)workspace) % 16 == 0, "mq_gemm_w4a8: workspace must be 16-byte aligned");
    GemmArgs p;
    p.a = a; p.lda = lda; p.a_tiled = a_tiled; p.w = (const uint8_t *)w; p.M = M; p.N = N; p.K_pad = K_pad;
    p.n_tiles = ceil_div(N, 16);
    p.n_pairs = ceil_div(N, 32);
    p.sx0 = s_x0; p.sx1 = s_x1; p.row_sel = row_sel; p.s_w = s_w; p.bias = bias; p.x0 = x0; p.w0 = w0;
    p.sx_vec = sx_vec;
    p.x1 = x1; p.w1 = w1;
    p.residual = residual; p.ldr = ldr;
    MQ_REQUIRE(!residual || (epi != EPI_I32 && ldr >= N), "mq_gemm_w4a8: bad residual geometry");
    p.res_vec = residual && (((uintptr_t)residual) % 16 == 0) && ((ldr * ((epi == EPI_F32) ? 4 : 2)) % 16 == 0);
    p.out = out; p.ldo = ldo;
    if (sx_groups || sw_groups) {
        // group-wise activation and / or weight scales: the symmetric 128 x 128 kernel, no split-K, floating-point outputs only
        MQ_REQUIRE(epi != EPI_I32 && group_k > 0 && (group_k == 64 || group_k % 128 == 0) && K_pad % 64 == 0 && !x1 && (!sx_vec || (sw_groups && !sx_groups)),
                   "mq_gemm_w4a8_groupscale: group size %d (64 or a multiple of 128)", group_k);
        MQ_REQUIRE((shift_groups == nullptr) == (wsum_groups == nullptr), "mq_gemm_w4a8_groupscale: shift_groups and wsum_groups go together");
        MQ_REQUIRE(!(sw_groups && shift_groups), "mq_gemm_w4a8_wgroupscale: asymmetric activation groups are not combined with weight groups");
        p.sx_groups = sx_groups; p.sw_groups = sw_groups; p.n_groups = n_groups; p.group_k = group_k; p.acc_float = 1;
        p.shift_groups = shift_groups; p.wsum_groups = wsum_groups;
        p.splits = 1; p.partial = nullptr;
        p.vec_ok = (N % 8 == 0) && (ldo % 8 == 0) && (((uintptr_t)out) % 16 == 0);
        auto al = [](const void *q) { return q == nullptr || ((uintptr_t)q) % 16 == 0; };
        p.par_ok = al(s_w) && al(bias) && al(w0);
        hipStream_t gst = (hipStream_t)stream;
        if (a_tiled && !shift_groups && (!sw_groups || (N % 4 == 0 && ((uintptr_t)sw_groups) % 16 == 0)) && g_force_tile != 26) {
            // symmetric group scales (weights, activations or both) on tiled activations, groups of 64 or of whole k-steps: the fold rides in the
            // wave-specialised 16x16x64 kernels (5.5-9 x faster than the round-1 kernel below, profiles/r5_wgroup_gemm_*.txt);
            // mq_gemm_debug_force(26) keeps the round-1 kernel for A/B
            const Plan gpl = make_plan(M, N, K_pad, false, 0, g_force_tile, 0, w_bits == 4, true, true);
            if (w_bits == 4) {
                switch (epi) {
                case EPI_F16: return dispatch_ws_wgroup<4, EPI_F16>(p, gpl.tile, gst);
                case EPI_BF16: return dispatch_ws_wgroup<4, EPI_BF16>(p, gpl.tile, gst);
                default: return dispatch_ws_wgroup<4, EPI_F32>(p, gpl.tile, gst);
                }
            }
            switch (epi) {
            case EPI_F16: return dispatch_ws_wgroup<8, EPI_F16>(p, gpl.tile, gst);
            case EPI_BF16: return dispatch_ws_wgroup<8, EPI_BF16>(p, gpl.tile, gst);
            default: return dispatch_ws_wgroup<8, EPI_F32>(p, gpl.tile, gst);
            }
        }
        if (w_bits == 4) {
            switch (epi) {
            case EPI_F16: return launch_gemm<128, 128, 2, 4, 3, 4, EPI_F16, 1, true>(p, gst);
            case EPI_BF16: return launch_gemm<128, 128, 2, 4, 3, 4, EPI_BF16, 1, true>(p, gst);
            default: return launch_gemm<128, 128, 2, 4, 3, 4, EPI_F32, 1, true>(p, gst);
            }
        }
        switch (epi) {
        case EPI_F16: return launch_gemm<128, 128, 2, 4, 3, 8, EPI_F16, 1, true>(p, gst);
        case EPI_BF16: return launch_gemm<128, 128, 2, 4, 3, 8, EPI_BF16, 1, true>(p, gst);
        default: return launch_gemm<128, 128, 2, 4, 3, 8, EPI_F32, 1, true>(p, gst);
        }
    }
    if (act != MQ_ACT_NONE) {
        // the activation lives in the act paths of the wave-specialised / ping-pong epilogues: everything they rely on is checked HERE
        MQ_REQUIRE(act == MQ_ACT_SILU_MUL || act == MQ_ACT_QUICK_GELU, "mq_gemm_w4a8_act_ws: unknown activation %d", act);
        MQ_REQUIRE(a_tiled && epi != EPI_I32 && !residual && !x0 && !x1 && !sx_groups && !sw_groups && !rope_cos,
                   "mq_gemm_w4a8_act_ws: needs tiled activations (lda = MQ_LD_TILED), a floating-point output and no residual / rank-1 / group / RoPE terms");
        const long n_out = act == MQ_ACT_SILU_MUL ? N / 2 : N;
        MQ_REQUIRE(act != MQ_ACT_SILU_MUL || N % 64 == 0, "mq_gemm_w4a8_act_ws: silu(gate)*up needs N = 2 x (a multiple of 32) channels, gate then up (N = %ld)", N);
        MQ_REQUIRE(n_out % 8 == 0 && ldo % 8 == 0 && ((uintptr_t)out) % 16 == 0, "mq_gemm_w4a8_act_ws: output columns and ldo must be multiples of 8, out 16-byte aligned");
        p.act = act;
    }
    const bool rope = rope_cos != nullptr;
    if (rope) {
        // RoPE in the store lives in the fast path of the wave-specialised epilogue: everything that path needs is checked HERE
        // (a launch that fell back to the general loop would silently skip the rotation)
        MQ_REQUIRE(rope_sin && a_tiled && (epi == EPI_F16 || epi == EPI_BF16) && !residual && rope_cols > 0 && rope_cols % 128 == 0 && rope_cols <= N
                       && N % 8 == 0 && ldo % 8 == 0 && ((uintptr_t)out) % 16 == 0 && ((uintptr_t)rope_cos) % 16 == 0 && ((uintptr_t)rope_sin) % 16 == 0,
                   "mq_gemm_w4a8_rope_ws: needs tiled activations, a 16-bit 16-byte aligned output with N and ldo multiples of 8, heads of 128 (rope_cols a multiple of 128, at most N) and 16-byte aligned tables");
        p.rope_cos = rope_cos; p.rope_sin = rope_sin; p.rope_cols = rope_cols;
    }
    Plan pl = make_plan(M, N, K_pad, workspace != nullptr && !rope && act == MQ_ACT_NONE, workspace_bytes, g_force_tile,
                        workspace ? g_force_splits : 0, w_bits == 4, a_tiled, rope ? 1 : (act != MQ_ACT_NONE ? 2 : 0));
    if (act != MQ_ACT_NONE && !(pl.tile >= 40 && pl.tile < 60) && pl.tile != 14 && pl.tile != 19 && pl.tile != 20) {
        // the activation epilogue exists in the 256-wide ping-pong tiles (a wave holds a gate pair AND its up pair) and in the
        // wave-specialised kernels: anything else the plan came up with takes the best wave-specialised tile
        pl.tile = make_plan(M, N, K_pad, false, 0, -1, 0, w_bits == 4, a_tiled, 1).tile;
    }
    p.splits = pl.splits;
    p.partial = (int32_t *)workspace;
    auto al16 = [](const void *q) { return q == nullptr || ((uintptr_t)q) % 16 == 0; };
    const size_t osz = (epi == EPI_F16 || epi == EPI_BF16) ? 2 : 4;
    p.vec_ok = (N % 8 == 0) && (ldo % 8 == 0) && (((uintptr_t)out) % 16 == 0);
    p.par_ok = al16(s_w) && al16(bias) && al16(w0) && al16(w1);
    (void)osz;
    if (p.splits > 1)
        MQ_REQUIRE((size_t)p.splits * M * N * 4 <= workspace_bytes, "mq_gemm_w4a8: workspace too small for split-K");
    hipStream_t st = (hipStream_t)stream;
    if (w_bits == 4) {
        switch (epi) {
        case EPI_F16: return dispatch_tile<4, EPI_F16>(p, pl.tile, st);
        case EPI_BF16: return dispatch_tile<4, EPI_BF16>(p, pl.tile, st);
        case EPI_F32: return dispatch_tile<4, EPI_F32>(p, pl.tile, st);
        case EPI_I32: return dispatch_tile<4, EPI_I32>(p, pl.tile, st);
        }
    } else {
        switch (epi) {
        case EPI_F16: return dispatch_tile<8, EPI_F16>(p, pl.tile, st);
        case EPI_BF16: return dispatch_tile<8, EPI_BF16>(p, pl.tile, st);
        case EPI_F32: return dispatch_tile<8, EPI_F32>(p, pl.tile, st);
        case EPI_I32: return dispatch_tile<8, EPI_I32>(p, pl.tile, st);
        }
    }
    return fail(MQ_EINVAL, "mq_gemm_w4a8: unknown output dtype %d", epi);
}

}  // namespace mq

// TEST-ONLY: the (tile, split-K) plan the dispatcher would take for a shape (host arithmetic, no device needed).
extern "C" int mq_gemm_debug_plan(long M, long N, long K_pad, int w_bits, int a_tiled, int have_workspace, int *tile, int *splits)
{
    if (!tile || !splits || K_pad <= 0 || K_pad % 128) return mq::fail(MQ_EINVAL, "mq_gemm_debug_plan: bad arguments");
    const mq::Plan pl = mq::make_plan(M, N, K_pad, have_workspace != 0, have_workspace ? ((size_t)1 << 40) : 0, -1, 0, w_bits == 4, a_tiled != 0);
    *tile = pl.tile;
    *splits = pl.splits;
    return MQ_OK;
}

extern "C" int mq_gemm_w4a8(const int8_t *a, long lda, const void *w, int w_bits, long M, long N,
                            long K_pad, float s_x0, float s_x1, const uint8_t *row_sel,
                            const float *s_w, const float *bias, const float *x0, const float *w0,
                            void *out, int out_dtype, long ldo, void *stream)
{
    if (out_dtype != MQ_F16 && out_dtype != MQ_BF16 && out_dtype != MQ_F32)
        return mq::fail(MQ_EINVAL, "mq_gemm_w4a8: unknown output dtype %d", out_dtype);
    return mq::gemm_common(a, lda, w, w_bits, M, N, K_pad, s_x0, s_x1, row_sel, s_w, bias, x0, w0,
                           out, out_dtype, ldo, nullptr, 0, stream);
}

extern "C" int mq_gemm_w4a8_ws(const int8_t *a, long lda, const void *w, int w_bits, long M, long N,
                               long K_pad, float s_x0, float s_x1, const uint8_t *row_sel,
                               const float *s_w, const float *bias, const float *x0,
                               const float *w0, void *out, int out_dtype, long ldo,
                               void *workspace, size_t workspace_bytes, void *stream)
{
    if (out_dtype != MQ_F16 && out_dtype != MQ_BF16 && out_dtype != MQ_F32)
        return mq::fail(MQ_EINVAL, "mq_gemm_w4a8_ws: unknown output dtype %d", out_dtype);
    return mq::gemm_common(a, lda, w, w_bits, M, N, K_pad, s_x0, s_x1, row_sel, s_w, bias, x0, w0,
                           out, out_dtype, ldo, workspace, workspace_bytes, stream);
}

extern "C" int mq_gemm_w4a8_residual_ws(const int8_t *a, long lda, const void *w, int w_bits, long M, long N,
                                        long K_pad, float s_x0, float s_x1, const uint8_t *row_sel,
                                        const float *s_w, const float *bias, const float *x0, const float *w0,
                                        const void *residual, long ldr, void *out, int out_dtype, long ldo,
                                        void *workspace, size_t workspace_bytes, void *stream)
{
    if (out_dtype != MQ_F16 && out_dtype != MQ_BF16 && out_dtype != MQ_F32)
        return mq::fail(MQ_EINVAL, "mq_gemm_w4a8_residual_ws: unknown output dtype %d", out_dtype);
    if (M == 0 || N == 0) return MQ_OK;
    if (!residual) return mq::fail(MQ_EINVAL, "mq_gemm_w4a8_residual_ws: residual is required");
    return mq::gemm_common(a, lda, w, w_bits, M, N, K_pad, s_x0, s_x1, row_sel, s_w, bias, x0, w0,
                           out, out_dtype, ldo, workspace, workspace_bytes, stream, nullptr, residual, ldr);
}

extern "C" int mq_gemm_w4a8_rope_ws(const int8_t *a, long lda, const void *w, int w_bits, long M, long N, long K_pad, float s_x0, float s_x1,
                                    const uint8_t *row_sel, const float *s_w, const float *bias, const void *rope_cos, const void *rope_sin,
                                    long rope_cols, int head_dim, void *out, int out_dtype, long ldo, void *stream)
{
    if (out_dtype != MQ_F16 && out_dtype != MQ_BF16)
        return mq::fail(MQ_EINVAL, "mq_gemm_w4a8_rope_ws: the rotation is defined on a 16-bit output (dtype %d)", out_dtype);
    if (head_dim != 128) return mq::fail(MQ_EUNSUPPORTED, "mq_gemm_w4a8_rope_ws: head_dim %d (the fused form needs 128 = one output tile; use mq_rope_inplace)", head_dim);
    if (M == 0 || N == 0) return MQ_OK;
    if (!rope_cos || !rope_sin) return mq::fail(MQ_EINVAL, "mq_gemm_w4a8_rope_ws: the cos / sin tables are required");
    return mq::gemm_common(a, lda, w, w_bits, M, N, K_pad, s_x0, s_x1, row_sel, s_w, bias, nullptr, nullptr, out, out_dtype, ldo,
                           nullptr, 0, stream, nullptr, nullptr, 0, nullptr, 0, 0, nullptr, nullptr, nullptr, nullptr, nullptr,
                           rope_cos, rope_sin, rope_cols);
}

extern "C" int mq_gemm_w4a8_act_ws(const int8_t *a, long lda, const void *w, int w_bits, long M, long N, long K_pad, float s_x0, float s_x1,
                                   const uint8_t *row_sel, const float *s_x_rows, const float *s_w, const float *bias, int act,
                                   void *out, int out_dtype, long ldo, void *stream)
{
    if (out_dtype != MQ_F16 && out_dtype != MQ_BF16 && out_dtype != MQ_F32)
        return mq::fail(MQ_EINVAL, "mq_gemm_w4a8_act_ws: unknown output dtype %d", out_dtype);
    if (M == 0 || N == 0) return MQ_OK;
    if (lda != MQ_LD_TILED) return mq::fail(MQ_EUNSUPPORTED, "mq_gemm_w4a8_act_ws: needs activations in the tiled layout (lda = MQ_LD_TILED)");
    return mq::gemm_common(a, lda, w, w_bits, M, N, K_pad, s_x_rows ? 1.0f : s_x0, s_x_rows ? 1.0f : s_x1, s_x_rows ? nullptr : row_sel, s_w, bias,
                           nullptr, nullptr, out, out_dtype, ldo, nullptr, 0, stream, s_x_rows, nullptr, 0, nullptr, 0, 0, nullptr, nullptr,
                           nullptr, nullptr, nullptr, nullptr, nullptr, 0, act);
}

extern "C" int mq_gemm_w4a8_rowscale_ws(const int8_t *a, long lda, const void *w, int w_bits, long M, long N,
                                        long K_pad, const float *s_x_rows, const float *s_w, const float *bias,
                                        const float *x0, const float *w0, void *out, int out_dtype, long ldo,
                                        void *workspace, size_t workspace_bytes, void *stream)
{
    if (out_dtype != MQ_F16 && out_dtype != MQ_BF16 && out_dtype != MQ_F32)
        return mq::fail(MQ_EINVAL, "mq_gemm_w4a8_rowscale_ws: unknown output dtype %d", out_dtype);
    if (M == 0 || N == 0) return MQ_OK;
    if (!s_x_rows) return mq::fail(MQ_EINVAL, "mq_gemm_w4a8_rowscale_ws: s_x_rows is required");
    return mq::gemm_common(a, lda, w, w_bits, M, N, K_pad, 1.0f, 1.0f, nullptr, s_w, bias, x0, w0,
                           out, out_dtype, ldo, workspace, workspace_bytes, stream, s_x_rows);
}

extern "C" int mq_gemm_w4a8_rank2_ws(const int8_t *a, long lda, const void *w, int w_bits, long M, long N, long K_pad,
                                     float s_x0, float s_x1, const uint8_t *row_sel, const float *s_x_rows,
                                     const float *s_w, const float *bias, const float *x0, const float *w0,
                                     const float *x1, const float *w1, void *out, int out_dtype, long ldo,
                                     void *workspace, size_t workspace_bytes, void *stream)
{
    if (out_dtype != MQ_F16 && out_dtype != MQ_BF16 && out_dtype != MQ_F32)
        return mq::fail(MQ_EINVAL, "mq_gemm_w4a8_rank2_ws: unknown output dtype %d", out_dtype);
    if (M == 0 || N == 0) return MQ_OK;
    if (!x0 || !w0 || !x1 || !w1) return mq::fail(MQ_EINVAL, "mq_gemm_w4a8_rank2_ws: both rank-1 terms are required (one term: mq_gemm_w4a8_ws)");
    return mq::gemm_common(a, lda, w, w_bits, M, N, K_pad, s_x_rows ? 1.0f : s_x0, s_x_rows ? 1.0f : s_x1, s_x_rows ? nullptr : row_sel,
                           s_w, bias, x0, w0, out, out_dtype, ldo, workspace, workspace_bytes, stream, s_x_rows, nullptr, 0,
                           nullptr, 0, 0, x1, w1);
}

extern "C" int mq_gemm_w4a8_groupscale(const int8_t *a, long lda, const void *w, int w_bits, long M, long N,
                                       long K_pad, const float *s_x_groups, long n_groups, int group_k, const float *s_w,
                                       const float *bias, void *out, int out_dtype, long ldo, void *stream)
{
    if (out_dtype != MQ_F16 && out_dtype != MQ_BF16 && out_dtype != MQ_F32)
        return mq::fail(MQ_EINVAL, "mq_gemm_w4a8_groupscale: unknown output dtype %d", out_dtype);
    if (M == 0 || N == 0) return MQ_OK;
    if (!s_x_groups || n_groups <= 0) return mq::fail(MQ_EINVAL, "mq_gemm_w4a8_groupscale: s_x_groups is required");
    if ((long)group_k * n_groups > K_pad || (long)group_k * n_groups + 127 < K_pad)
        return mq::fail(MQ_EINVAL, "mq_gemm_w4a8_groupscale: %ld groups of %d do not cover K_pad=%ld", n_groups, group_k, K_pad);
    return mq::gemm_common(a, lda, w, w_bits, M, N, K_pad, 1.0f, 1.0f, nullptr, s_w, bias, nullptr, nullptr,
                           out, out_dtype, ldo, nullptr, 0, stream, nullptr, nullptr, 0, s_x_groups, n_groups, group_k);
}

extern "C" int mq_gemm_w4a8_groupscale_asym(const int8_t *a, long lda, const void *w, int w_bits, long M, long N, long K_pad,
                                            const float *s_x_groups, const float *shift_groups, const float *wsum_groups,
                                            long n_groups, int group_k, const float *s_w, const float *bias, void *out,
                                            int out_dtype, long ldo, void *stream)
{
    if (out_dtype != MQ_F16 && out_dtype != MQ_BF16 && out_dtype != MQ_F32)
        return mq::fail(MQ_EINVAL, "mq_gemm_w4a8_groupscale_asym: unknown output dtype %d", out_dtype);
    if (M == 0 || N == 0) return MQ_OK;
    if (!s_x_groups || !shift_groups || !wsum_groups || n_groups <= 0)
        return mq::fail(MQ_EINVAL, "mq_gemm_w4a8_groupscale_asym: s_x_groups, shift_groups and wsum_groups are required");
    if ((long)group_k * n_groups > K_pad || (long)group_k * n_groups + 127 < K_pad)
        return mq::fail(MQ_EINVAL, "mq_gemm_w4a8_groupscale_asym: %ld groups of %d do not cover K_pad=%ld", n_groups, group_k, K_pad);
    return mq::gemm_common(a, lda, w, w_bits, M, N, K_pad, 1.0f, 1.0f, nullptr, s_w, bias, nullptr, nullptr,
                           out, out_dtype, ldo, nullptr, 0, stream, nullptr, nullptr, 0, s_x_groups, n_groups, group_k,
                           nullptr, nullptr, shift_groups, wsum_groups);
}

extern "C" int mq_gemm_w4a8_wgroupscale(const int8_t *a, long lda, const void *w, int w_bits, long M, long N, long K_pad,
                                        const float *s_w_groups, long n_groups, int group_k, float s_x0, float s_x1,
                                        const uint8_t *row_sel, const float *s_x_rows, const float *s_x_groups,
                                        const float *bias, void *out, int out_dtype, long ldo, void *stream)
{
    if (out_dtype != MQ_F16 && out_dtype != MQ_BF16 && out_dtype != MQ_F32)
        return mq::fail(MQ_EINVAL, "mq_gemm_w4a8_wgroupscale: unknown output dtype %d", out_dtype);
    if (M == 0 || N == 0) return MQ_OK;
    if (!s_w_groups || n_groups <= 0) return mq::fail(MQ_EINVAL, "mq_gemm_w4a8_wgroupscale: s_w_groups is required");
    if ((long)group_k * n_groups > K_pad || (long)group_k * n_groups + 127 < K_pad)
        return mq::fail(MQ_EINVAL, "mq_gemm_w4a8_wgroupscale: %ld groups of %d do not cover K_pad=%ld", n_groups, group_k, K_pad);
    if (s_x_groups && (s_x_rows || row_sel))
        return mq::fail(MQ_EINVAL, "mq_gemm_w4a8_wgroupscale: group-wise activation scales exclude per-row / per-token-type scales");
    const bool unit = s_x_groups || s_x_rows;           // the row scale is elsewhere: x 1.0 in the epilogue is exact
    return mq::gemm_common(a, lda, w, w_bits, M, N, K_pad, unit ? 1.0f : s_x0, unit ? 1.0f : s_x1, unit ? nullptr : row_sel, nullptr, bias,
                           nullptr, nullptr, out, out_dtype, ldo, nullptr, 0, stream, s_x_groups ? nullptr : s_x_rows, nullptr, 0,
                           s_x_groups, n_groups, group_k, nullptr, nullptr, nullptr, nullptr, s_w_groups);
}

extern "C" int mq_gemm_w4a8_i32(const int8_t *a, long lda, const void *w, int w_bits, long M,
                                long N, long K_pad, int32_t *acc, long ldacc, void *stream)
{
    return mq::gemm_common(a, lda, w, w_bits, M, N, K_pad, 1.0f, 1.0f, nullptr, nullptr, nullptr,
                           nullptr, nullptr, acc, mq::EPI_I32, ldacc, nullptr, 0, stream);
}

extern "C" int mq_gemm_w4a8_i32_ws(const int8_t *a, long lda, const void *w, int w_bits, long M,
                                   long N, long K_pad, int32_t *acc, long ldacc, void *workspace,
                                   size_t workspace_bytes, void *stream)
{
    return mq::gemm_common(a, lda, w, w_bits, M, N, K_pad, 1.0f, 1.0f, nullptr, nullptr, nullptr,
                           nullptr, nullptr, acc, mq::EPI_I32, ldacc, workspace, workspace_bytes,
                           stream);
}

// TEST-ONLY hook: force a tile shape (-1 = heuristic; the ids are the cases of dispatch_tile /
// dispatch_ws) and a split-K factor (0 = heuristic) for the CALLING THREAD's later GEMM calls.
// Not part of the drop-in surface.
extern "C" int mq_gemm_debug_force(int tile, int splits)
{
    mq::g_force_tile = tile;
    mq::g_force_splits = splits > 0 ? (splits & 0xff) : 0;
    mq::g_gemm_force_xm = splits > 0 ? (splits >> 8) & 0xff : 0;
    mq::g_pp_act_slab = splits > 0 ? (splits >> 16) & 1 : 0;
    return MQ_OK;
}
