// gemm_common.h -- pieces shared by the W4A8 GEMM kernels (gemm_w4a8.hip, gemm_ws.hip):
// argument block, LDS-DMA helper, dequantisation epilogue.
#pragma once
#include <type_traits>

#include "mq_common.h"

namespace mq {

typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void gbl_void;

__device__ __forceinline__ void dma16(const void *g, void *lds_wave_base)
{
    __builtin_amdgcn_global_load_lds((gbl_void *)g, (lds_void *)lds_wave_base, 16, 0, 0);
}

// The same with a wave-uniform 64-bit base in SGPRs and a 32-bit per-lane offset: the address needs no
// vector-ALU instruction, so a loader wave never queues for the VALU port behind the math waves' MFMAs
// (and consecutive DMAs do not serialise on one reused address register pair).
__device__ __forceinline__ void dma16_s(const char *uniform_base, unsigned lane_off, unsigned lds_wave_base)
{
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %0"
                 :: "s"(uniform_base), "v"(lane_off), "s"(lds_wave_base) : "memory");
}

// One fp32 per lane (group scales of the wave-specialised fold): lane l's word lands at lds_wave_base + 4 l.
__device__ __forceinline__ void dma4_s(const char *uniform_base, unsigned lane_off, unsigned lds_wave_base)
{
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, %0"
                 :: "s"(uniform_base), "v"(lane_off), "s"(lds_wave_base) : "memory");
}

enum { EPI_F16 = MQ_F16, EPI_BF16 = MQ_BF16, EPI_F32 = MQ_F32, EPI_I32 = 3 };

struct GemmArgs {
    const int8_t *a;
    long lda;
    int a_tiled = 0;   // activations in the tiled layout (MQ_LD_TILED): [ceil(M/16)][K_pad/64] pieces of 1 KiB
    const uint8_t *w;
    long M, N, K_pad;
    long n_tiles;  // ceil(N / 16)
    long n_pairs;  // ceil(N / 32): 16-channel tile pairs in the W4 image
    float sx0, sx1;
    const uint8_t *row_sel;
    const float *sx_vec = nullptr;   // per-row activation scales (dynamic per-token quantizer); overrides sx0/sx1
    // group-wise activation scales (--a_groupsize, quant_utils.py:181-203): sx_groups[m * n_groups + g], one group =
    // group_k consecutive k (64 or a multiple of 128).  The grouped kernel folds them into fp32 accumulators group by
    // group; the epilogue then receives FLOAT bits in the accumulator registers (acc_float) and skips the row scale.
    const float *sx_groups = nullptr;
    // asymmetric groups: x~ = s_g a + shift_g with shift_groups[m * n_groups + g] = s_g (2^(b-1) - z_g); the constant part meets
    // the group's weight sum wsum_groups[g * N + n] = sum_{k in g} q_w[n][k] (as fp32): facc += shift_g * wsum_g
    const float *shift_groups = nullptr, *wsum_groups = nullptr;
    // group-wise WEIGHT scales (--w_groupsize, reference gptq/gptq_utils.py:263-273): sw_groups[g * N + n] is the scale the GPTQ
    // solver found for channel n on k-group g.  The grouped kernel multiplies a group's (row-scaled) sum by it before the fp32
    // accumulation; the epilogue then applies the ROW scale (s_x per tensor / token-type / token) instead of s_w[n].
    const float *sw_groups = nullptr;
    // rotary position embedding folded into the store (gemm_ws.hip fast path; the q | k columns of a fused q|k|v projection):
    // output columns n < rope_cols are heads of 128 = one 128-wide tile, rotate-half partner d +- 64 in the same row of the
    // tile; rope_cos / rope_sin: [M, 128] in the OUTPUT dtype (row = token position).  Same roundings as the separate
    // launch mq_rope_inplace on the rounded Linear output: out = cast(cast(x cos) + cast(rotate_half(x) sin)).
    const void *rope_cos = nullptr, *rope_sin = nullptr;
    long rope_cols = 0;
    long n_groups = 0;
    int group_k = 0;
    int acc_float = 0;
    // Activation of the CONSUMER folded into the store (round 6; the HF modules around quant_utils.py:330-391: Qwen2MLP's
    // down_proj(act_fn(gate_proj(x)) * up_proj(x)), the vision MLP's fc2(quick_gelu(fc1(x)))), evaluated like the torch ops on
    // tensors of the OUTPUT dtype -- every op rounds once (mq_common.h act_silu_mul / act_quick_gelu):
    //   MQ_ACT_SILU_MUL:   the weight image holds gate (channels 0 .. N/2-1) and up (N/2 .. N-1) of a fused gate|up projection;
    //                      out[m][c] = silu(y[m][c]) * y[m][N/2 + c], c < N/2 -- ONE tensor of M x N/2 instead of M x N.  The loaders
    //                      pair the two halves inside a workgroup tile (a BN-wide tile = BN/2 gate + the same BN/2 up channels)
    //   MQ_ACT_QUICK_GELU: out[m][n] = quick_gelu(y[m][n])
    // Host-checked (gemm_common): tiled activations, floating-point output, no split-K / residual / rank-1 terms, N (N/2) a
    // multiple of 8 (32), aligned operands -- the kernels' act paths rely on it.
    int act = MQ_ACT_NONE;
    const void *residual = nullptr;  // [M, ldr] in the output dtype: out = cast(cast(y) + residual)
    long ldr = 0;
    const float *s_w, *bias, *x0, *w0;
    const float *x1 = nullptr, *w1 = nullptr;   // second rank-1 term y += x1[m] * w1[n] (flag combinations that need the slot twice)
    void *out;
    long ldo;
    int splits;        // split-K factor (1 = none)
    int vec_ok;        // N, ldo multiples of 8 and a 16-byte aligned output
    int par_ok;        // s_w / bias / w0 16-byte aligned
    int res_vec;       // residual rows 16-byte aligned
    int32_t *partial;  // [splits][M][N] when splits > 1
    // Launch geometry, filled on the host by set_geometry(): the kernels do no integer division (the
    // five runtime divisions of the round-1 prologue, two of them 64-bit, cost ~1.5 us per launch).
    unsigned m_blocks = 1, n_blocks = 1;
    unsigned mag_m = 0, mag_n = 0;   // ceil(2^32 / d): n / d == umulhi(n, mag) for n, d < 2^16
    int kq = 0, kr = 0;              // k units per split: split s owns kq (+1 if s < kr) units from s*kq + min(s, kr)
    // XCD grouping of the m-blocks (set_geometry): xm groups of mg = m_blocks / xm blocks each
    unsigned xm = 1, mg = 1, mag_xm = 0;
};

// The 8 XCDs have private L2s: whatever two XCDs both touch crosses the fabric twice.  Work ids are
// dealt to the XCDs in 8 contiguous ranges (tile_of_block) of the order (split, m-group, n-block,
// m-block inside the group), so the weight panels are fetched by xm XCDs each and the activation rows
// by ~8 / xm: fabric reads ~ W * xm + A * 8 / xm.  xm = the divisor of m_blocks that minimises it
// (1 for the wide gate/up GEMM, 2 for down_proj, whose activations are 40 % of its weights' size).
inline unsigned pick_m_groups(unsigned m_blocks, double w_bytes, double a_bytes)
{
    unsigned best = 1;
    double best_cost = w_bytes + 8.0 * a_bytes;
    for (unsigned xm = 2; xm <= 8 && xm <= m_blocks; ++xm) {
        if (m_blocks % xm) continue;
        const double cost = w_bytes * xm + a_bytes * 8.0 / xm;
        if (cost < 0.97 * best_cost) { best = xm; best_cost = cost; }
    }
    return best;
}

extern thread_local int g_gemm_force_xm;   // test hook (mq_gemm_debug_force, bits 8.. of `splits`); 0 = automatic
extern thread_local int g_pp_act_slab;     // test hook (bit 16 of `splits`): the slab form of the ping-pong activation epilogue

inline void set_geometry(GemmArgs &p, int BM, int BN, int k_unit, int w_bits)
{
    p.m_blocks = (unsigned)ceil_div(p.M, BM);
    p.n_blocks = (unsigned)ceil_div(p.n_tiles * 16, BN);
    const double w_bytes = (double)p.n_tiles * 16.0 * (double)p.K_pad * (w_bits == 4 ? 0.5 : 1.0);
    p.xm = pick_m_groups(p.m_blocks, w_bytes, (double)p.M * (double)p.K_pad);
    if (g_gemm_force_xm > 0 && p.m_blocks % (unsigned)g_gemm_force_xm == 0) p.xm = (unsigned)g_gemm_force_xm;
    p.mg = p.m_blocks / p.xm;
    p.mag_m = (unsigned)(((1ULL << 32) + p.mg - 1) / p.mg);
    p.mag_n = (unsigned)(((1ULL << 32) + p.n_blocks - 1) / p.n_blocks);
    p.mag_xm = (unsigned)(((1ULL << 32) + p.xm - 1) / p.xm);
    const long units = p.K_pad / k_unit;
    p.kq = (int)(units / p.splits);
    p.kr = (int)(units % p.splits);
}

// tile_of_block divides by multiplying with ceil(2^32 / d): exact while n * d < 2^32 for every (n, d) pair it
// forms (n < total work ids, d in {mg, n_blocks, xm}).  A grid beyond that would map workgroups to wrong tiles.
inline bool geometry_in_range(const GemmArgs &p)
{
    const unsigned long long total = (unsigned long long)p.m_blocks * p.n_blocks * (unsigned)p.splits;
    unsigned long long d = p.mg > p.n_blocks ? p.mg : p.n_blocks;
    if (p.xm > d) d = p.xm;
    return total * d < (1ULL << 32) && total < (1ULL << 31);
}

// Kernel arguments up front.  hipcc loads a by-value argument block lazily, field by field next to its first use, with an
// s_waitcnt in front of every dependent step: the tile map alone (grid size -> geometry -> operand pointers) cost FOUR serialized
// scalar-load round trips, ~1 us between a workgroup's entry and its first LDS-DMA (profiles/r5_ws_fixed_cost_timeline.txt: 0.55-1.3 us
// of every wave-specialised launch; the argument block of a launch is never in the scalar cache).  Naming the fields in one empty asm
// at the top of the kernel makes them live there, so their loads go out as ONE clause with ONE wait.
__device__ __forceinline__ void args_up_front(const GemmArgs &p)
{
#ifndef MQ_LAZY_ARGS
    asm volatile("" ::"s"(p.m_blocks), "s"(p.n_blocks), "s"(p.mag_m), "s"(p.mag_n), "s"(p.kq), "s"(p.kr), "s"(p.xm), "s"(p.mg),
                 "s"(p.mag_xm), "s"(p.splits), "s"(p.a), "s"(p.w), "s"(p.M), "s"(p.N), "s"(p.K_pad), "s"(p.n_pairs), "s"(p.n_tiles),
                 "s"(gridDim.x));
    asm volatile("" ::"s"(p.s_w), "s"(p.bias), "s"(p.w0), "s"(p.x0), "s"(p.sx_vec), "s"(p.w1), "s"(p.x1), "s"(p.row_sel), "s"(p.out),
                 "s"(p.ldo));
#endif
}

// workgroup -> (bm, bn, split): XCD-aware and bijective.  Block b runs on XCD b % 8; the remap gives
// every XCD one contiguous range of the order (split, m-group, n-block, m-block in group): the
// m-blocks that share a weight panel (and, under split-K, the tiles of one k-slice) are consecutive on
// ONE XCD, so a panel is fetched into one L2 once per m-group.
__device__ __forceinline__ void tile_of_id(const GemmArgs &p, unsigned total, unsigned b, int &bm, int &bn, int &split)
{
    const unsigned xcd = b & 7, idx = b >> 3;
    const unsigned q = total >> 3, r = total & 7;
    const unsigned wid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    const unsigned t1 = p.mg == 1 ? wid : __umulhi(wid, p.mag_m);
    const unsigned bm_in = wid - t1 * p.mg;
    const unsigned t2 = p.n_blocks == 1 ? t1 : __umulhi(t1, p.mag_n);
    bn = (int)(t1 - t2 * p.n_blocks);
    const unsigned sp = p.xm == 1 ? t2 : __umulhi(t2, p.mag_xm);
    split = (int)sp;
    bm = (int)((t2 - sp * p.xm) * p.mg + bm_in);
}

__device__ __forceinline__ void tile_of_block(const GemmArgs &p, int &bm, int &bn, int &split)
{
    tile_of_id(p, gridDim.x, blockIdx.x, bm, bn, split);
}

__device__ __forceinline__ void k_range_of_split(const GemmArgs &p, int split, int &k_begin, int &nk)
{
    const int extra = split < p.kr ? split : p.kr;
    k_begin = split * p.kq + extra;
    nk = p.kq + (split < p.kr ? 1 : 0);
}

// y = ((float(acc) * sx) * s_w[n]) + bias[n] + x0 * w0[n] + x1 * w1[n]; one rounding per operation.
template <int EPI>
__device__ __forceinline__ void store_quad(const GemmArgs &p, long m, long n, v4i a, float sx,
                                           float xz, float x1v = 0.0f)
{
    const bool full = (n + 4 <= p.N) && (p.ldo % 4 == 0);
    if (EPI == EPI_I32) {
        int *o = reinterpret_cast<int *>(p.out) + m * p.ldo + n;
        if (full) {
            *reinterpret_cast<v4i *>(o) = a;
        } else {
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (n + r < p.N) o[r] = a[r];
        }
        return;
    }
    float y[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const long nn = (n + r < p.N) ? n + r : p.N - 1;
        float t = (float)a[r] * sx;
        t = t * p.s_w[nn];
        if (p.bias) t = t + p.bias[nn];
        if (p.x0) {
            const float pr = xz * p.w0[nn];
            t = t + pr;
        }
        if (p.x1) {
            const float pr = x1v * p.w1[nn];
            t = t + pr;
        }
        y[r] = t;
    }
    if (p.residual) {   // torch: hidden + linear(x): the Linear's output is rounded first
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            if (n + r >= p.N) continue;
            if (EPI == EPI_F32) {
                y[r] = y[r] + reinterpret_cast<const float *>(p.residual)[m * p.ldr + n + r];
            } else {
                const unsigned short rb = reinterpret_cast<const unsigned short *>(p.residual)[m * p.ldr + n + r];
                y[r] = (EPI == EPI_F16) ? f16_bits_to_f32(f32_to_f16_bits(y[r])) + f16_bits_to_f32(rb)
                                        : bf16_bits_to_f32(f32_to_bf16_bits(y[r])) + bf16_bits_to_f32(rb);
            }
        }
    }
    if (EPI == EPI_F32) {
        float *o = reinterpret_cast<float *>(p.out) + m * p.ldo + n;
        if (full) {
            store_out(reinterpret_cast<v4f *>(o), v4f{y[0], y[1], y[2], y[3]});
        } else {
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (n + r < p.N) o[r] = y[r];
        }
    } else {
        unsigned short h[4];
#pragma unroll
        for (int r = 0; r < 4; ++r)
            h[r] = (EPI == EPI_F16) ? f32_to_f16_bits(y[r]) : f32_to_bf16_bits(y[r]);
        unsigned short *o = reinterpret_cast<unsigned short *>(p.out) + m * p.ldo + n;
        if (full) {
            store_out(reinterpret_cast<v4us *>(o), v4us{h[0], h[1], h[2], h[3]});
        } else {
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (n + r < p.N) o[r] = h[r];
        }
    }
}

// ---- epilogue -------------------------------------------------------------------------------
// Code executed once per workgroup is instruction-fetch bound (cold I-cache), so the epilogue is
// kept SMALL: each wave parks its raw int32 accumulators in a private LDS slab with a handful of
// unrolled ds_write_b128, then a ROLLED loop re-reads them row-contiguously, dequantises and
// stores 16 B (fp16) / 32 B per lane: whole 128-byte row segments, edges in the same loop.
template <int TM, int TN, int NWAVES, int RING_BYTES, int W_BITS, int EPI, bool ACT = false>
__device__ __forceinline__ void gemm_epilogue(const GemmArgs &p, v4i (&acc)[TN][TM], char *smem,
                                              int wave, int lane, int wm, int wn, long m0, long nt0,
                                              int split)
{
    // D layout: col = lane & 15 -> m, row = (lane >> 4) * 4 + r -> n
    constexpr int WN_COLS = TN * 16;                 // columns of the wave's sub-tile
    constexpr int SLAB_LD = WN_COLS * 4 + 16;        // bytes per slab row (+16: conflict-free)
    constexpr int PASS_MT =                          // m-tiles parked per pass (slab must fit)
        (TM % 4 == 0 && NWAVES * 64 * (SLAB_LD + 16) <= RING_BYTES) ? 4
        : (TM % 2 == 0 && NWAVES * 32 * (SLAB_LD + 16) <= RING_BYTES) ? 2 : 1;
    constexpr int PASS_ROWS = PASS_MT * 16;
    constexpr int SLAB_BYTES = PASS_ROWS * SLAB_LD + PASS_ROWS * 16;
    constexpr int LANES_PER_ROW = WN_COLS / 8;       // 8 outputs per lane
    constexpr int ROWS_PER_IT = 64 / LANES_PER_ROW;
    static_assert(NWAVES * SLAB_BYTES <= RING_BYTES, "epilogue slab must fit the ring");
    static_assert(TM % PASS_MT == 0 && 64 % LANES_PER_ROW == 0 && PASS_ROWS % ROWS_PER_IT == 0, "epilogue geometry");

    __syncthreads();                                 // every wave has left the operand ring
    char *slab = smem + wave * SLAB_BYTES;
    const int ml = lane & 15, nq = (lane >> 4) * 4;
    const bool to_partial = p.splits > 1;
    const int lrow = lane / LANES_PER_ROW;
    const int c8 = (lane % LANES_PER_ROW) * 8;
    const long n = nt0 * 16 + wn * WN_COLS + c8;     // first of this lane's 8 output channels
    const bool n_full = (n + 8 <= p.N) && p.vec_ok;

    // per-channel parameters of this lane's 8 outputs: two 16-byte loads each when aligned
    float swv[8], bsv[8], wzv[8], w1v[8];
    if (EPI != EPI_I32 && !to_partial) {
        if (n_full && p.par_ok) {
            const v4f one4 = {1.f, 1.f, 1.f, 1.f};      // weight-group scales were applied group by group: x 1.0 is exact
            const v4f s0 = p.sw_groups ? one4 : *reinterpret_cast<const v4f *>(p.s_w + n), s1 = p.sw_groups ? one4 : *reinterpret_cast<const v4f *>(p.s_w + n + 4);
            v4f b0 = {0.f, 0.f, 0.f, 0.f}, b1 = b0, z0 = b0, z1 = b0, u0 = b0, u1 = b0;
            if (p.bias) { b0 = *reinterpret_cast<const v4f *>(p.bias + n); b1 = *reinterpret_cast<const v4f *>(p.bias + n + 4); }
            if (p.w0) { z0 = *reinterpret_cast<const v4f *>(p.w0 + n); z1 = *reinterpret_cast<const v4f *>(p.w0 + n + 4); }
            if (p.w1) { u0 = *reinterpret_cast<const v4f *>(p.w1 + n); u1 = *reinterpret_cast<const v4f *>(p.w1 + n + 4); }
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                swv[e] = s0[e]; swv[4 + e] = s1[e];
                bsv[e] = b0[e]; bsv[4 + e] = b1[e];
                wzv[e] = z0[e]; wzv[4 + e] = z1[e];
                w1v[e] = u0[e]; w1v[4 + e] = u1[e];
            }
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const long nn = (n + e < p.N) ? n + e : p.N - 1;
                swv[e] = (n < p.N) ? (p.sw_groups ? 1.0f : p.s_w[nn]) : 0.0f;
                bsv[e] = (p.bias && n < p.N) ? p.bias[nn] : 0.0f;
                wzv[e] = (p.w0 && n < p.N) ? p.w0[nn] : 0.0f;
                w1v[e] = (p.w1 && n < p.N) ? p.w1[nn] : 0.0f;
            }
        }
    }
    // per-row parameters (activation scale set, split term) are fetched once per pass, one row
    // per lane, and parked behind the slab so the store loop never waits on global memory
    float *rowpar = reinterpret_cast<float *>(slab + PASS_ROWS * SLAB_LD);   // [PASS_ROWS][4]: s_x, x0, x1, unused

    // Activation in the store (GemmArgs::act; host-checked preconditions, see there).  Same slab, same dequantisation arithmetic as
    // the fast path below -- the Linear's output y is formed and rounded to the output dtype exactly as the plain launch stores
    // it -- then the activation as the torch ops evaluate it on that tensor.  SILU_MUL: the wave's sub-tile holds [32 gate | 32 up]
    // channel pairs (gemm_pp.hip's loader), a lane owns 8 OUTPUT channels: two 8-channel slab reads 128 bytes apart.
    // Its own instantiation (ACT; gemm_pp.hip launches it when GemmArgs::act is set): inside the plain kernels this path's loop
    // invariants, hoisted in front of the persistent kernels' k-loop, overflowed the register file of the 256 x 256 tile.
    if constexpr (ACT) {
        if constexpr (NWAVES <= 8 && EPI != EPI_I32) {
            constexpr int DT = (EPI == EPI_F16) ? MQ_F16 : (EPI == EPI_BF16 ? MQ_BF16 : MQ_F32);
            const bool silu_rt = p.act == MQ_ACT_SILU_MUL;
            const bool bias_rt = p.bias != nullptr;
            if (silu_rt && (WN_COLS % 64 != 0)) return;                     // (host: only tiles whose waves hold whole pairs)
            const long H = p.N >> 1;
            constexpr int LPR_S = WN_COLS / 16 > 0 ? WN_COLS / 16 : 1;    // lanes per slab row: SILU_MUL (half as many outputs) / GELU
            const int rpi = silu_rt ? 64 / LPR_S : ROWS_PER_IT;              // rows per iteration
            const int arow = silu_rt ? lane / LPR_S : lane / LANES_PER_ROW;
            const int ob = (silu_rt ? lane % LPR_S : lane % LANES_PER_ROW) * 8;   // first of the lane's 8 outputs inside the wave's sub-tile
            const int gc = silu_rt ? (ob >> 5) * 64 + (ob & 31) : ob;        // slab column of the (gate) operand; up: + 32
            const long no = silu_rt ? nt0 * 8 + wn * (WN_COLS / 2) + ob : nt0 * 16 + wn * WN_COLS + ob;   // output column
            const bool n_ok = silu_rt ? (no + 8 <= H) : (no + 8 <= p.N);
            // per-channel parameters of the lane's 8 outputs (and of the 8 up channels behind them): 16-byte loads -- the operands are
            // 16-byte aligned and the columns multiples of 8 (host-checked); scalar loads with their 64-bit addresses, hoisted above the
            // parking of the accumulators, overflowed the register file of the 256 x 256 tile
            const long cb = n_ok ? no : 0;
            const float *bias_or_sw = bias_rt ? p.bias : p.s_w;
            const v4f sg0 = *reinterpret_cast<const v4f *>(p.s_w + cb), sg1 = *reinterpret_cast<const v4f *>(p.s_w + cb + 4);
            const v4f su0 = *reinterpret_cast<const v4f *>(p.s_w + (silu_rt ? H : 0) + cb), su1 = *reinterpret_cast<const v4f *>(p.s_w + (silu_rt ? H : 0) + cb + 4);
            const v4f bg0 = *reinterpret_cast<const v4f *>(bias_or_sw + cb), bg1 = *reinterpret_cast<const v4f *>(bias_or_sw + cb + 4);
            const v4f bu0 = *reinterpret_cast<const v4f *>(bias_or_sw + (silu_rt ? H : 0) + cb), bu1 = *reinterpret_cast<const v4f *>(bias_or_sw + (silu_rt ? H : 0) + cb + 4);
            float sg[8], su[8], bg[8], bu[8];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                sg[e] = sg0[e]; sg[4 + e] = sg1[e];
                su[e] = su0[e]; su[4 + e] = su1[e];
                bg[e] = bg0[e]; bg[4 + e] = bg1[e];
                bu[e] = bu0[e]; bu[4 + e] = bu1[e];
            }
            // the row loop once per (activation, bias) combination: no uniform branch left between two elements
            auto rows = [&](auto silu_c, auto bias_c) {
                constexpr bool silu = decltype(silu_c)::value, has_bias = decltype(bias_c)::value;
#pragma unroll
                for (int pass = 0; pass < TM / PASS_MT; ++pass) {              // (unrolled: the accumulators are registers, no dynamic index)
#pragma unroll
                    for (int jj = 0; jj < PASS_MT; ++jj)
#pragma unroll
                        for (int i = 0; i < TN; ++i)
                            *reinterpret_cast<v4i *>(slab + (jj * 16 + ml) * SLAB_LD + (i * 16 + nq) * 4) = acc[i][pass * PASS_MT + jj];
                    if (lane < PASS_ROWS) {
                        const long mr = m0 + (wm * TM + pass * PASS_MT) * 16 + lane;
                        float sxl = p.sx0;
                        if (mr < p.M) {
                            if (p.sx_vec) sxl = p.sx_vec[mr];
                            else if (p.row_sel && p.row_sel[mr]) sxl = p.sx1;
                        }
                        rowpar[lane * 4] = sxl;
                    }
                    asm volatile("" ::: "memory");
#pragma unroll 1
                    for (int r0 = 0; r0 < PASS_ROWS; r0 += rpi) {
                        const int row = r0 + arow;
                        const long m = m0 + (wm * TM + pass * PASS_MT) * 16 + row;
                        const v4i g0 = *reinterpret_cast<const v4i *>(slab + row * SLAB_LD + gc * 4);
                        const v4i g1 = *reinterpret_cast<const v4i *>(slab + row * SLAB_LD + gc * 4 + 16);
                        v4i u0 = g0, u1 = g1;
                        if (silu) {
                            u0 = *reinterpret_cast<const v4i *>(slab + row * SLAB_LD + gc * 4 + 128);
                            u1 = *reinterpret_cast<const v4i *>(slab + row * SLAB_LD + gc * 4 + 144);
                        }
                        const float sxe = (W_BITS == 4) ? rowpar[row * 4] * 0.0625f : rowpar[row * 4];
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
                        if (m >= p.M || !n_ok) continue;
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
                }
            };
            if (silu_rt) { if (bias_rt) rows(std::true_type{}, std::true_type{}); else rows(std::true_type{}, std::false_type{}); }
            else { if (bias_rt) rows(std::false_type{}, std::true_type{}); else rows(std::false_type{}, std::false_type{}); }
        }
        return;
    }

    // Fast path (what every Linear of the prefill takes: whole 8-channel groups, fp16 / bf16 / fp32 output, no residual,
    // no split-K, integer accumulators).  Straight-line code per half pass: the slab reads and row parameters of four row
    // groups first, then the packed arithmetic, then the stores back to back.  The general loop below spent 16.5 k cycles
    // per 256 x 256 tile (a fifth of the tile, with or without the stores: profiles/r4_pp_cu_timeline.txt) on one
    // LDS round trip and a chain of uniform branches per row group.
    if (NWAVES <= 8 && EPI != EPI_I32 && !to_partial && !p.residual && !p.acc_float && __all(n_full)) {   // (16 waves: 128 registers, no room)
        constexpr int ITERS = PASS_ROWS / ROWS_PER_IT;
        constexpr int UNR = ITERS % 4 == 0 ? 4 : (ITERS % 2 == 0 ? 2 : 1);
        typedef float v2f __attribute__((ext_vector_type(2)));
        const bool has_bias = p.bias != nullptr, has_x0 = p.x0 != nullptr, has_x1 = p.x1 != nullptr;
        v2f sw2[4], bs2[4], wz2[4], w12[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            sw2[e] = v2f{swv[2 * e], swv[2 * e + 1]};
            bs2[e] = v2f{bsv[2 * e], bsv[2 * e + 1]};
            wz2[e] = v2f{wzv[2 * e], wzv[2 * e + 1]};
            w12[e] = v2f{w1v[2 * e], w1v[2 * e + 1]};
        }
#pragma unroll
        for (int pass = 0; pass < TM / PASS_MT; ++pass) {
#pragma unroll
            for (int jj = 0; jj < PASS_MT; ++jj)
#pragma unroll
                for (int i = 0; i < TN; ++i)
                    *reinterpret_cast<v4i *>(slab + (jj * 16 + ml) * SLAB_LD + (i * 16 + nq) * 4) =
                        acc[i][pass * PASS_MT + jj];
            if (lane < PASS_ROWS) {
                const long mr = m0 + (wm * TM + pass * PASS_MT) * 16 + lane;
                float sxl = p.sx0, xzl = 0.0f, x1l = 0.0f;
                if (mr < p.M) {
                    if (p.sx_vec) sxl = p.sx_vec[mr];
                    else if (p.row_sel && p.row_sel[mr]) sxl = p.sx1;
                    if (p.x0) xzl = p.x0[mr];
                    if (p.x1) x1l = p.x1[mr];
                }
                rowpar[lane * 4] = sxl;
                rowpar[lane * 4 + 1] = xzl;
                rowpar[lane * 4 + 2] = x1l;
            }
            asm volatile("" ::: "memory");          // the reads below follow the slab / parameter writes of THIS pass
            const long mp = m0 + (wm * TM + pass * PASS_MT) * 16 + lrow;
#pragma unroll
            for (int t0 = 0; t0 < ITERS; t0 += UNR) {
                v4i q0[UNR], q1[UNR];
                float sxr[UNR], xzr[UNR], x1r[UNR];
#pragma unroll
                for (int u = 0; u < UNR; ++u) {
                    const int row = (t0 + u) * ROWS_PER_IT + lrow;
                    q0[u] = *reinterpret_cast<const v4i *>(slab + row * SLAB_LD + c8 * 4);
                    q1[u] = *reinterpret_cast<const v4i *>(slab + row * SLAB_LD + c8 * 4 + 16);
                    sxr[u] = rowpar[row * 4];
                    xzr[u] = rowpar[row * 4 + 1];
                    x1r[u] = rowpar[row * 4 + 2];
                }
#pragma unroll
                for (int u = 0; u < UNR; ++u) {
                    const long m = mp + (t0 + u) * ROWS_PER_IT;
                    const int a[8] = {q0[u][0], q0[u][1], q0[u][2], q0[u][3], q1[u][0], q1[u][1], q1[u][2], q1[u][3]};
                    // int4 weights: the accumulator carries a factor 16; float(16 a) * (s_x / 16) is the same real product
                    // as float(a) * s_x (exact power-of-two rescale on both sides): same bits, no shift
                    const float sxe = (W_BITS == 4) ? sxr[u] * 0.0625f : sxr[u];
                    const v2f sx2 = v2f{sxe, sxe}, xz2 = v2f{xzr[u], xzr[u]}, x12 = v2f{x1r[u], x1r[u]};
                    float y[8];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        v2f v = v2f{(float)a[2 * e], (float)a[2 * e + 1]};
                        v = v * sx2;
                        v = v * sw2[e];
                        if (has_bias) v = v + bs2[e];
                        if (has_x0) {
                            const v2f pr = xz2 * wz2[e];
                            v = v + pr;
                        }
                        if (has_x1) {
                            const v2f pr = x12 * w12[e];
                            v = v + pr;
                        }
                        y[2 * e] = v[0];
                        y[2 * e + 1] = v[1];
                    }
                    if (m >= p.M) continue;
                    if (EPI == EPI_F32) {
                        float *o = reinterpret_cast<float *>(p.out) + m * p.ldo + n;
                        store_out(reinterpret_cast<v4f *>(o), v4f{y[0], y[1], y[2], y[3]});
                        store_out(reinterpret_cast<v4f *>(o + 4), v4f{y[4], y[5], y[6], y[7]});
                    } else {
                        v4i h;
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            h[e] = (int)((EPI == EPI_F16) ? pack2_f16(y[2 * e], y[2 * e + 1]) : pack2_bf16(y[2 * e], y[2 * e + 1]));
                        store_out(reinterpret_cast<v4i *>(reinterpret_cast<unsigned short *>(p.out) + m * p.ldo + n), h);
                    }
                }
            }
        }
        return;
    }

#pragma unroll
    for (int pass = 0; pass < TM / PASS_MT; ++pass) {
#pragma unroll
        for (int jj = 0; jj < PASS_MT; ++jj)
#pragma unroll
            for (int i = 0; i < TN; ++i)
                *reinterpret_cast<v4i *>(slab + (jj * 16 + ml) * SLAB_LD + (i * 16 + nq) * 4) =
                    acc[i][pass * PASS_MT + jj];
        if (EPI != EPI_I32 && !to_partial && lane < PASS_ROWS) {
            const long mr = m0 + (wm * TM + pass * PASS_MT) * 16 + lane;
            float sxl = p.sx0, xzl = 0.0f, x1l = 0.0f;
            if (mr < p.M) {
                if (p.sx_vec) sxl = p.sx_vec[mr];
                else if (p.row_sel && p.row_sel[mr]) sxl = p.sx1;
                if (p.x0) xzl = p.x0[mr];
                if (p.x1) x1l = p.x1[mr];
            }
            rowpar[lane * 4] = sxl;
            rowpar[lane * 4 + 1] = xzl;
            rowpar[lane * 4 + 2] = x1l;
        }
        asm volatile("" ::: "memory");              // (the compiler must not move the reads below above these writes)
        // the slab is wave-private: LDS operations of one wave complete in order
#pragma unroll 1
        for (int r0 = 0; r0 < PASS_ROWS; r0 += ROWS_PER_IT) {
            const int row = r0 + lrow;
            const long m = m0 + (wm * TM + pass * PASS_MT) * 16 + row;
            v4i q0 = *reinterpret_cast<const v4i *>(slab + row * SLAB_LD + c8 * 4);
            v4i q1 = *reinterpret_cast<const v4i *>(slab + row * SLAB_LD + c8 * 4 + 16);
            if (m >= p.M || n >= p.N) continue;
            int a[8] = {q0[0], q0[1], q0[2], q0[3], q1[0], q1[1], q1[2], q1[3]};
            if (to_partial || EPI == EPI_I32) {
                if (W_BITS == 4) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) a[e] >>= 4;
                }
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
            // int4 weights: the accumulator carries a factor 16; float(16 a) * (s_x / 16) is the same real
            // product as float(a) * s_x (exact power-of-two rescale on both sides): same bits, no shift
            // float accumulators: activation-group scales (and the 1/16 of int4 levels) are already in; with weight-group scales
            // the ROW scale is still due
            const float sx = p.acc_float ? (p.sw_groups ? rowpar[row * 4] : 1.0f) : ((W_BITS == 4) ? rowpar[row * 4] * 0.0625f : rowpar[row * 4]);
            const float xz = rowpar[row * 4 + 1], x1r = rowpar[row * 4 + 2];
            float res[8];
            if (p.residual) {   // issued ahead of the arithmetic below
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
            typedef float v2f __attribute__((ext_vector_type(2)));   // v_pk_mul_f32 / v_pk_add_f32: two outputs per instruction
            const v2f sx2 = v2f{sx, sx}, xz2 = v2f{xz, xz}, x12 = v2f{x1r, x1r};
#pragma unroll
            for (int e = 0; e < 8; e += 2) {
                v2f t = p.acc_float ? v2f{__int_as_float(a[e]), __int_as_float(a[e + 1])} : v2f{(float)a[e], (float)a[e + 1]};
                if (!p.acc_float || p.sw_groups) t = t * sx2;
                t = t * v2f{swv[e], swv[e + 1]};
                if (p.bias) t = t + v2f{bsv[e], bsv[e + 1]};
                if (p.x0) {
                    const v2f pr = xz2 * v2f{wzv[e], wzv[e + 1]};
                    t = t + pr;
                }
                if (p.x1) {
                    const v2f pr = x12 * v2f{w1v[e], w1v[e + 1]};
                    t = t + pr;
                }
                y[e] = t[0];
                y[e + 1] = t[1];
            }
            if (p.residual) {   // torch: hidden + linear(x), the Linear's output rounded first
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    float t = y[e];
                    if (EPI == EPI_F16) t = f16_bits_to_f32(f32_to_f16_bits(t));
                    if (EPI == EPI_BF16) t = bf16_bits_to_f32(f32_to_bf16_bits(t));
                    y[e] = t + res[e];
                }
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
}

}  // namespace mq
