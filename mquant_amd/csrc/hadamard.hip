// hadamard.hip -- online Hadamard rotation y = (H_K (x) H_m) [x;0] / sqrt(n), m = n/K = 2^p,
// optionally fused with the static int8 quantizer so the rotated activations never reach HBM.
//
// Reference semantics: fake_quant/utils.py:465-471 (zero pad), fake_quant/hadamard_utils.py:
// 115-128 (matmul_hadU_cuda: third-party FHT over the last m elements, then hadK @ .),
// fake_quant/quant_utils.py:334-341 (casts), and uniform.py:20-33 when quantizing.
//
// One workgroup (4 waves) owns one activation row at a time, staged in LDS:
//   A. butterflies in ascending stride, exactly the (a+b, a-b) order of the reference:
//      strides 1,2,4 inside a lane's 8 registers, strides 8..64 between lanes on the DPP network, 128 / 256 with wavefront shuffles
//      (lane ^ stride/8), strides >= 512 through LDS; then * 1/sqrt(n) (fp32 scalar) and the
//      cast to x's dtype that the FHT extension performs for half inputs.  In that mode the
//      values ARE half-precision numbers, so the LDS copy is stored as 16-bit (half the
//      footprint: three workgroups per CU instead of one);
//   B. the K x K +-1 stage on the matrix core: V_MFMA_F32_16X16X4_F32 is an exact k-ordered
//      fp32 fma chain (fma(+-1, y, acc) == acc +- y), so ascending k-steps reproduce the
//      oracle's sequential add/sub chain bit for bit.  hadK is re-packed once per workgroup
//      into word-aligned sign rows; a lane derives its +-1.0f operand with three VALU ops.
//      A unit of work is (16 output rows j) x (64 columns i); column tile g of a unit holds
//      the columns 4*c + g, so one 8-byte LDS read feeds four MFMAs and the four results of
//      a lane are adjacent in memory (one 4-byte / 8-byte store per accumulator row);
//   C. cast to x's dtype, then either store or quantize (IEEE divide, rint, clamp) to int8.
#include <math.h>

#include "hadamard_common.h"

namespace mq {

// byte offsets inside a prepared descriptor: sign words | lane masks of the fp32 sign operand | half operand images
static size_t prepared_masks_offset(int K) { return ((size_t)K * ((K + 31) / 32) * 4 + 7) / 8 * 8; }
static size_t prepared_half_offset(int K) { return (prepared_masks_offset(K) + (size_t)((K + 15) / 16) * (K / 4) * 8 + 15) / 16 * 16; }
static size_t prepared_half_bytes(int K) { return (size_t)((K + 31) / 32) * ((K + 15) / 16) * 1024; }   // per dtype

// storage of the staged row in LDS: 16-bit when the values are exactly half-precision
template <int DT, bool HALF_LDS> struct Stage;
template <int DT> struct Stage<DT, true> {
    typedef unsigned short T;
    static __device__ __forceinline__ T st(float f) { return Elem<DT>::st(f); }
    static __device__ __forceinline__ float ld(T v) { return Elem<DT>::ld(v); }
};
template <int DT> struct Stage<DT, false> {
    typedef float T;
    static __device__ __forceinline__ T st(float f) { return f; }
    static __device__ __forceinline__ float ld(T v) { return v; }
};

template <int DT, bool QUANT>
__device__ __forceinline__ int had_finish(const HadArgs &p, long row, long col, float v, float s,
                                          float *vout)
{
    v = Elem<DT>::rnd(v);
    *vout = v;
    if (!QUANT) return 0;
    if (p.skip_col0 && col == 0) {
        if (p.x0_out) p.x0_out[row] = v;
        return 0;
    }
    return quant_level(v, s, -128.0f, 127.0f);
}

template <int DT, bool QUANT>
__device__ __forceinline__ void had_emit1(const HadArgs &p, long row, long col, float v, float s)
{
    float r;
    const int q = had_finish<DT, QUANT>(p, row, col, v, s, &r);
    if (QUANT) {
        p.qout[act_offset(row, col, p.K_pad, p.ldq)] = (int8_t)q;
    } else {
        typedef typename Elem<DT>::T T;
        reinterpret_cast<T *>(p.out)[row * p.ldo + col] = Elem<DT>::st(r);
    }
}

// N = 4 / 2 adjacent columns col..col+N-1 of one row (col a multiple of N); q.s = scale of the row, q.inv = 1 / s
struct RowScale {
    float s, inv;
    bool rcp;
};

__device__ __forceinline__ RowScale row_scale(float s)
{
    return RowScale{s, 1.0f / s, quant_rcp_ok(s)};
}

// ``o``: the int8 output address (QUANT), computed by the caller: the K x K units walk their output rows j with a
// constant byte stride, had_out_stride: no 64-bit layout arithmetic per group of levels)
template <int DT, bool QUANT, int N>
__device__ __forceinline__ void had_emit_n(const HadArgs &p, long row, long col, const float (&v)[N], const RowScale &rs, bool aligned,
                                           int8_t *o)
{
    float r[N];
#pragma unroll
    for (int e = 0; e < N; ++e) r[e] = Elem<DT>::rnd(v[e]);
    if (QUANT && N == 4 && aligned) {
        unsigned w[1];
        const float r4[4] = {r[0], r[1], r[N > 2 ? 2 : 0], r[N > 3 ? 3 : 0]};   // (N == 4 here)
        quant_levels_i8_packed<4>(r4, rs.s, rs.inv, rs.rcp, w);
        if (p.skip_col0 && col == 0) {
            if (p.x0_out) p.x0_out[row] = r[0];
            w[0] &= 0xffffff00u;
        }
        *reinterpret_cast<unsigned *>(o) = w[0];
    } else if (QUANT) {
        int q[N];
        quant_levels<N>(r, rs.s, rs.inv, rs.rcp, -128.0f, 127.0f, q);
        if (p.skip_col0 && col == 0) {
            if (p.x0_out) p.x0_out[row] = r[0];
            q[0] = 0;
        }
        if (N == 2) {
            *reinterpret_cast<unsigned short *>(o) = (unsigned short)((q[0] & 0xff) | ((q[1] & 0xff) << 8));
        } else {
#pragma unroll
            for (int e = 0; e < N; ++e) o[e] = (int8_t)q[e];
        }
    } else {
        typedef typename Elem<DT>::T T;
        T *o = reinterpret_cast<T *>(p.out) + row * p.ldo + col;
#pragma unroll
        for (int e = 0; e < N; ++e) o[e] = Elem<DT>::st(r[e]);
    }
}


// K x K stage for one unit of UJ 16-row tiles x UG 16-column tiles, sign operands from the prepared
// lane masks: per 4-wide k-step UJ v_cndmask (the +-1.0f operand of a 16-row tile: ONE VALU op, the
// 64-bit lane mask arrives by scalar load), UG fp16->fp32 conversions of the staged row and UJ*UG
// V_MFMA_F32_16X16X4_F32.  The fp32 MFMA shares the vector ALU's datapath, so every VALU instruction
// beside it is lost matrix time: the round-1 form (one tile x four column tiles, three VALU ops per
// sign operand) spent 8 VALU per 4 MFMAs, a 5 x 2 unit spends 7 per 10.
template <int DT, bool QUANT, bool HALF_LDS, int UJ, int UG, bool CONTIG = false>
__device__ __forceinline__ void had_kxk_unit(const HadArgs &p, long row, const RowScale &rs, const char *ybase, int row_bytes, int swz,
                                             int jg, int cg, int lane)
{
    typedef Stage<DT, HALF_LDS> S;
    typedef typename S::T YT;
    constexpr int ESZ = (int)sizeof(YT);
    const int K = p.K, m = p.m, JT = (K + 15) / 16;
    // Zero padding (fake_quant/utils.py:465-471: down_proj 18944 -> 19968): the staged rows k >= ceil(n_in / m) are all +0 -- their
    // inputs were pad and a butterfly of zeros is zeros -- so their k-steps add +-0 to every chain: skipped when the result is
    // quantized (the chain's value is unchanged; only the SIGN of an exactly-zero sum could differ, and both signs are level 0).
    // The plain transform (QUANT = false) keeps every step: its output bits include that sign.
    const int kz = (int)((p.n_in + m - 1) >> __builtin_ctz((unsigned)m));
#ifdef MQ_HAD_NO_PAD_SKIP
    const int ksteps = K / 4;
    (void)kz;
#else
    const int ksteps = (QUANT && (kz + 3) / 4 < K / 4) ? (kz + 3) / 4 : K / 4;
#endif
    const int lc = lane & 15, lk = lane >> 4;
    jg = __builtin_amdgcn_readfirstlane(jg);        // wave-uniform: the mask loads below become scalar loads
    cg = __builtin_amdgcn_readfirstlane(cg);
    v4f acc[UJ][UG];
    float posv[UJ], negv[UJ];
#pragma unroll
    for (int jj = 0; jj < UJ; ++jj) {
        const int j = (jg * UJ + jj) * 16 + lc;
        posv[jj] = j < K ? 1.0f : 0.0f;            // rows past K contribute nothing (their outputs are never stored)
        negv[jj] = j < K ? -1.0f : 0.0f;
#pragma unroll
        for (int g = 0; g < UG; ++g) acc[jj][g] = v4f{0.f, 0.f, 0.f, 0.f};
    }
    const int col0 = cg * (16 * UG) + UG * lc;      // this lane's UG adjacent columns
    int c0 = col0 * ESZ;
    if (swz) c0 ^= (lk & 1) << 7;                   // k = 4 t + lk: the row swizzle is a per-lane constant
    const char *yp = ybase + lk * row_bytes + c0;
    const int kstride = 4 * row_bytes;
    typedef const __attribute__((address_space(4))) unsigned long long cmask_t;
    // masks[t][jt] (JT tiles per k-step, contiguous): the unit's UJ masks of a step are adjacent; tiles past
    // JT alias the last one (posv = negv = 0 there)
    int jts[UJ];
#pragma unroll
    for (int jj = 0; jj < UJ; ++jj) jts[jj] = (jg * UJ + jj < JT) ? jg * UJ + jj : JT - 1;
    cmask_t *mrow = (cmask_t *)(p.masks) + jg * UJ;
    // Two register sets, filled one k-step AHEAD (the loop is unrolled by two, so no set is ever copied): the scalar loads
    // of the masks and the LDS read of the staged values of step t+1 are in flight while the ten MFMAs of step t issue, and
    // the wave never waits for them at the top of a step.  (Per-CU timelines, profiles/r4_hadamard_cu_timeline.txt: a
    // row alone on its CU spent 1077 cycles per k-step on a SIMD whose two waves need 640 for their MFMAs.)
    typedef unsigned long long u64;
    u64 mk[2][UJ];
    unsigned hvw[2][UG == 2 ? 1 : 2];
    float bf[2][UG];
    auto fetch = [&](int set, int t) {
        const char *src = yp + t * kstride;
        if (HALF_LDS) {
            if (UG == 2) {
                hvw[set][0] = *reinterpret_cast<const unsigned *>(src);
            } else {
                const v2i hv = *reinterpret_cast<const v2i *>(src);
                hvw[set][0] = (unsigned)hv[0];
                hvw[set][UG == 2 ? 0 : 1] = (unsigned)hv[1];
            }
        } else {
#pragma unroll
            for (int g = 0; g < UG; ++g) bf[set][g] = *reinterpret_cast<const float *>(src + 4 * g);
        }
        if constexpr (CONTIG) {
            // no tile of this unit lies past JT: its UJ masks of a step are adjacent -- ONE address (one 64-bit add per k-step instead
            // of UJ) and wide scalar loads (s_load_dwordx8 / x4 + x2 instead of UJ x2)
            typedef u64 u64x4 __attribute__((ext_vector_type(4), aligned(8)));
            typedef u64 u64x2 __attribute__((ext_vector_type(2), aligned(8)));
            typedef const __attribute__((address_space(4))) u64x4 cmask4_t;
            typedef const __attribute__((address_space(4))) u64x2 cmask2_t;
            cmask_t *q = mrow + (long)t * JT;
            static_assert(UJ == 5 || UJ == 3, "unit heights");
            if constexpr (UJ == 5) {
                const u64x4 a = *reinterpret_cast<cmask4_t *>(q);
                mk[set][0] = a[0]; mk[set][1] = a[1]; mk[set][2] = a[2]; mk[set][3] = a[3];
                mk[set][4] = q[4];
            } else {
                const u64x2 a = *reinterpret_cast<cmask2_t *>(q);
                mk[set][0] = a[0]; mk[set][1] = a[1];
                mk[set][2] = q[2];
            }
            return;
        }
#pragma unroll
        for (int jj = 0; jj < UJ; ++jj) mk[set][jj] = ((cmask_t *)(p.masks))[(long)t * JT + jts[jj]];   // scalar loads
    };
    // (scalar loads return out of order, so ANY wait for one is lgkmcnt(0): the operands of a step are therefore formed --
    //  one v_cndmask per 16-row tile, the conversions -- BEFORE the next step's loads go out, and its MFMAs issue after)
    float av[UJ], b[UG];
    auto prep = [&](int set) {
        if (HALF_LDS) {
#pragma unroll
            for (int g = 0; g < UG; ++g) {
                const unsigned wd = hvw[set][UG == 2 ? 0 : g >> 1];
                b[g] = S::ld((YT)((g & 1) ? (wd >> 16) : (wd & 0xffff)));
            }
        } else {
#pragma unroll
            for (int g = 0; g < UG; ++g) b[g] = bf[set][g];
        }
        // lane l: bit l set <=> hadK[16 jt + (l & 15)][4 t + (l >> 4)] == -1; one v_cndmask with the SGPR pair
#pragma unroll
        for (int jj = 0; jj < UJ; ++jj) av[jj] = __builtin_amdgcn_inverse_ballot_w64(mk[set][jj]) ? negv[jj] : posv[jj];
    };
    auto mma = [&]() {
#pragma unroll
        for (int jj = 0; jj < UJ; ++jj)
#pragma unroll
            for (int g = 0; g < UG; ++g) acc[jj][g] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[jj], b[g], acc[jj][g], 0, 0, 0);
    };
    fetch(0, 0);
    int t = 0;
    for (; t + 2 <= ksteps; t += 2) {
        prep(0);
        __builtin_amdgcn_sched_barrier(0);
        fetch(1, t + 1);
        __builtin_amdgcn_sched_barrier(0);
        mma();
        __builtin_amdgcn_sched_barrier(0);
        prep(1);
        __builtin_amdgcn_sched_barrier(0);
        fetch(0, t + 2 < ksteps ? t + 2 : t + 1);          // (the last fetch of an even count is a repeat, never used)
        __builtin_amdgcn_sched_barrier(0);
        mma();
        __builtin_amdgcn_sched_barrier(0);
    }
    if (t < ksteps) {
        prep(0);
        mma();
    }
    const int j0 = jg * UJ * 16 + lk * 4;                       // first output row j of this lane
    const long ostride = had_out_stride(p);
    int8_t *obase = p.qout + act_offset(row, (long)j0 * m + col0, p.K_pad, p.ldq);     // unused without QUANT
#pragma unroll
    for (int jj = 0; jj < UJ; ++jj)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int j = j0 + jj * 16 + r;
            if (UG == 2 && QUANT) {
                // two output rows (r, r + 1) per group of four levels: one quantizer branch and one column-0 test per four
                if ((r & 1) || j >= K) continue;                   // K % 4 == 0 and j0 % 4 == 0: rows r, r + 1 are both valid or both past K
                const float v4[4] = {Elem<DT>::rnd(acc[jj][0][r]), Elem<DT>::rnd(acc[jj][1][r]),
                                     Elem<DT>::rnd(acc[jj][0][r + 1]), Elem<DT>::rnd(acc[jj][1][r + 1])};
                unsigned w[1];
                quant_levels_i8_packed<4>(v4, rs.s, rs.inv, rs.rcp, w);
                if (p.skip_col0 && j == 0 && col0 == 0) {          // (row j = 0, column 0) is flat column 0
                    if (p.x0_out) p.x0_out[row] = v4[0];
                    w[0] &= 0xffffff00u;
                }
                int8_t *o = obase + (jj * 16 + r) * ostride;
                *reinterpret_cast<unsigned short *>(o) = (unsigned short)w[0];
                *reinterpret_cast<unsigned short *>(o + ostride) = (unsigned short)(w[0] >> 16);
                continue;
            }
            if (j < K) {
                int8_t *o = obase + (jj * 16 + r) * ostride;
                if (UG == 4) {
                    const float v4[4] = {acc[jj][0][r], acc[jj][1][r], acc[jj][2][r], acc[jj][3][r]};
                    had_emit_n<DT, QUANT, 4>(p, row, (long)j * m + col0, v4, rs, (p.ldq & 3) == 0, o);
                } else {
                    const float v2[2] = {acc[jj][0][r], acc[jj][1][r]};
                    had_emit_n<DT, QUANT, 2>(p, row, (long)j * m + col0, v2, rs, true, o);
                }
            }
        }
}

// (16-bit staging: four waves per SIMD -- two 8-wave or four 4-wave rows per CU -- are part of the design: at most 128 registers)
template <int DT, bool QUANT, bool HALF_LDS, int THREADS, bool ACT = false, int UNIT = 0>
__global__ __launch_bounds__(THREADS, HALF_LDS ? 4 : 1) void hadamard_kernel(HadArgs p)
{
    kernarg_warm<sizeof(HadArgs), true>();         // one scalar-load round trip instead of three to six (mq_common.h)
    constexpr int HAD_THREADS = THREADS;
    constexpr int HAD_WAVES = THREADS / 64;
    typedef typename Elem<DT>::T T;
    typedef Stage<DT, HALF_LDS> S;
    typedef typename S::T YT;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char *ybase = smem;
    unsigned *hw = reinterpret_cast<unsigned *>(smem + p.y_bytes);  // [K][WPR] sign words

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const long n = p.n;
    const int K = p.K, m = p.m;
    const int mshift = __builtin_ctz((unsigned)m);      // n / K is a power of two (checked on the host)
    const int WPR = (K + 31) / 32;
    const float scale = p.inv_sqrt_n;
    const bool mid_round = (DT != MQ_F32) && !p.fp32_had;
    const int row_bytes = p.row_bytes;
    const int swz = p.swz;
    constexpr int ESZ = (int)sizeof(YT);

    // byte offset of element (k, i) of the staged row
    auto yoff = [&](int k, int i) -> int {
        int c = i * ESZ;
        if (swz) c ^= (k & 1) << 7;
        return k * row_bytes + c;
    };

    // ---- hadK sign rows (word aligned by the caller): bit b of hw[j*WPR + w] = hadK[j][32w+b] > 0
    if (K > 1) {
        const unsigned *gw = reinterpret_cast<const unsigned *>(p.had_bits);
        for (int t = tid; t < K * WPR; t += THREADS) hw[t] = gw[t];
    }

    // tiled int8 output: the 16 rows of a piece row are handled on one XCD (tiled_row_of, mq_common.h)
    // (not below one 128-row group: the map would hand ALL rows of a short batch to the workgroups of one XCD -- 16 rows took
    //  170 us on two workgroups where 16 workgroups need 26, profiles/r5_hadamard_small_m.txt)
    const bool remap = QUANT && p.ldq == MQ_LD_TILED && (gridDim.x & 7) == 0 && p.M >= 128;
    // Short batches (generation steps: one row would keep ONE workgroup busy for ~24 us on down_proj): 2^parts_log2 workgroups share a
    // row -- each stages the whole row (phase A is cheap) and runs its share of the K x K units, whose outputs it quantizes and stores.
    const int parts = (UNIT == 5 || UNIT == 3) ? (1 << p.parts_log2) : 1;
    const long v_end = (remap ? ceil_div(p.M, 128) * 128 : p.M) << ((UNIT == 5 || UNIT == 3) ? p.parts_log2 : 0);
    for (long vv = blockIdx.x; vv < v_end; vv += gridDim.x) {
        const long v = (UNIT == 5 || UNIT == 3) ? (vv >> p.parts_log2) : vv;
        const int part = (UNIT == 5 || UNIT == 3) ? (int)(vv & (parts - 1)) : 0;
        const long row = remap ? tiled_row_of(v) : v;
        if (row >= p.M) continue;                  // uniform over the workgroup
        const T *xr = reinterpret_cast<const T *>(p.x) + row * p.ldx;
        const float s = (p.row_sel && p.row_sel[row]) ? p.s1 : p.s0;
        const RowScale rs = row_scale(s);

        // ---------------- A: butterflies ------------------------------------------------
        // four-wave rows (four resident per CU): the butterflies go ahead of the other rows' matrix phase (vis.fc2 16.9 -> 16.1 us;
        // with eight-wave rows the same priority costs 4 %: profiles/r4_hadamard_cu_timeline.txt)
        if (HAD_WAVES == 4) __builtin_amdgcn_s_setprio(3);
        if (m >= 8) {
            const long nchunks = ceil_div(n, 512);
            // chunks whose global loads are in flight together; the activation prologue holds two
            // operands per chunk and must stay within 80 VGPRs (three workgroups per CU)
            constexpr int NB = ACT ? 2 : 4;
            for (long cb = wave; cb < nchunks; cb += HAD_WAVES * NB) {
              float vb[NB][8];
#pragma unroll
              for (int u = 0; u < NB; ++u)
                had_load_chunk<DT, ACT>(p, row, (cb + (long)u * HAD_WAVES) * 512 + lane * 8, vb[u]);
#pragma unroll
              for (int u = 0; u < NB; ++u) {
                const long c = cb + (long)u * HAD_WAVES;
                if (c >= nchunks) break;                       // wave-uniform
                const long idx = c * 512 + lane * 8;
                float v[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) v[i] = vb[u][i];
                // 16-bit staging rounds on the store: st(rnd(t)) == st(t), the explicit round trip would be wasted work
                // (a chunk that lies entirely in the zero padding stays +0 through every stage: no butterflies -- wave-uniform)
#ifdef MQ_HAD_NO_PAD_SKIP
                had_butterfly_chunk<DT>(v, lane, m, scale, mid_round && !HALF_LDS);
#else
                if (c * 512 < p.n_in) had_butterfly_chunk<DT>(v, lane, m, scale, mid_round && !HALF_LDS);
#endif
                if (idx < n) {   // n is a multiple of 8 here (m >= 8)
                    const int k = (int)(idx >> mshift), i0 = (int)idx & (m - 1);   // m = 2^mshift
                    char *dst = ybase + yoff(k, i0);
                    if (HALF_LDS) {
                        v8us h;
#pragma unroll
                        for (int i = 0; i < 8; ++i) h[i] = (unsigned short)S::st(v[i]);
                        *reinterpret_cast<v8us *>(dst) = h;
                    } else {
                        *reinterpret_cast<v4f *>(dst) = v4f{v[0], v[1], v[2], v[3]};
                        *reinterpret_cast<v4f *>(dst + 16) = v4f{v[4], v[5], v[6], v[7]};
                    }
                }
              }
            }
        } else {
            for (long i = tid; i < n; i += HAD_THREADS) {
                const int k = (int)(i >> mshift), i0 = (int)i & (m - 1);
                *reinterpret_cast<YT *>(ybase + yoff(k, i0)) =
                    S::st((i < p.n_in) ? Elem<DT>::ld(xr[i]) : 0.0f);
            }
        }
        if (HAD_WAVES == 4) __builtin_amdgcn_s_setprio(0);
        __syncthreads();
        // (co-factors above 512 always stage in fp32: the register block below exists only in those instantiations and
        //  cannot raise the register count -- and lower the occupancy -- of the half-precision-staging kernels)
        if (!HALF_LDS && m > 512 && m <= 8192) {
            // Strides 512 .. m/2: the co-factor row is R = m / 512 chunks whose 512-point transforms are done; what is
            // left is an R-point butterfly over the elements at the same offset of the R chunks, in the same ascending
            // stride order.  One thread takes one (k, offset) column through all of it in registers: ONE pass over
            // LDS (then scale and cast) instead of log2(R) read-modify-write passes plus a scaling pass.
            const int R = m >> 9;
            for (long t = tid; t < (n >> 9) / R * 512; t += HAD_THREADS) {
                const int k = (int)(t >> 9), o = (int)t & 511;
                float w[16];
#pragma unroll
                for (int c = 0; c < 16; ++c)
                    if (c < R) w[c] = S::ld(*reinterpret_cast<const YT *>(ybase + yoff(k, c * 512 + o)));
#pragma unroll
                for (int h = 1; h < 16; h <<= 1) {
                    if (h >= R) break;
#pragma unroll
                    for (int c = 0; c < 16; ++c) {
                        if ((c & h) == 0 && c + h < R) {
                            const float a0 = w[c], a1 = w[c + h];
                            w[c] = a0 + a1;
                            w[c + h] = a0 - a1;
                        }
                    }
                }
#pragma unroll
                for (int c = 0; c < 16; ++c) {
                    if (c < R) {
                        float v = w[c] * scale;
                        if (mid_round) v = Elem<DT>::rnd(v);
                        *reinterpret_cast<YT *>(ybase + yoff(k, c * 512 + o)) = S::st(v);
                    }
                }
            }
            __syncthreads();
        } else if (!HALF_LDS && (m > 512 || m < 8)) {   // remaining strides through LDS (fp32 staging: HALF_LDS is off)
            for (long h = (m < 8) ? 1 : 512; h < m; h <<= 1) {
                const int hs = __builtin_ctzl((unsigned long)h);               // h is a power of two
                for (long b = tid; b < n / 2; b += HAD_THREADS) {
                    const long i = ((b >> hs) << (hs + 1)) + (b & (h - 1));
                    const int k = (int)(i >> mshift), i0 = (int)i & (m - 1);
                    YT *pa = reinterpret_cast<YT *>(ybase + yoff(k, i0));
                    YT *pb = reinterpret_cast<YT *>(ybase + yoff(k, i0 + (int)h));
                    const float a0 = S::ld(*pa), a1 = S::ld(*pb);
                    *pa = S::st(a0 + a1);
                    *pb = S::st(a0 - a1);
                }
                __syncthreads();
            }
            for (long i = tid; i < n; i += HAD_THREADS) {
                const int k = (int)(i >> mshift), i0 = (int)i & (m - 1);
                YT *py = reinterpret_cast<YT *>(ybase + yoff(k, i0));
                float t = S::ld(*py) * scale;
                if (mid_round) t = Elem<DT>::rnd(t);
                *py = S::st(t);
            }
            __syncthreads();
        }

        // ---------------- B/C: K x K stage, cast, store / quantize -----------------------
        if (UNIT == 0 && K == 1) {
            if ((n & 3) == 0 && (!QUANT || ((p.ldq & 3) == 0 && (((uintptr_t)p.qout) & 3) == 0))) {
                for (long idx = (long)tid * 4; idx < n; idx += HAD_THREADS * 4) {
                    float v4[4];
#pragma unroll
                    for (int i = 0; i < 4; ++i) v4[i] = S::ld(*reinterpret_cast<const YT *>(ybase + yoff(0, (int)(idx + i))));
                    had_emit_n<DT, QUANT, 4>(p, row, idx, v4, rs, true, p.qout + act_offset(row, idx, p.K_pad, p.ldq));
                }
            } else {
                for (long idx = (long)tid * 8; idx < n; idx += HAD_THREADS * 8) {
#pragma unroll
                    for (int i = 0; i < 8; ++i)
                        if (idx + i < n)
                            had_emit1<DT, QUANT>(p, row, idx + i,
                                                 S::ld(*reinterpret_cast<const YT *>(ybase + yoff(0, (int)(idx + i)))), s);
                }
            }
        } else if (UNIT == 3) {
            // prepared descriptor: 3 x 2 units, ONE per wave (the vision tower's fc2, 40 x 128: four; Qwen-VL's 172 x 64: eight).
            // The classic path below has six (16 rows x 64 columns) units for the four waves of a row; the waves with two
            // sit on the same two SIMDs in every resident workgroup, which then carry twice the matrix work of the others.
            const int JT = (K + 15) / 16;
            const int JG = (JT + 2) / 3, CG = m / 32;
            for (int u = wave * parts + part; u < JG * CG; u += HAD_WAVES * parts)
#ifndef MQ_HAD_MASKS_SEPARATE
                if ((u / CG + 1) * 3 <= (p.K + 15) / 16) had_kxk_unit<DT, QUANT, HALF_LDS, 3, 2, true>(p, row, rs, ybase, row_bytes, swz, u / CG, u % CG, lane);
                else
#endif
                had_kxk_unit<DT, QUANT, HALF_LDS, 3, 2>(p, row, rs, ybase, row_bytes, swz, u / CG, u % CG, lane);
        } else if (UNIT == 5) {
            // prepared descriptor: mask-driven 5 x 2 units (had_kxk_unit), one unit per wave for K = 156, m = 128.
            // Its own instantiation: the 40 accumulator registers must not raise the register count (and lower
            // the occupancy) of the shapes that take the classic path below.
            const int JT = (K + 15) / 16;
            const int JG = (JT + 4) / 5, CG = m / 32;
            for (int u = wave * parts + part; u < JG * CG; u += HAD_WAVES * parts)
#ifndef MQ_HAD_MASKS_SEPARATE
                if ((u / CG + 1) * 5 <= (p.K + 15) / 16) had_kxk_unit<DT, QUANT, HALF_LDS, 5, 2, true>(p, row, rs, ybase, row_bytes, swz, u / CG, u % CG, lane);
                else
#endif
                had_kxk_unit<DT, QUANT, HALF_LDS, 5, 2>(p, row, rs, ybase, row_bytes, swz, u / CG, u % CG, lane);
        } else if (m >= 64) {
            const int JT = (K + 15) / 16;
            const int CG = m / 64;
            const int ksteps = K / 4;
            const int lc = lane & 15, lk = lane >> 4;
            for (int u = wave; u < JT * CG; u += HAD_WAVES) {
                const int jt = u / CG, cg = u - jt * CG;
                v4f acc[4];
#pragma unroll
                for (int g = 0; g < 4; ++g) acc[g] = v4f{0.f, 0.f, 0.f, 0.f};
                const int ja = jt * 16 + lc;
                const bool jvalid = ja < K;
                const unsigned *hrow = hw + (jvalid ? ja : 0) * WPR;
                const int col0 = cg * 64 + 4 * lc;           // this lane's 4 adjacent columns
                // k = 4*ks + lk, so (k & 1) == (lk & 1): the row swizzle is a per-lane constant
                const char *yp = ybase + yoff(lk, col0);
                const int kstride = 4 * row_bytes;           // bytes per k-step
                const unsigned sh0 = 31u - (unsigned)lk;
                auto kstep = [&](unsigned word, int q, const char *src) {
                    // +-1.0f from bit (4q + lk): move it to the sign position; 0xBF800000 is -1.0f
                    unsigned t = word << (sh0 - 4u * (unsigned)q);
                    t = (t & 0x80000000u) ^ 0xBF800000u;
                    const float a = jvalid ? __uint_as_float(t) : 0.0f;
                    float b[4];
                    if (HALF_LDS) {
                        const v4us hv = *reinterpret_cast<const v4us *>(src);
#pragma unroll
                        for (int g = 0; g < 4; ++g) b[g] = S::ld((YT)hv[g]);
                    } else {
                        const v4f fv = *reinterpret_cast<const v4f *>(src);
#pragma unroll
                        for (int g = 0; g < 4; ++g) b[g] = fv[g];
                    }
#pragma unroll
                    for (int g = 0; g < 4; ++g)
                        acc[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b[g], acc[g], 0, 0, 0);
                };
                const int full_words = ksteps >> 3, tail = ksteps & 7;
                for (int w = 0; w < full_words; ++w) {
                    const unsigned word = hrow[w];
#pragma unroll
                    for (int q = 0; q < 8; ++q) kstep(word, q, yp + (w * 8 + q) * kstride);
                }
                if (tail) {
                    const unsigned word = hrow[full_words];
#pragma unroll
                    for (int q = 0; q < 7; ++q)
                        if (q < tail) kstep(word, q, yp + (full_words * 8 + q) * kstride);
                }
                const int j0 = jt * 16 + lk * 4;
                int8_t *obase = p.qout + act_offset(row, (long)j0 * m + col0, p.K_pad, p.ldq);     // unused without QUANT
                const long ostride = had_out_stride(p);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int j = j0 + r;
                    if (j < K) {
                        const float v4[4] = {acc[0][r], acc[1][r], acc[2][r], acc[3][r]};
                        had_emit_n<DT, QUANT, 4>(p, row, (long)j * m + col0, v4, rs, (p.ldq & 3) == 0, obase + r * ostride);
                    }
                }
            }
        } else {
            // narrow co-factor (m < 64): scalar chain per output element
            for (long o = tid; o < n; o += HAD_THREADS) {
                const int j = (int)(o >> mshift), i = (int)o & (m - 1);
                float acc = 0.0f;
                for (int k = 0; k < K; ++k) {
                    const float v = S::ld(*reinterpret_cast<const YT *>(ybase + yoff(k, i)));
                    acc = ((hw[j * WPR + (k >> 5)] >> (k & 31)) & 1u) ? (acc + v) : (acc - v);
                }
                had_emit1<DT, QUANT>(p, row, o, acc, s);
            }
        }
        if (QUANT) {
            for (long c = n + tid; c < p.K_pad; c += HAD_THREADS) p.qout[act_offset(row, c, p.K_pad, p.ldq)] = 0;
        }
        __syncthreads();  // the staged row is reused by the next row
    }
}

template <int DT, bool QUANT, bool HALF_LDS, int THREADS, bool ACT = false, int UNIT = 0>
static int launch_hadamard_t(HadArgs p, hipStream_t st)
{
    const int esz = HALF_LDS ? 2 : 4;
    p.row_bytes = p.m * esz;
    p.swz = (p.K > 1 && p.row_bytes % 256 == 0) ? 1 : 0;
    long yb = p.n * esz;
    p.y_bytes = (int)((yb + 15) / 16 * 16);
    const size_t wpr = (p.K + 31) / 32;
    const size_t lds = (size_t)p.y_bytes + (p.K > 1 ? (size_t)p.K * wpr * 4 : 0);
    if (lds > 160 * 1024) return fail(MQ_EUNSUPPORTED, "mq_hadamard: n=%ld needs %zu B of LDS (> 160 KiB)", p.n, lds);
    auto kern = hadamard_kernel<DT, QUANT, HALF_LDS, THREADS, ACT, UNIT>;
    {   // per device and instantiation; the first call of a shape happens outside any stream capture
        const int rc = ensure_dynamic_lds((const void *)kern, 160 * 1024);
        if (rc != MQ_OK) return rc;
    }
    long per_cu = (160 * 1024) / (long)lds;
    if (per_cu > 8) per_cu = 8;
#ifndef MQ_HAD_GRID_BY_LDS
    // Only as many workgroups as are RESIDENT (16-bit staging: 16 waves per CU, the kernel's launch bounds): the row loop
    // hands a workgroup its next row the moment it is done, whereas a workgroup dispatched into a freed slot started
    // 15 k cycles later (profiles/r4_hadamard_cu_timeline.txt)
    if (HALF_LDS && per_cu > 16 / (THREADS / 64)) per_cu = 16 / (THREADS / 64);
#endif
    if (per_cu < 1) per_cu = 1;
    long blocks = (long)device_cu_count() * per_cu;
    p.parts_log2 = 0;
    if ((UNIT == 5 || UNIT == 3) && QUANT && p.M < 128) {
        // short batch: share a row's units among workgroups while a CU is still free (units per row: a power of two here)
        const int JT = (p.K + 15) / 16, units = ((JT + UNIT - 1) / UNIT) * (p.m / 32);
        while ((2 << p.parts_log2) <= units && units % (2 << p.parts_log2) == 0 && (p.M << (p.parts_log2 + 1)) <= device_cu_count()) ++p.parts_log2;
    }
    // (with the row map the VIRTUAL rows count: 130 rows are 256 virtual ones, and 136 workgroups would walk two of them each)
    const long vrows = (QUANT && p.ldq == MQ_LD_TILED && p.M >= 128) ? ceil_div(p.M, 128) * 128 : (p.M << p.parts_log2);
    if (blocks > vrows) blocks = vrows;
    if (QUANT && p.ldq == MQ_LD_TILED) blocks = ceil_div(blocks, 8) * 8;   // XCD-consistent row map (tiled_row_of)
    hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(THREADS), lds, st, p);
    return check_launch("hadamard");
}

// Prepared descriptor: [K][ceil(K/32)] sign words (as passed to mq_hadamard*) followed, 8-byte aligned, by
// the lane masks of the MFMA sign operand: masks[t][jt] for t < K/4, jt < ceil(K/16), bit l set when
// hadK[16 jt + (l & 15)][4 t + (l >> 4)] == -1 (rows >= K: 0).
__global__ __launch_bounds__(64) void hadamard_prepare_kernel(const unsigned *words, int K, unsigned *out_words,
                                                                unsigned long long *out_masks, v4i *out_half)
{
    const int WPR = (K + 31) / 32, ksteps = K / 4;
    const int lane = threadIdx.x;
    for (int i = blockIdx.x * 64 + lane; i < K * WPR; i += gridDim.x * 64) out_words[i] = words[i];
    const int JT = (K + 15) / 16;
    for (int u = blockIdx.x; u < JT * ksteps; u += gridDim.x) {
        const int t = u / JT, jt = u - t * JT;                      // masks[t][jt]
        const int j = jt * 16 + (lane & 15), k = 4 * t + (lane >> 4);
        const bool neg = j < K && ((words[j * WPR + (k >> 5)] >> (k & 31)) & 1u) == 0;   // bit set = +1
        const unsigned long long mask = __ballot(neg);
        if (lane == 0) out_masks[u] = mask;
    }
    // operand images of the fast mode (hadamard_fast.hip): [fp16 | bf16][ceil(K/32) tiles][ceil(K/16) k-steps][64 lanes],
    // lane l = the 8 signs H[32 jt + (l & 31)][16 ks + 8 (l >> 5) + e] as +-1.0 half numbers (0 outside the matrix)
    const int JT32 = (K + 31) / 32, KS16 = (K + 15) / 16;
    for (int u = blockIdx.x; u < 2 * JT32 * KS16; u += gridDim.x) {
        const int dt = u / (JT32 * KS16), r = u - dt * JT32 * KS16, jt = r / KS16, ks = r - jt * KS16;
        const unsigned one = dt == 0 ? 0x3C00u : 0x3F80u;
        const int j = jt * 32 + (lane & 31), k0 = ks * 16 + 8 * (lane >> 5);
        v4i f;
#pragma unroll
        for (int e2 = 0; e2 < 4; ++e2) {
            unsigned h[2];
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                const int k = k0 + 2 * e2 + b;
                h[b] = 0;
                if (j < K && k < K) h[b] = ((words[j * WPR + (k >> 5)] >> (k & 31)) & 1u) ? one : (one | 0x8000u);
            }
            f[e2] = (int)(h[0] | (h[1] << 16));
        }
        out_half[(long)u * 64 + lane] = f;
    }
}

static thread_local int g_had_threads = 0;   // test hook (mq_hadamard_debug_threads); 0: choose by shape

template <int DT, bool QUANT, bool HALF_LDS>
static int launch_hadamard(const HadArgs &p, hipStream_t st)
{
    // 8 waves per row pay off once the K x K stage has enough (16 rows x 64 columns) units to
    // keep them busy (down_proj: 10 x 2 = 20 units; measured 83 -> 76 us), else 4 waves
    int units = (p.K > 1 && p.m >= 64) ? ((p.K + 15) / 16) * (p.m / 64) : 0;
    int threads = g_had_threads ? g_had_threads : (units >= 16 ? 512 : 256);
    if (p.masks && p.K > 1 && p.m >= 64) {          // mask-driven units: one unit per wave where possible
        const int JT = (p.K + 15) / 16;
        units = p.unit_j == 5 ? ((JT + 4) / 5) * (p.m / 32) : p.unit_j == 3 ? ((JT + 2) / 3) * (p.m / 32) : JT * (p.m / 64);
        if (!g_had_threads) threads = units >= 8 ? 512 : 256;
    }
    const bool unit5 = p.masks && p.unit_j == 5 && threads == 512;
    const bool unit3 = p.masks && p.unit_j == 3 && threads == 256 && units == 4;
    const bool unit3w = p.masks && p.unit_j == 3 && threads == 512 && units == 8;
    if (QUANT && p.act != MQ_ACT_NONE) {
        if (unit5) return launch_hadamard_t<DT, QUANT, HALF_LDS, 512, QUANT, 5>(p, st);
        if (unit3) return launch_hadamard_t<DT, QUANT, HALF_LDS, 256, QUANT, 3>(p, st);
        if (unit3w) return launch_hadamard_t<DT, QUANT, HALF_LDS, 512, QUANT, 3>(p, st);
        if (threads == 512) return launch_hadamard_t<DT, QUANT, HALF_LDS, 512, QUANT>(p, st);
        return launch_hadamard_t<DT, QUANT, HALF_LDS, 256, QUANT>(p, st);
    }
    if (unit5) return launch_hadamard_t<DT, QUANT, HALF_LDS, 512, false, 5>(p, st);
    if (unit3) return launch_hadamard_t<DT, QUANT, HALF_LDS, 256, false, 3>(p, st);
    if (unit3w) return launch_hadamard_t<DT, QUANT, HALF_LDS, 512, false, 3>(p, st);
    if (threads == 512) return launch_hadamard_t<DT, QUANT, HALF_LDS, 512>(p, st);
    return launch_hadamard_t<DT, QUANT, HALF_LDS, 256>(p, st);
}

template <int DT, bool QUANT>
static int launch_hadamard_dt(const HadArgs &p, hipStream_t st)
{
    // 16-bit staging is exact only when the scaled butterflies are rounded to x's dtype AND no
    // later butterfly pass runs through LDS (m <= 512, m >= 8)
    const bool half_ok = (DT != MQ_F32) && !p.fp32_had && p.m >= 8 && p.m <= 512;
    if (DT != MQ_F32 && half_ok) return launch_hadamard<DT, QUANT, (DT != MQ_F32)>(p, st);
    return launch_hadamard<DT, QUANT, false>(p, st);
}

static int hadamard_common(HadArgs p, int x_dtype, bool quant, void *stream)
{
    MQ_REQUIRE(p.M >= 0 && p.n > 0 && p.n_in > 0 && p.n_in <= p.n, "mq_hadamard: bad shape (n_in=%ld, n=%ld)", p.n_in, p.n);
    if (p.M == 0) return MQ_OK;
    MQ_REQUIRE(p.K >= 1 && p.n % p.K == 0, "mq_hadamard: K=%d does not divide n=%ld", p.K, p.n);
    p.m = (int)(p.n / p.K);
    p.inv_sqrt_n = 1.0f / sqrtf((float)p.n);
    MQ_REQUIRE((p.m & (p.m - 1)) == 0, "mq_hadamard: n/K=%d is not a power of two", p.m);
    MQ_REQUIRE(p.K == 1 || (p.K % 4 == 0 && p.had_bits && ((uintptr_t)p.had_bits) % 4 == 0), "mq_hadamard: K=%d needs 4-byte aligned had_words and K %% 4 == 0", p.K);
    // fp32_had carries flags: bit 0 = --fp32_had, bit 1 = had_words is a prepared descriptor (mq_hadamard_prepare),
    // bit 2 = THIS call may take the non-default fast K x K stage (MQ_HAD_FAST)
    const bool prepared = (p.fp32_had & MQ_HAD_PREPARED) != 0;
    const bool fast = (p.fp32_had & MQ_HAD_FAST) != 0;
    p.fp32_had &= MQ_HAD_FP32;
    p.masks = nullptr;
    p.hfrag = nullptr;
    p.unit_j = 1; p.unit_g = 4;
    if (prepared && p.K > 1) {
        MQ_REQUIRE(((uintptr_t)p.had_bits) % 16 == 0, "mq_hadamard: a prepared descriptor is 16-byte aligned");
        p.masks = reinterpret_cast<const unsigned long long *>(p.had_bits + prepared_masks_offset(p.K));
        if (x_dtype != MQ_F32)
            p.hfrag = reinterpret_cast<const v4i *>(p.had_bits + prepared_half_offset(p.K) + (x_dtype == MQ_BF16 ? prepared_half_bytes(p.K) : 0));
        const int JT = (p.K + 15) / 16;
        // measured (profiles/r2_hadamard.txt): the 5 x 2 units pay off for the large factors (K = 140 / 156 /
        // 172, down_proj 76 -> 72 us); for small K the round-1 form (one tile x four column tiles) is faster
        // measured (profiles/r2_hadamard.txt, r4_hadamard_cu_timeline.txt): mask-driven units pay off where a row splits into
        // exactly ONE unit per wave (wave w of every resident workgroup runs on the same SIMD, so uneven shares pile up there):
        // 5 x 2 units for K = 140 / 156 with m = 128 (eight waves), 3 x 2 units for K = 172 with m = 64 (eight waves) and for the
        // vision tower's 40 x 128 (four waves); everything else keeps the round-1 form (one tile x four column tiles)
        const int u5 = ((JT + 4) / 5) * (p.m / 32), u3 = ((JT + 2) / 3) * (p.m / 32);
        if (p.m >= 64 && p.m % 32 == 0 && JT >= 8 && u5 == 8) {
            p.unit_j = 5;
            p.unit_g = 2;
        } else if (p.m >= 64 && p.m % 32 == 0 && (u3 == 8 || u3 == 4)) {
            p.unit_j = 3;
            p.unit_g = 2;
        } else {
            p.masks = nullptr;
        }
    }
    MQ_REQUIRE(p.ldx >= p.n_in, "mq_hadamard: ldx < n_in");
    const size_t esz = (x_dtype == MQ_F32) ? 4 : 2;
    p.vec_ok = (((uintptr_t)p.x) % 16 == 0) && ((p.ldx * esz) % 16 == 0);
    p.vec_ok2 = p.x2 && (((uintptr_t)p.x2) % 16 == 0) && ((p.ldx * esz) % 16 == 0);
    hipStream_t st = (hipStream_t)stream;
    if (quant) {
        MQ_REQUIRE(p.qout && p.K_pad >= p.n && (p.ldq == MQ_LD_TILED ? p.K_pad % 64 == 0 : p.ldq >= p.K_pad),
                   "mq_hadamard_quant_i8: bad output geometry");
        MQ_REQUIRE(((uintptr_t)p.qout) % 4 == 0, "mq_hadamard_quant_i8: out must be 4-byte aligned");
        p.vec_ok_q = (((uintptr_t)p.qout) % 16 == 0) && (p.ldq == MQ_LD_TILED || p.ldq % 16 == 0);
    } else {
        MQ_REQUIRE(p.out && p.ldo >= p.n, "mq_hadamard: bad output geometry");
    }
    if (fast) {              // non-default per-call flag MQ_HAD_FAST: K x K stage on the half-precision matrix core
        const int rc = hadamard_fast_dispatch(p, x_dtype, quant, st);
        if (rc != MQ_EUNSUPPORTED) return rc;
    }
    if (quant) {
        switch (x_dtype) {
        case MQ_F16: return launch_hadamard_dt<MQ_F16, true>(p, st);
        case MQ_BF16: return launch_hadamard_dt<MQ_BF16, true>(p, st);
        case MQ_F32: return launch_hadamard_dt<MQ_F32, true>(p, st);
        }
    } else {
        MQ_REQUIRE(p.out && p.ldo >= p.n, "mq_hadamard: bad output geometry");
        switch (x_dtype) {
        case MQ_F16: return launch_hadamard_dt<MQ_F16, false>(p, st);
        case MQ_BF16: return launch_hadamard_dt<MQ_BF16, false>(p, st);
        case MQ_F32: return launch_hadamard_dt<MQ_F32, false>(p, st);
        }
    }
    return fail(MQ_EINVAL, "mq_hadamard: unknown dtype %d", x_dtype);
}

}  // namespace mq

extern "C" size_t mq_hadamard_prepared_bytes(int K)
{
    if (K <= 1 || K % 4 != 0) return 0;
    return mq::prepared_half_offset(K) + 2 * mq::prepared_half_bytes(K);
}

extern "C" int mq_hadamard_prepare(const uint32_t *had_words, int K, void *descriptor, void *stream)
{
    using namespace mq;
    MQ_REQUIRE(K > 1 && K % 4 == 0 && had_words && descriptor, "mq_hadamard_prepare: K=%d must be a multiple of 4 (> 1), buffers non-null", K);
    MQ_REQUIRE(((uintptr_t)descriptor) % 16 == 0 && ((uintptr_t)had_words) % 4 == 0, "mq_hadamard_prepare: alignment (descriptor: 16 bytes)");
    char *d = reinterpret_cast<char *>(descriptor);
    hipLaunchKernelGGL(hadamard_prepare_kernel, dim3(64), dim3(64), 0, (hipStream_t)stream, had_words, K,
                       reinterpret_cast<unsigned *>(d), reinterpret_cast<unsigned long long *>(d + prepared_masks_offset(K)),
                       reinterpret_cast<v4i *>(d + prepared_half_offset(K)));
    return check_launch("hadamard_prepare");
}

extern "C" int mq_hadamard_debug_threads(int threads)
{
    // TEST-ONLY: 256 / 512 threads per row for the calling thread's later launches; 0: by shape
    mq::g_had_threads = (threads == 512 || threads == 256) ? threads : 0;
    return MQ_OK;
}

extern "C" int mq_hadamard(const void *x, int x_dtype, long M, long n_in, long ldx, long n, int K,
                           const uint32_t *had_words, int fp32_had, void *out, long ldo, void *stream)
{
    mq::HadArgs p;
    memset(&p, 0, sizeof(p));
    p.x = x; p.M = M; p.n_in = n_in; p.ldx = ldx; p.n = n; p.K = K; p.had_bits = reinterpret_cast<const uint8_t *>(had_words);
    p.fp32_had = fp32_had; p.out = out; p.ldo = ldo; p.s0 = p.s1 = 1.0f;
    return mq::hadamard_common(p, x_dtype, false, stream);
}

extern "C" int mq_hadamard_quant_i8(const void *x, int x_dtype, long M, long n_in, long ldx, long n,
                                    int K, const uint32_t *had_words, int fp32_had, float scale0,
                                    float scale1, const uint8_t *row_sel, int skip_col0,
                                    float *x0_out, int8_t *out, long K_pad, long ldo, void *stream)
{
    mq::HadArgs p;
    memset(&p, 0, sizeof(p));
    p.x = x; p.M = M; p.n_in = n_in; p.ldx = ldx; p.n = n; p.K = K; p.had_bits = reinterpret_cast<const uint8_t *>(had_words);
    p.fp32_had = fp32_had; p.s0 = scale0; p.s1 = scale1; p.row_sel = row_sel;
    p.skip_col0 = skip_col0; p.x0_out = x0_out; p.qout = out; p.K_pad = K_pad; p.ldq = ldo;
    return mq::hadamard_common(p, x_dtype, true, stream);
}

extern "C" int mq_act_hadamard_quant_i8(const void *x, const void *x2, int act, int x_dtype, long M,
                                        long n_in, long ldx, long n, int K, const uint32_t *had_words,
                                        int fp32_had, float scale0, float scale1, const uint8_t *row_sel,
                                        int skip_col0, float *x0_out, int8_t *out, long K_pad, long ldo,
                                        void *stream)
{
    using namespace mq;
    if (M == 0) return MQ_OK;                       // empty input: nothing to do (null pointers allowed)
    MQ_REQUIRE(act == MQ_ACT_SILU_MUL || act == MQ_ACT_QUICK_GELU, "mq_act_hadamard_quant_i8: unknown activation %d", act);
    MQ_REQUIRE(act != MQ_ACT_SILU_MUL || x2 != nullptr, "mq_act_hadamard_quant_i8: silu(gate)*up needs the second operand");
    MQ_REQUIRE(K >= 1 && n % K == 0 && n / K >= 8, "mq_act_hadamard_quant_i8: n/K must be >= 8");
    HadArgs p;
    memset(&p, 0, sizeof(p));
    p.x = x; p.x2 = x2; p.act = act;
    p.M = M; p.n_in = n_in; p.ldx = ldx; p.n = n; p.K = K; p.had_bits = reinterpret_cast<const uint8_t *>(had_words);
    p.fp32_had = fp32_had; p.s0 = scale0; p.s1 = scale1; p.row_sel = row_sel;
    p.skip_col0 = skip_col0; p.x0_out = x0_out; p.qout = out; p.K_pad = K_pad; p.ldq = ldo;
    return hadamard_common(p, x_dtype, true, stream);
}
