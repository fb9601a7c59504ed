// hadamard_valu.hip -- EXACT online Hadamard rotation (+ static int8 quantizer) for half-precision activations with
// co-factor m = n / K = 128: the K x K +-1 stage on the packed-fp32 vector ALU with SCALAR sign operands.
//
// Reference: fake_quant/hadamard_utils.py:115-128 (matmul_hadU_cuda: FHT over the last m elements, then hadK @ .),
// fake_quant/utils.py:465-471 (zero pad), quant_utils.py:334-341 (casts), quantizer/uniform.py:20-33 (quantizer).
// Bit-identical to hadamard.hip (and so to hadamard_fwd.npz / the oracle): the K x K stage is the same sequential
// ascending-k fp32 chain, z_j <- fma(+-1, y_k, z_j) = z_j +- y_k with its single rounding.
//
// Why not the matrix core (round 4, profiles/r4_probe_valu_pk_rate.txt): V_MFMA_F32_16X16X4_F32 runs at the plain
// vector rate (64 flop / clk / SIMD) and blocks the SIMD while it runs, and every MFMA needs a generated +-1 operand
// (v_cndmask), a staged-row fragment and its conversions.  v_pk_fma_f32 with the sign as an SGPR operand retires
// 128 lane-FMAs in ~3 cycles (85 flop / clk / SIMD with >= 4 waves per SIMD) and needs NO per-FMA operand work:
// a lane owns two adjacent columns, the accumulators of JB output rows j are register pairs, and per k the wave
// reads one staged pair (ds_read_b32 + two converts) and issues JB instructions
//      v_pk_fma_f32 acc[j], y_k, s[j][k]      s[j][k] = +-1.0f in an SGPR (low or high word of a pair, op_sel)
// whose signs arrive by s_load_dwordx16 / x8 / x2 from a table in the prepared descriptor (mq_hadamard_prepare).
//
// One workgroup = one activation row (NJB = ceil(K / JB) waves, wave w owns output rows JB w ...), rows looped:
//   A. load (16 k-rows x 128 columns per wave pass, 4 x 8 elements per lane; zero pad beyond n_in; optional fused
//      activation, had_load_chunk), butterflies in the reference's ascending-stride (a + b, a - b) order: strides
//      1, 2, 4 and 32, 64 inside the lane's registers, strides 8, 16 between lanes (ds_swizzle + one fma(own, +-1,
//      other) per element), * 1/sqrt(n), stored as the 16-bit values the FHT extension returns (exact staging);
//   B. the K x K stage as above;
//   C. cast to x's dtype, quantize (quant_levels: the exact division-free quantizer of mq_common.h), 2 levels per
//      16-bit store into the tiled (or row-major) int8 matrix; or store the rotated values.
#include "hadamard_common.h"

namespace mq {

typedef float v2f __attribute__((ext_vector_type(2)));
typedef unsigned long long u64;
typedef u64 v8u64 __attribute__((ext_vector_type(8)));
typedef u64 v4u64 __attribute__((ext_vector_type(4)));
typedef u64 v2u64 __attribute__((ext_vector_type(2)));
typedef const __attribute__((address_space(4))) v8u64 c_v8u64;
typedef const __attribute__((address_space(4))) v4u64 c_v4u64;
typedef const __attribute__((address_space(4))) v2u64 c_v2u64;
typedef const __attribute__((address_space(4))) u64 c_u64;

constexpr int VALU_ROW_DWORDS = 32;      // sign table: [block][k][32 dwords] (JB <= 32 signs used)

// acc += y * s, s in the LOW / HIGH word of an SGPR pair, broadcast to both halves of the packed operation
__device__ __forceinline__ void pk_fma_lo(v2f &acc, v2f y, u64 s)
{
    asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(acc) : "v"(y), "s"(s));
}
__device__ __forceinline__ void pk_fma_hi(v2f &acc, v2f y, u64 s)
{
    asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,1,0] op_sel_hi:[1,1,1]" : "+v"(acc) : "v"(y), "s"(s));
}

// JB signs of one k (JB / 2 SGPR pairs), loaded with the widest scalar loads that cover them
template <int JB> struct SignRow;
template <> struct SignRow<26> {          // 13 pairs = 8 + 4 + 1
    v8u64 a; v4u64 b; u64 c;
    __device__ __forceinline__ void load(const char *t)
    {
        a = *(c_v8u64 *)(t);
        b = *(c_v4u64 *)(t + 64);
        c = *(c_u64 *)(t + 96);
    }
    __device__ __forceinline__ u64 pair(int i) const { return i < 8 ? a[i] : (i < 12 ? b[i - 8] : c); }
};
template <> struct SignRow<20> {          // 10 pairs = 8 + 2
    v8u64 a; v2u64 b;
    __device__ __forceinline__ void load(const char *t)
    {
        a = *(c_v8u64 *)(t);
        b = *(c_v2u64 *)(t + 64);
    }
    __device__ __forceinline__ u64 pair(int i) const { return i < 8 ? a[i] : b[i - 8]; }
};

// NW = ceil(K / JB) waves per workgroup; OCC = workgroups per CU the register budget is asked to allow
template <int DT, bool QUANT, bool ACT, int JB, int NW, int OCC>
__global__ __launch_bounds__(NW * 64, (OCC * NW + 3) / 4) void hadamard_valu_kernel(HadArgs p, const char *sign_table)
{
    typedef typename Elem<DT>::T T;
    constexpr int M128 = 128, ROW_BYTES = M128 * 2;
    extern __shared__ __attribute__((aligned(16))) char ybase[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    constexpr int waves = NW;                          // = ceil(K / JB)
    const int K = p.K;
    const long n = p.n;
    const float scale = p.inv_sqrt_n;

    // staged row: y[k][i] as 16-bit values, 64-byte groups of odd k-rows swapped (phase A's 16-byte stores of two
    // k-rows per 8-lane group then fall into different banks; phase B reads whole 256-byte rows either way)
    auto yoff = [&](int k, int i) -> int { return k * ROW_BYTES + ((i * 2) ^ ((k & 1) << 6)); };

    // Warm the scalar cache with this wave's sign rows (K x 128 bytes, shared by every workgroup of the CU): behind a GEMM
    // the table is cold, and phase B would pay one miss per k in sequence (its loads run only one k ahead).  Sixteen
    // independent one-dword loads per batch, one per 64-byte line; the values only have to arrive.
    {
        typedef const __attribute__((address_space(4))) unsigned c_u32;
        const char *tabw = sign_table + (long)wave * K * (VALU_ROW_DWORDS * 4);
        unsigned sink = 0;
        for (int l0 = 0; l0 < 2 * K; l0 += 16) {
            unsigned t[16];
#pragma unroll
            for (int i = 0; i < 16; ++i) t[i] = *(c_u32 *)(tabw + (long)((l0 + i < 2 * K) ? l0 + i : 0) * 64);
#pragma unroll
            for (int i = 0; i < 16; ++i) sink ^= t[i];
        }
        asm volatile("" ::"s"(sink));
    }
    const bool remap = QUANT && p.ldq == MQ_LD_TILED && (gridDim.x & 7) == 0;
    const long v_end = remap ? ceil_div(p.M, 128) * 128 : p.M;
    // phase A geometry: lane -> k-row (lane >> 2) of a 16-row group, columns 8 (lane & 3) + 32 q + 0..7
    const int ka = lane >> 2, ca = 8 * (lane & 3);
    const float sg8 = (lane & 1) ? -1.0f : 1.0f, sg16 = (lane & 2) ? -1.0f : 1.0f;

    for (long v = blockIdx.x; v < v_end; v += gridDim.x) {
        const long row = remap ? tiled_row_of(v) : v;
        if (row >= p.M) continue;                      // uniform over the workgroup
        const float s = (p.row_sel && p.row_sel[row]) ? p.s1 : p.s0;
        const float inv_s = 1.0f / s;
        const bool rcp = quant_rcp_ok(s);

#ifdef MQ_HV_STAMP
        const unsigned long long st0 = __builtin_readcyclecounter();
#endif
        // ---------------- A: load, butterflies over the 128 columns of each k-row, stage -------------------------
        for (int g = wave; g * 16 < K; g += waves) {
            const int k = g * 16 + ka;
            float e[4][8];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                if (k < K) {
                    had_load_chunk<DT, ACT>(p, row, (long)k * M128 + ca + 32 * q, e[q]);
                } else {
#pragma unroll
                    for (int i = 0; i < 8; ++i) e[q][i] = 0.0f;
                }
            }
            // strides 1, 2, 4: inside each group of 8
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int h = 1; h < 8; h <<= 1)
#pragma unroll
                    for (int i = 0; i < 8; ++i)
                        if ((i & h) == 0) {
                            const float a0 = e[q][i], a1 = e[q][i + h];
                            e[q][i] = a0 + a1;
                            e[q][i + h] = a0 - a1;
                        }
            // strides 8, 16: lanes ^1, ^2 (lower lane of a pair: a + b, upper: a - b = fma(own, -1, other))
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const float o = __int_as_float(__builtin_amdgcn_ds_swizzle(__float_as_int(e[q][i]), 0x041f));   // lane ^ 1
                    e[q][i] = __builtin_fmaf(e[q][i], sg8, o);
                }
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const float o = __int_as_float(__builtin_amdgcn_ds_swizzle(__float_as_int(e[q][i]), 0x081f));   // lane ^ 2
                    e[q][i] = __builtin_fmaf(e[q][i], sg16, o);
                }
            // strides 32, 64: between the four groups of the lane
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const float a0 = e[0][i], a1 = e[1][i], a2 = e[2][i], a3 = e[3][i];
                const float b0 = a0 + a1, b1 = a0 - a1, b2 = a2 + a3, b3 = a2 - a3;
                e[0][i] = b0 + b2;
                e[1][i] = b1 + b3;
                e[2][i] = b0 - b2;
                e[3][i] = b1 - b3;
            }
            if (k < K) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    v8us h;
#pragma unroll
                    for (int i = 0; i < 8; ++i) h[i] = (unsigned short)Elem<DT>::st(e[q][i] * scale);   // the FHT extension's cast
                    *reinterpret_cast<v8us *>(ybase + yoff(k, ca + 32 * q)) = h;
                }
            }
        }
        __syncthreads();
#ifdef MQ_HV_STAMP
        const unsigned long long st1 = __builtin_readcyclecounter();
        unsigned long long st2 = 0;
#endif

        // ---------------- B: z[j][c] = sum_k H[j][k] y[k][c], sequential in k, JB rows j per wave ------------------
        {
            const int j0 = wave * JB;
            const char *tab = sign_table + (long)wave * K * (VALU_ROW_DWORDS * 4);
            v2f acc[JB];
#pragma unroll
            for (int j = 0; j < JB; ++j) acc[j] = v2f{0.0f, 0.0f};
            SignRow<JB> sr[2];
            sr[0].load(tab);
            unsigned ycur = *reinterpret_cast<const unsigned *>(ybase + yoff(0, 2 * lane));
            for (int k = 0; k < K; k += 2) {                            // K % 4 == 0
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const int kk = k + u;
                    const int kn = kk + 1 < K ? kk + 1 : kk;
#ifndef MQ_HV_NOSIGN
                    sr[u ^ 1].load(tab + (long)kn * (VALU_ROW_DWORDS * 4));
#else
                    sr[u ^ 1] = sr[u];
#endif
                    const unsigned ynext = *reinterpret_cast<const unsigned *>(ybase + yoff(kn, 2 * lane));
                    const v2f y = v2f{Elem<DT>::ld((T)(ycur & 0xffff)), Elem<DT>::ld((T)(ycur >> 16))};
#pragma unroll
                    for (int j = 0; j < JB; j += 2) {
                        const u64 sp = sr[u].pair(j >> 1);
                        pk_fma_lo(acc[j], y, sp);
                        pk_fma_hi(acc[j + 1], y, sp);
                    }
                    ycur = ynext;
                }
            }

#ifdef MQ_HV_STAMP
            st2 = __builtin_readcyclecounter();
#endif
            // ---------------- C: cast, quantize / store --------------------------------------------------------
            const int c0 = 2 * lane;
            if (QUANT) {
                const long ostride = had_out_stride(p);
                int8_t *obase = p.qout + act_offset(row, (long)j0 * M128 + c0, p.K_pad, p.ldq);
#pragma unroll
                for (int j = 0; j < JB; j += 2) {
                    if (j0 + j >= K) continue;                           // K and JB are even: rows j, j + 1 exist together (wave-uniform)
                    const float v4[4] = {Elem<DT>::rnd(acc[j][0]), Elem<DT>::rnd(acc[j][1]), Elem<DT>::rnd(acc[j + 1][0]), Elem<DT>::rnd(acc[j + 1][1])};
                    int q[4];
                    quant_levels<4>(v4, s, inv_s, rcp, -128.0f, 127.0f, q);
                    if (p.skip_col0 && j0 + j == 0 && c0 == 0) {        // (row j = 0, column 0) is flat column 0
                        if (p.x0_out) p.x0_out[row] = v4[0];
                        q[0] = 0;
                    }
                    int8_t *o = obase + j * ostride;
                    *reinterpret_cast<unsigned short *>(o) = (unsigned short)((q[0] & 0xff) | ((q[1] & 0xff) << 8));
                    *reinterpret_cast<unsigned short *>(o + ostride) = (unsigned short)((q[2] & 0xff) | ((q[3] & 0xff) << 8));
                }
            } else {
                T *o = reinterpret_cast<T *>(p.out) + row * p.ldo + (long)j0 * M128 + c0;
#pragma unroll
                for (int j = 0; j < JB; ++j) {
                    if (j0 + j >= K) continue;                           // wave-uniform
                    const unsigned lo = (unsigned)Elem<DT>::st(acc[j][0]), hi = (unsigned)Elem<DT>::st(acc[j][1]);
                    *reinterpret_cast<unsigned *>(o + (long)j * M128) = lo | (hi << 16);
                }
            }
        }
#ifdef MQ_HV_STAMP
        if (p.x0_out && !p.skip_col0 && lane == 0) {
            const unsigned long long st3 = __builtin_readcyclecounter();
            float *o = p.x0_out + (row * NW + wave) * 4;
            o[0] = (float)(st1 - st0); o[1] = (float)(st2 - st1); o[2] = (float)(st3 - st2); o[3] = (float)(st0 & 0xffffff);
        }
#endif
        if (QUANT) {
            for (long c = n + tid; c < p.K_pad; c += NW * 64) p.qout[act_offset(row, c, p.K_pad, p.ldq)] = 0;
        }
        __syncthreads();  // the staged row is reused by the next row
    }
}

// rows j per wave for a factor K (0: this kernel does not serve K)
int hadamard_valu_jb(int K)
{
#ifdef MQ_HV_OFF            // A/B builds: every shape on the matrix-core kernel
    return 0;
#endif
    // K = 156 (Qwen2-VL-7B down_proj) was built and measured too (JB = 26, six waves per row): its K x K phase runs at the
    // full v_pk_fma_f32 rate (61 k cycles per wave at 4.5 waves per SIMD), but under that load the chip clocks at ~1.0-1.3 GHz
    // (power) where the matrix-core kernel keeps 2.39 GHz: 85-103 us against 65-71 us (profiles/r4_hadamard_valu.txt).  This
    // kernel serves the small factors, whose time is mostly phases A and C.
    switch (K) {
    case 52: return 26;
    case 40: case 60: case 20: return 20;
    default: return 0;
    }
}

size_t hadamard_valu_table_bytes(int K)
{
    const int jb = hadamard_valu_jb(K);
    if (!jb) return 0;
    return (size_t)((K + jb - 1) / jb) * K * VALU_ROW_DWORDS * 4;
}

// table[b][k][d] = +-1.0f = hadK[b * JB + d][k] (0.0f past K / past JB): bit set in the sign words = +1
__global__ __launch_bounds__(64) void hadamard_valu_table_kernel(const unsigned *words, int K, int JB, float *table)
{
    const int WPR = (K + 31) / 32, NJB = (K + JB - 1) / JB;
    for (long i = (long)blockIdx.x * 64 + threadIdx.x; i < (long)NJB * K * VALU_ROW_DWORDS; i += (long)gridDim.x * 64) {
        const int d = (int)(i % VALU_ROW_DWORDS), k = (int)((i / VALU_ROW_DWORDS) % K), b = (int)(i / ((long)VALU_ROW_DWORDS * K));
        const int j = b * JB + d;
        float v = 0.0f;
        if (d < JB && j < K) v = ((words[j * WPR + (k >> 5)] >> (k & 31)) & 1u) ? 1.0f : -1.0f;
        table[i] = v;
    }
}

void hadamard_valu_fill_table(const unsigned *words, int K, void *table, hipStream_t st)
{
    const int jb = hadamard_valu_jb(K);
    if (!jb) return;
    hipLaunchKernelGGL(hadamard_valu_table_kernel, dim3(64), dim3(64), 0, st, words, K, jb, reinterpret_cast<float *>(table));
}

template <int DT, bool QUANT, bool ACT, int JB, int NW, int OCC>
static int launch_valu_t(HadArgs p, const char *table, hipStream_t st)
{
    constexpr int waves = NW;
    const size_t lds = (size_t)p.K * 256;
    if ((p.K + JB - 1) / JB != NW || lds > 160 * 1024) return MQ_EUNSUPPORTED;
    auto kern = hadamard_valu_kernel<DT, QUANT, ACT, JB, NW, OCC>;
    const int rc = ensure_dynamic_lds((const void *)kern, 160 * 1024);
    if (rc != MQ_OK) return rc;
    long per_cu = (160 * 1024) / (long)lds;
    if (per_cu > OCC) per_cu = OCC;
    if (per_cu < 1) per_cu = 1;
    long blocks = 256L * per_cu;
    if (blocks > p.M) blocks = p.M;
    if (QUANT && p.ldq == MQ_LD_TILED) blocks = ceil_div(blocks, 8) * 8;   // XCD-consistent row map (tiled_row_of)
    hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(waves * 64), lds, st, p, table);
    return check_launch("hadamard_valu");
}

template <int DT, bool QUANT, bool ACT>
static int launch_valu_jb(const HadArgs &p, const char *table, hipStream_t st)
{
    switch (p.K) {
    case 52: return launch_valu_t<DT, QUANT, ACT, 26, 2, 8>(p, table, st);
    case 40: return launch_valu_t<DT, QUANT, ACT, 20, 2, 8>(p, table, st);
    case 60: return launch_valu_t<DT, QUANT, ACT, 20, 3, 5>(p, table, st);
    case 20: return launch_valu_t<DT, QUANT, ACT, 20, 1, 16>(p, table, st);
    default: return MQ_EUNSUPPORTED;
    }
}

// Returns MQ_EUNSUPPORTED (without an error message) when the shape / dtype is outside this kernel: the caller then
// runs the matrix-core kernel of hadamard.hip (same results).
int hadamard_valu_dispatch(const HadArgs &p, const char *table, int x_dtype, bool quant, hipStream_t st)
{
    if (!table || p.K <= 1 || p.fp32_had || x_dtype == MQ_F32 || p.m != 128 || (p.K % 4) != 0 || !hadamard_valu_jb(p.K)) return MQ_EUNSUPPORTED;
    const bool act = p.act != MQ_ACT_NONE;
    if (act && !quant) return MQ_EUNSUPPORTED;
    if (x_dtype == MQ_F16) {
        if (quant) return act ? launch_valu_jb<MQ_F16, true, true>(p, table, st) : launch_valu_jb<MQ_F16, true, false>(p, table, st);
        return launch_valu_jb<MQ_F16, false, false>(p, table, st);
    }
    if (quant) return act ? launch_valu_jb<MQ_BF16, true, true>(p, table, st) : launch_valu_jb<MQ_BF16, true, false>(p, table, st);
    return launch_valu_jb<MQ_BF16, false, false>(p, table, st);
}

}  // namespace mq
