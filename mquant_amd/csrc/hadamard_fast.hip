// hadamard_fast.hip -- NON-DEFAULT mode of the online Hadamard rotation (per-call flag MQ_HAD_FAST): the K x K
// +-1 stage on the half-precision matrix core instead of the exact fp32 one.
//
// Reference: fake_quant/hadamard_utils.py:115-128 -- on a GPU the reference itself evaluates this stage as a
// half-precision GEMM (``hadK.to(input.dtype) @ input``, fp32 accumulation in library order).  The staged
// values are exactly fp16 / bf16 numbers (the FHT extension's output dtype), +-1 * y is exact, so this mode
// differs from the exact kernel (hadamard.hip: a sequential fp32 add chain, the order of the reference's CPU
// run and of the goldens) ONLY in the order in which the 156 (40, 28, ...) exact products are accumulated in
// fp32.  It is not bit-identical to hadamard_fwd.npz and therefore never the default; DESIGN.md 4.2 gives
// the measured int8 level flip rate against the exact mode.
//
// Structure (one 4-wave workgroup per activation row, rows looped persistently):
//   A. the exact kernel's phase A (had_load_chunk / had_butterfly_chunk), staged row-major y[k][i] in LDS as
//      16-bit values, rows k >= K (up to the next multiple of 16) kept zero;
//   B. out^T[i][j] = sum_k y^T[i][k] H^T[k][j] with V_MFMA_F32_32X32X16_{F16,BF16}: the A operand (32 columns
//      i x 16 k) comes out of the row-major image through ds_read_b64_tr_b16, the hardware transpose read
//      (a 16-lane group reads a [4 k][16 i] block, lane t receives y[k0..k0+3][i0 + t];
//      tools/probes/ds_read_tr.hip); the B operand (+-1 as half numbers) lives in REGISTERS for the j tiles a
//      wave owns, generated once per kernel from the sign words -- no per-MFMA operand arithmetic at all;
//   C. D layout: lane = output row j, registers = 4 x 4 adjacent columns i: cast, quantize (IEEE divide, rint,
//      clamp like the exact kernel), pack 4 levels per dword, two v_permlane32_swap turn the 4 dwords of a lane
//      pair into two 16-byte runs, ONE 16-byte store per lane and 32 x 32 tile (a whole chunk of the tiled
//      activation layout).
#include "hadamard_common.h"

namespace mq {

typedef short v4s_t __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) v4s_t lds_v4s_t;
typedef _Float16 v8h_t __attribute__((ext_vector_type(8)));
typedef __bf16 v8bf_t __attribute__((ext_vector_type(8)));
typedef float v16f_t __attribute__((ext_vector_type(16)));

template <int DT> struct HalfMma;
template <> struct HalfMma<MQ_F16> {
    static constexpr unsigned ONE = 0x3C00u, MINUS_ONE = 0xBC00u;
    static __device__ __forceinline__ v16f_t mma(v4i a, v4i b, v16f_t c)
    {
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(v8h_t, a), __builtin_bit_cast(v8h_t, b), c, 0, 0, 0);
    }
};
template <> struct HalfMma<MQ_BF16> {
    static constexpr unsigned ONE = 0x3F80u, MINUS_ONE = 0xBF80u;
    static __device__ __forceinline__ v16f_t mma(v4i a, v4i b, v16f_t c)
    {
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(v8bf_t, a), __builtin_bit_cast(v8bf_t, b), c, 0, 0, 0);
    }
};

// KS = ceil(K / 16) k-steps, JT = ceil(K / 32) output-row tiles.  A wave keeps the sign operand of NH tiles:
// all of them when JT <= 2, else its "own" tile (wave index) plus, for JT = 5 / 6, one shared extra tile.
template <int DT, bool QUANT, bool ACT, int KS, int JT>
__global__ __launch_bounds__(256, 3) void hadamard_fast_kernel(HadArgs p)
{
    kernarg_warm<sizeof(HadArgs), true>();
    typedef typename Elem<DT>::T T;
    typedef HalfMma<DT> MM;
    constexpr int WAVES = 4;
    constexpr int NH = JT <= 2 ? JT : (JT > 4 ? 2 : 1);
    constexpr int EXTRA = JT > 4 ? JT - 4 : 1;      // extra tiles beyond the four "own" ones
    static_assert(JT == 1 || JT == 2 || (JT >= 4 && JT <= 6), "tile assignment is written for these counts");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char *ybase = smem;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const long n = p.n;
    const int K = p.K, m = p.m;
    const int mshift = __builtin_ctz((unsigned)m);      // n / K is a power of two (checked on the host)
    const int IT = m >> 5;                          // 32-column tiles of the co-factor
    const int row_bytes = p.row_bytes;              // m * 2
    const float scale = p.inv_sqrt_n;

    // byte offset of element (k, i) of the staged row: 32-byte groups of a row are XOR-ed by (k & 3), so the
    // four k-rows a transpose read touches sit in four different bank groups (row stride = 2^q * 128 B)
    auto yoff = [&](int k, int i) -> int { return k * row_bytes + ((i * 2) ^ ((k & 3) << 5)); };

    // ---- sign operand of this wave's tiles, once per kernel: lane l holds H[j = 32 jt + (l & 31)][k = 16 ks + 8 (l >> 5) + e]
    int own_jt[NH];
    if (JT <= 2) {
#pragma unroll
        for (int h = 0; h < NH; ++h) own_jt[h] = h;
    } else {
        own_jt[0] = wave;
        if (NH == 2) own_jt[1] = 4 + (wave % EXTRA);
    }
    v4i Hf[NH][KS];
    if (p.hfrag) {           // prepared descriptor: one 16-byte load per operand (a workgroup owns ONE row: nothing amortises a rebuild)
#pragma unroll
        for (int h = 0; h < NH; ++h)
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) Hf[h][ks] = p.hfrag[((long)own_jt[h] * KS + ks) * 64 + lane];
    } else {
        const unsigned *gw = reinterpret_cast<const unsigned *>(p.had_bits);
        const int WPR = (K + 31) / 32;
#pragma unroll
        for (int h = 0; h < NH; ++h) {
            const int j = own_jt[h] * 32 + (lane & 31);
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const int k0 = ks * 16 + 8 * (lane >> 5);          // 8 consecutive k: inside one 32-bit word
                unsigned bits = 0, valid = 0;
                if (j < K && k0 < K) {
                    bits = gw[j * WPR + (k0 >> 5)] >> (k0 & 31);   // bit e set = +1
                    valid = (K - k0 >= 8) ? 0xffu : ((1u << (K - k0)) - 1u);
                }
                v4i f;
#pragma unroll
                for (int e2 = 0; e2 < 4; ++e2) {
                    unsigned lo = 0, hi = 0;
                    if ((valid >> (2 * e2)) & 1u) lo = ((bits >> (2 * e2)) & 1u) ? MM::ONE : MM::MINUS_ONE;
                    if ((valid >> (2 * e2 + 1)) & 1u) hi = ((bits >> (2 * e2 + 1)) & 1u) ? MM::ONE : MM::MINUS_ONE;
                    f[e2] = (int)(lo | (hi << 16));
                }
                Hf[h][ks] = f;
            }
        }
    }
    // rows K .. 16 KS - 1 of the staged image are multiplied by sign 0: they must hold finite values
    for (int b = tid * 16; b < (16 * KS - K) * row_bytes; b += 256 * 16)
        *reinterpret_cast<v4i *>(ybase + K * row_bytes + b) = v4i{0, 0, 0, 0};

    const bool remap = QUANT && p.ldq == MQ_LD_TILED && (gridDim.x & 7) == 0 && p.M >= 128;   // (short batches: hadamard.hip)
    const long v_end = remap ? ceil_div(p.M, 128) * 128 : p.M;
    for (long v = blockIdx.x; v < v_end; v += gridDim.x) {
        const long row = remap ? tiled_row_of(v) : v;
        if (row >= p.M) continue;                  // uniform over the workgroup
        const float s = (p.row_sel && p.row_sel[row]) ? p.s1 : p.s0;
        const float inv_s = 1.0f / s;
        const bool rcp = quant_rcp_ok(s);

        // ---------------- A: butterflies, staged as 16-bit values --------------------------------------
        {
            const long nchunks = ceil_div(n, 512);
            auto finish_chunk = [&](float (&vv)[8], long c) {
                const long idx = c * 512 + lane * 8;
                had_butterfly_chunk<DT>(vv, lane, m, scale, false);   // the 16-bit store below is the FHT extension's cast
                if (idx < n) {
                    const int k = (int)(idx >> mshift), i0 = (int)idx & (m - 1);   // m = 2^mshift
                    v8us h;
#pragma unroll
                    for (int i = 0; i < 8; ++i) h[i] = (unsigned short)Elem<DT>::st(vv[i]);
                    *reinterpret_cast<v8us *>(ybase + yoff(k, i0)) = h;
                }
            };
            // chunks whose global loads are in flight together.  (Requesting ALL chunks of the wave up front, as raw
            // 16-bit vectors, was measured slower -- 40.6 vs 37.2 us on down_proj: the kernel is bound by
            // vector-ALU issue, not by load latency; profiles/r3_hadamard_fast_mode.txt)
            constexpr int NB = ACT ? 2 : 4;
            for (long cb = wave; cb < nchunks; cb += WAVES * NB) {
                float vb[NB][8];
#pragma unroll
                for (int u = 0; u < NB; ++u)
                    had_load_chunk<DT, ACT>(p, row, (cb + (long)u * WAVES) * 512 + lane * 8, vb[u]);
#pragma unroll
                for (int u = 0; u < NB; ++u) {
                    const long c = cb + (long)u * WAVES;
                    if (c >= nchunks) break;                        // wave-uniform
                    finish_chunk(vb[u], c);
                }
            }
        }
        __syncthreads();

        // ---------------- B / C: K x K stage on the half-precision matrix core, quantize, store ---------
        // Units (j tile, 32-column tile it) of this wave.  JT <= 2: every wave holds every tile and takes the units
        // u = jt * IT + it with u % 4 == wave.  JT >= 4: all column tiles of the wave's own tile, and the column
        // tiles it % SH == rank of the extra tile it shares with SH - 1 other waves.
        const int t16 = lane & 15, g16 = (lane >> 4) & 1, ko = lane >> 5;
#pragma unroll
        for (int h = 0; h < NH; ++h) {
            int it_begin, it_step;
            if (JT <= 2) {
                it_begin = ((wave - h * IT) % WAVES + WAVES) % WAVES;
                it_step = WAVES;
            } else if (h == 0) {
                it_begin = 0;
                it_step = 1;
            } else {
                it_begin = wave / EXTRA;
                it_step = WAVES / EXTRA;
            }
            const int jt = own_jt[h];
            for (int it = it_begin; it < IT; it += it_step) {
                v16f_t acc;
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[e] = 0.0f;
                const int i_lane = it * 32 + 16 * g16 + 4 * (t16 & 3);      // this lane's 8-byte piece of the [4 k][16 i] block
                const int k_lane = 8 * ko + (t16 >> 2);
                constexpr int CH = KS <= 6 ? KS : (KS + 1) / 2;             // k-steps whose fragments are in flight together
#pragma unroll
                for (int c0 = 0; c0 < KS; c0 += CH) {
                    v4i a[CH];
#pragma unroll
                    for (int c = 0; c < CH; ++c) {
                        if (c0 + c >= KS) break;
                        const int k = (c0 + c) * 16 + k_lane;
                        const v4s_t r0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4s_t *)(ybase + yoff(k, i_lane)));
                        const v4s_t r1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4s_t *)(ybase + yoff(k + 4, i_lane)));
                        const v2i lo = __builtin_bit_cast(v2i, r0), hi = __builtin_bit_cast(v2i, r1);
                        a[c] = v4i{lo[0], lo[1], hi[0], hi[1]};
                    }
#pragma unroll
                    for (int c = 0; c < CH; ++c) {
                        if (c0 + c >= KS) break;
                        acc = MM::mma(a[c], Hf[h][c0 + c], acc);
                    }
                }
                // D[i][j]: lane -> j = 32 jt + (lane & 31); registers 4 q + r -> i = 32 it + 8 q + 4 (lane >> 5) + r
                const int j = jt * 32 + (lane & 31);
                const bool jvalid = j < K;
                if (QUANT) {
                    unsigned d[4];
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        float val[4];
                        int lv[4];
#pragma unroll
                        for (int r = 0; r < 4; ++r) val[r] = Elem<DT>::rnd(acc[4 * q + r]);
                        quant_levels<4>(val, s, inv_s, rcp, -128.0f, 127.0f, lv);
                        if (p.skip_col0 && q == 0 && j == 0 && it == 0 && ko == 0) {
                            if (p.x0_out) p.x0_out[row] = val[0];
                            lv[0] = 0;
                        }
                        d[q] = (lv[0] & 0xff) | ((lv[1] & 0xff) << 8) | ((lv[2] & 0xff) << 16) | ((unsigned)(lv[3] & 0xff) << 24);
                    }
                    // lower half-wave keeps columns 0..15 of the tile, upper half-wave 16..31
                    const auto s0 = __builtin_amdgcn_permlane32_swap(d[0], d[2], false, false);
                    const auto s1 = __builtin_amdgcn_permlane32_swap(d[1], d[3], false, false);
                    if (jvalid) {
                        const long col = (long)j * m + it * 32 + 16 * ko;
                        int8_t *o = p.qout + act_offset(row, col, p.K_pad, p.ldq);
                        const v4i out4 = v4i{(int)s0[0], (int)s0[1], (int)s1[0], (int)s1[1]};
                        if (p.vec_ok_q) {
                            *reinterpret_cast<v4i *>(o) = out4;
                        } else {
#pragma unroll
                            for (int w4 = 0; w4 < 4; ++w4) *reinterpret_cast<int *>(o + 4 * w4) = out4[w4];
                        }
                    }
                } else if (jvalid) {
                    T *o = reinterpret_cast<T *>(p.out) + row * p.ldo + (long)j * m + it * 32 + 4 * ko;
#pragma unroll
                    for (int q = 0; q < 4; ++q)
#pragma unroll
                        for (int r = 0; r < 4; ++r) o[8 * q + r] = Elem<DT>::st(acc[4 * q + r]);
                }
            }
        }
        if (QUANT) {
            for (long c = n + tid; c < p.K_pad; c += 256) p.qout[act_offset(row, c, p.K_pad, p.ldq)] = 0;
        }
        __syncthreads();  // the staged row is reused by the next row
    }
}

template <int DT, bool QUANT, bool ACT, int KS, int JT>
static int launch_fast_t(HadArgs p, hipStream_t st)
{
    p.row_bytes = p.m * 2;
    const size_t lds = (size_t)16 * KS * p.row_bytes;
    if (lds > 160 * 1024) return MQ_EUNSUPPORTED;
    auto kern = hadamard_fast_kernel<DT, QUANT, ACT, KS, JT>;
    const int rc = ensure_dynamic_lds((const void *)kern, 160 * 1024);
    if (rc != MQ_OK) return rc;
    long per_cu = (160 * 1024) / (long)lds;
    if (per_cu > 4) per_cu = 4;
    if (per_cu < 1) per_cu = 1;
    long blocks = 256L * per_cu;
    const long vrows = (QUANT && p.ldq == MQ_LD_TILED && p.M >= 128) ? ceil_div(p.M, 128) * 128 : p.M;   // (virtual rows of the row map)
    if (blocks > vrows) blocks = vrows;
    if (QUANT && p.ldq == MQ_LD_TILED) blocks = ceil_div(blocks, 8) * 8;   // XCD-consistent row map (tiled_row_of)
    hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(256), lds, st, p);
    return check_launch("hadamard_fast");
}

template <int DT, bool QUANT, bool ACT>
static int launch_fast_k(const HadArgs &p, hipStream_t st)
{
    switch (p.K) {
    case 12: return launch_fast_t<DT, QUANT, ACT, 1, 1>(p, st);
    case 20: case 28: return launch_fast_t<DT, QUANT, ACT, 2, 1>(p, st);
    case 36: case 40: return launch_fast_t<DT, QUANT, ACT, 3, 2>(p, st);
    case 52: case 60: return launch_fast_t<DT, QUANT, ACT, 4, 2>(p, st);
    case 108: return launch_fast_t<DT, QUANT, ACT, 7, 4>(p, st);
    case 140: return launch_fast_t<DT, QUANT, ACT, 9, 5>(p, st);
    case 156: return launch_fast_t<DT, QUANT, ACT, 10, 5>(p, st);
    case 172: return launch_fast_t<DT, QUANT, ACT, 11, 6>(p, st);
    default: return MQ_EUNSUPPORTED;
    }
}

// Returns MQ_EUNSUPPORTED (without setting an error message) when the shape / dtype is outside this mode: the
// caller then runs the exact kernel.
int hadamard_fast_dispatch(const HadArgs &p, int x_dtype, bool quant, hipStream_t st)
{
    if (p.K <= 1 || p.fp32_had || x_dtype == MQ_F32 || p.m < 64 || p.m > 512 || (p.K % 4) != 0) return MQ_EUNSUPPORTED;
    const bool act = p.act != MQ_ACT_NONE;
    if (act && !quant) return MQ_EUNSUPPORTED;
    if (x_dtype == MQ_F16) {
        if (quant) return act ? launch_fast_k<MQ_F16, true, true>(p, st) : launch_fast_k<MQ_F16, true, false>(p, st);
        return launch_fast_k<MQ_F16, false, false>(p, st);
    }
    if (quant) return act ? launch_fast_k<MQ_BF16, true, true>(p, st) : launch_fast_k<MQ_BF16, true, false>(p, st);
    return launch_fast_k<MQ_BF16, false, false>(p, st);
}

}  // namespace mq
