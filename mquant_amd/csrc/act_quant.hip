// act_quant.hip -- static activation quantizer (fp16/bf16/fp32 -> int8 levels, or the
// fused quantize->dequantize form).  HBM-bound: 2 B in + 1 B out per element for half
// inputs.  One thread owns 16 consecutive channels of one row: two 16-byte loads, one
// 16-byte store, rows are walked by a grid-stride loop.
//
// Reference semantics: fake_quant/quantizer/uniform.py:20-43, base.py:44-50.
#include "mq_common.h"

namespace mq {

template <int DT, bool DEQUANT>
__global__ __launch_bounds__(256) void act_quant_kernel(
    const typename Elem<DT>::T *__restrict__ x, long M, long K, long ldx,
    float scale0, float scale1, const float *__restrict__ svec0,
    const float *__restrict__ svec1, const uint8_t *__restrict__ row_sel, int skip_col0,
    float *__restrict__ x0_out, void *__restrict__ out_, long K_pad, long ldo, int vec_ok)
{
    typedef typename Elem<DT>::T T;
    const long chunks_per_row = K_pad / 16;
    const long total = (!DEQUANT && ldo == MQ_LD_TILED) ? ((M + 15) / 16) * 16 * chunks_per_row : M * chunks_per_row;
    for (long c = (long)blockIdx.x * blockDim.x + threadIdx.x; c < total;
         c += (long)gridDim.x * blockDim.x) {
        long row, col;
        if (!DEQUANT && ldo == MQ_LD_TILED) {
            // tiled output: a wave writes ONE contiguous 1 KiB piece (16 rows x 64 channels, lane =
            // 16 chunk + row) instead of 64 chunks scattered over 16 pieces; its reads become a
            // 16-row gather of 128-byte segments
            const unsigned piece = (unsigned)(c >> 6), kts = (unsigned)(K_pad >> 6);   // < 2^31 pieces (checked by the launcher)
            const int l = (int)(c & 63);
            const unsigned mt = piece / kts;             // uniform per wave
            row = (long)mt * 16 + (l & 15);
            col = (long)(piece - mt * kts) * 64 + (l >> 4) * 16;
            if (row >= M) continue;
        } else if (total < (1L << 31)) {                // 32-bit divide: a 64-bit one costs ~100 cycles per thread
            const unsigned r32 = (unsigned)c / (unsigned)chunks_per_row;
            row = r32;
            col = (long)((unsigned)c - r32 * (unsigned)chunks_per_row) * 16;
        } else {
            row = c / chunks_per_row;
            col = (c - row * chunks_per_row) * 16;
        }
        const int sel = row_sel ? (row_sel[row] != 0) : 0;
        const float s_t = sel ? scale1 : scale0;
        const float *sv = sel ? svec1 : svec0;
        const T *xr = x + row * ldx + col;

        float v[16];
        if (col + 16 <= K && vec_ok) {
            if (sizeof(T) == 2) {
                const v8us a = *reinterpret_cast<const v8us *>(xr);
                const v8us b = *reinterpret_cast<const v8us *>(xr + 8);
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    v[i] = Elem<DT>::ld((T)a[i]);
                    v[8 + i] = Elem<DT>::ld((T)b[i]);
                }
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const v4f a = *reinterpret_cast<const v4f *>((const float *)xr + 4 * j);
#pragma unroll
                    for (int i = 0; i < 4; ++i) v[4 * j + i] = a[i];
                }
            }
        } else {
#pragma unroll
            for (int i = 0; i < 16; ++i) v[i] = (col + i < K) ? Elem<DT>::ld(xr[i]) : 0.0f;
        }

        int q[16];
        if (sv) {
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const float s = (col + i < K) ? sv[col + i] : 1.0f;
                q[i] = quant_level(v[i], s, -128.0f, 127.0f);
            }
        } else {
            // one scale per row: x * (1 / s) with the exact quotient only next to a half-integer (mq_common.h) -- the
            // same levels as the IEEE division for a third of its instructions
            quant_levels<16>(v, s_t, 1.0f / s_t, quant_rcp_ok(s_t), -128.0f, 127.0f, q);
        }
        if (skip_col0 && col == 0) {
            if (x0_out) x0_out[row] = v[0];
            q[0] = 0;
        }

        if (!DEQUANT) {
            int8_t *out = reinterpret_cast<int8_t *>(out_) + act_offset(row, col, K_pad, ldo);
            v4i p;
#pragma unroll
            for (int j = 0; j < 4; ++j)
                p[j] = (q[4 * j] & 0xff) | ((q[4 * j + 1] & 0xff) << 8) |
                       ((q[4 * j + 2] & 0xff) << 16) | ((q[4 * j + 3] & 0xff) << 24);
            *reinterpret_cast<v4i *>(out) = p;
        } else {
            T *out = reinterpret_cast<T *>(out_) + row * ldo + col;
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                if (col + i >= K) break;
                const float s = sv ? sv[col + i] : s_t;
                float d = (float)q[i] * s;
                if (skip_col0 && col == 0 && i == 0) d = v[0];
                out[i] = Elem<DT>::st(d);
            }
        }
    }
}

// The prefill's case on its own kernel: one scale per row (two with the token-type mask), int8 levels into the tiled
// image, 16-byte aligned rows, K a multiple of 16.  A wave owns one 1 KiB piece (EPT = 16: lane = 16 chunk + row) or
// half of one (EPT = 8: lane = 32 chunk + 2 row + half chunk) and writes it as ONE contiguous run; the piece comes from
// the grid (x: four k-units per workgroup, y: 16-row tile), so there is no division, and 1 / s comes from the host.
struct AqTiledArgs {
    const void *x;
    long M, K, ldx;
    float s0, s1, inv0, inv1;
    int rcp0, rcp1;
    const uint8_t *row_sel;
    int skip_col0;
    float *x0_out;
    int8_t *out;
    int kts;
};

#ifndef MQ_AQ_EPT
#define MQ_AQ_EPT 8
#endif
#ifndef MQ_AQ_THREADS
#define MQ_AQ_THREADS 256
#endif

template <int DT, int EPT, int THREADS>
__global__ __launch_bounds__(THREADS) void act_quant_tiled_kernel(AqTiledArgs p)
{
    kernarg_warm<sizeof(AqTiledArgs)>();     // one scalar-load round trip instead of five (mq_common.h)
    typedef typename Elem<DT>::T T;
    constexpr int WPP = 16 / EPT;                                  // waves per piece
    constexpr int SUB = 16 / EPT;                                  // lanes per 16-byte chunk of the image
    constexpr int CPW = EPT / 4;                                   // chunks (of 16 rows) per wave
    const int lane = threadIdx.x & 63;
    const int unit = blockIdx.x * (THREADS / 64) + (threadIdx.x >> 6);   // kt * WPP + part, wave-uniform
    if (unit >= p.kts * WPP) return;
    const int kt = unit / WPP, part = unit % WPP;
    const int mt = blockIdx.y;
    const int r = (lane / SUB) & 15;
    const int col = kt * 64 + (part * CPW + lane / (SUB * 16)) * 16 + (lane % SUB) * EPT;
    const long row = (long)mt * 16 + r;
    if (row >= p.M) return;
    const bool sel = p.row_sel && p.row_sel[row] != 0;             // requested ahead of the activations: it returns first
    int8_t *o = p.out + ((long)mt * p.kts + kt) * 1024 + part * (64 * EPT) + lane * EPT;
    if (col >= p.K) {                                               // pad columns of the image
        if (EPT == 16) *reinterpret_cast<v4i *>(o) = v4i{0, 0, 0, 0};
        else if (EPT == 8) *reinterpret_cast<v2i *>(o) = v2i{0, 0};
        else *reinterpret_cast<int *>(o) = 0;
        return;
    }
    const T *xr = reinterpret_cast<const T *>(p.x) + row * p.ldx + col;
    float v[EPT];
    if (sizeof(T) == 2 && EPT == 4) {
        const v4us a = *reinterpret_cast<const v4us *>(xr);
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = Elem<DT>::ld((T)a[i]);
    } else if (sizeof(T) == 2) {
#pragma unroll
        for (int j = 0; j < EPT / 8; ++j) {
            const v8us a = *reinterpret_cast<const v8us *>(xr + 8 * j);
#pragma unroll
            for (int i = 0; i < 8; ++i) v[8 * j + i] = Elem<DT>::ld((T)a[i]);
        }
    } else {
#pragma unroll
        for (int j = 0; j < EPT / 4; ++j) {
            const v4f a = *reinterpret_cast<const v4f *>((const float *)xr + 4 * j);
#pragma unroll
            for (int i = 0; i < 4; ++i) v[4 * j + i] = a[i];
        }
    }
    unsigned w[EPT / 4];
    quant_levels_i8_packed<EPT>(v, sel ? p.s1 : p.s0, sel ? p.inv1 : p.inv0, (sel ? p.rcp1 : p.rcp0) != 0, w);
    if (p.skip_col0 && col == 0) {
        if (p.x0_out) p.x0_out[row] = v[0];
        w[0] &= 0xffffff00u;
    }
    if (EPT == 16) *reinterpret_cast<v4i *>(o) = v4i{(int)w[0], (int)w[1], (int)w[2], (int)w[3]};
    else if (EPT == 8) *reinterpret_cast<v2i *>(o) = v2i{(int)w[0], (int)w[EPT / 4 - 1]};
    else *reinterpret_cast<unsigned *>(o) = w[0];
}

template <int DT, bool DEQUANT>
static int launch_act_quant(const void *x, long M, long K, long ldx, float scale0, float scale1,
                            const float *sv0, const float *sv1, const uint8_t *row_sel,
                            int skip_col0, float *x0_out, void *out, long K_pad, long ldo,
                            hipStream_t st)
{
    typedef typename Elem<DT>::T T;
    const long total = ((!DEQUANT && ldo == MQ_LD_TILED) ? ((M + 15) / 16) * 16 : M) * (K_pad / 16);
    if (total == 0) return MQ_OK;
    const int vec_ok = (((uintptr_t)x) % 16 == 0) && ((ldx * (long)sizeof(T)) % 16 == 0);
#ifdef MQ_AQ_GENERIC_ONLY   // A/B builds (tools/bench_ab.sh): everything on the general kernel, as before round 4
    const bool own_kernel = false;
#else
    const bool own_kernel = true;
#endif
    if (own_kernel && !DEQUANT && ldo == MQ_LD_TILED && !sv0 && !sv1 && vec_ok && K % 16 == 0 && (M + 15) / 16 <= 65535) {
        constexpr int EPT = MQ_AQ_EPT;
        AqTiledArgs a{x, M, K, ldx, scale0, scale1, 1.0f / scale0, 1.0f / scale1, quant_rcp_ok(scale0), quant_rcp_ok(scale1),
                      row_sel, skip_col0, x0_out, (int8_t *)out, (int)(K_pad / 64)};
        constexpr int THREADS = MQ_AQ_THREADS;
        const unsigned gx = (unsigned)ceil_div(K_pad / 64 * (16 / EPT), THREADS / 64);
        hipLaunchKernelGGL((act_quant_tiled_kernel<DT, EPT, THREADS>), dim3(gx, (unsigned)((M + 15) / 16)), dim3(THREADS), 0, st, a);
        return check_launch("act_quant_tiled");
    }
    long blocks = ceil_div(total, 256);
    if (blocks > 256L * 16) blocks = 256L * 16;
    hipLaunchKernelGGL((act_quant_kernel<DT, DEQUANT>), dim3((unsigned)blocks), dim3(256), 0, st,
                       (const T *)x, M, K, ldx, scale0, scale1, sv0, sv1, row_sel, skip_col0,
                       x0_out, out, K_pad, ldo, vec_ok);
    return check_launch("act_quant");
}

}  // namespace mq

extern "C" int mq_quantize_act_i8(const void *x, int x_dtype, long M, long K, long ldx,
                                  float scale0, float scale1, const float *scale_vec0,
                                  const float *scale_vec1, const uint8_t *row_sel,
                                  int skip_col0, float *x0_out, int8_t *out, long K_pad,
                                  long ldo, void *stream)
{
    using namespace mq;
    MQ_REQUIRE(M >= 0 && K >= 0, "mq_quantize_act_i8: negative shape");
    if (M == 0 || K == 0) return MQ_OK;
    MQ_REQUIRE(x && out, "mq_quantize_act_i8: null buffer");
    MQ_REQUIRE(K_pad >= K && K_pad % 16 == 0, "mq_quantize_act_i8: K_pad=%ld must be >= K=%ld and a multiple of 16", K_pad, K);
    MQ_REQUIRE(((uintptr_t)out) % 16 == 0 && (ldo == MQ_LD_TILED ? K_pad % 64 == 0 : (ldo >= K_pad && ldo % 16 == 0)),
               "mq_quantize_act_i8: out must be 16-byte aligned with ldo %% 16 == 0, or ldo = MQ_LD_TILED with K_pad %% 64 == 0 (ldo=%ld)", ldo);
    MQ_REQUIRE(ldx >= K, "mq_quantize_act_i8: ldx < K");
    MQ_REQUIRE(!row_sel || !scale_vec0 || scale_vec1, "mq_quantize_act_i8: row_sel with per-channel scales needs scale_vec1");
    if (!scale_vec1) scale_vec1 = scale_vec0;
    hipStream_t st = (hipStream_t)stream;
    switch (x_dtype) {
    case MQ_F16: return launch_act_quant<MQ_F16, false>(x, M, K, ldx, scale0, scale1, scale_vec0, scale_vec1, row_sel, skip_col0, x0_out, out, K_pad, ldo, st);
    case MQ_BF16: return launch_act_quant<MQ_BF16, false>(x, M, K, ldx, scale0, scale1, scale_vec0, scale_vec1, row_sel, skip_col0, x0_out, out, K_pad, ldo, st);
    case MQ_F32: return launch_act_quant<MQ_F32, false>(x, M, K, ldx, scale0, scale1, scale_vec0, scale_vec1, row_sel, skip_col0, x0_out, out, K_pad, ldo, st);
    }
    return fail(MQ_EINVAL, "mq_quantize_act_i8: unknown dtype %d", x_dtype);
}

extern "C" int mq_fakequant_act(const void *x, int x_dtype, long M, long K, long ldx,
                                float scale0, float scale1, const float *scale_vec0,
                                const float *scale_vec1, const uint8_t *row_sel, int skip_col0,
                                void *out, long ldo, void *stream)
{
    using namespace mq;
    MQ_REQUIRE(M >= 0 && K >= 0, "mq_fakequant_act: negative shape");
    if (M == 0 || K == 0) return MQ_OK;
    MQ_REQUIRE(x && out, "mq_fakequant_act: null buffer");
    MQ_REQUIRE(ldx >= K && ldo >= K, "mq_fakequant_act: leading dimension < K");
    if (!scale_vec1) scale_vec1 = scale_vec0;
    const long K_pad = ceil_div(K, 16) * 16;
    hipStream_t st = (hipStream_t)stream;
    switch (x_dtype) {
    case MQ_F16: return launch_act_quant<MQ_F16, true>(x, M, K, ldx, scale0, scale1, scale_vec0, scale_vec1, row_sel, skip_col0, nullptr, out, K_pad, ldo, st);
    case MQ_BF16: return launch_act_quant<MQ_BF16, true>(x, M, K, ldx, scale0, scale1, scale_vec0, scale_vec1, row_sel, skip_col0, nullptr, out, K_pad, ldo, st);
    case MQ_F32: return launch_act_quant<MQ_F32, true>(x, M, K, ldx, scale0, scale1, scale_vec0, scale_vec1, row_sel, skip_col0, nullptr, out, K_pad, ldo, st);
    }
    return fail(MQ_EINVAL, "mq_fakequant_act: unknown dtype %d", x_dtype);
}
