// act_quant_dyn.hip -- the reference's DEFAULT activation quantizer: dynamic, symmetric, per token
// (ActQuantizer.find_params + forward, fake_quant/quant_utils.py:116-133,205-268 with
// act_per_tensor = False, groupsize = -1, sym = True), as int8 levels + one fp32 scale per row:
//     xmin = min(min_k x, 0) * clip;  xmax = max(max_k x, 0) * clip      (fp32: the `tmp` of :239
//     s    = max(|xmin|, xmax) / maxq,  s = 1 where that is 0              promotes half tensors)
//     q    = clamp(rint(x / s), -(maxq+1), maxq)
// One workgroup per row, the row stays in registers between the max pass and the quantize pass.
// skip_col0 (ActQuantWrapper.split, :367-372): column 0 is excluded from the range, comes back in
// x0_out as fp32 and gets level 0.
#include "mq_common.h"

namespace mq {

constexpr int DQ_THREADS = 256;
constexpr int DQ_MAX_CHUNKS = 8;   // rows up to 32768 elements

struct DqArgs {
    const void *x;
    long M, K, ldx;
    float clip, maxq;
    int skip_col0;
    float *x0_out, *scale_out;
    float *zero_out, *shift_out;   // asymmetric mode: zero point and scale * (2^(bits-1) - zero) per row
    float half;                    // asymmetric mode: 2^(bits-1), the offset of the stored levels
    const float *range_in;         // per-tensor mode: device [min, max] of the whole tensor (mq_minmax_tensor), else NULL
    int8_t *out;
    long K_pad, ldo;
    int vec_ok;
};

template <int DT, bool ASYM = false>
__global__ __launch_bounds__(DQ_THREADS) void act_quant_dyn_kernel(DqArgs p)
{
    kernarg_warm<sizeof(DqArgs)>();
    typedef typename Elem<DT>::T T;
    __shared__ float rmin[DQ_THREADS / 64], rmax[DQ_THREADS / 64];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const long row = blockIdx.x;
    const T *xr = reinterpret_cast<const T *>(p.x) + row * p.ldx;
    const long chunks = ceil_div(p.K, 16);

    float v[DQ_MAX_CHUNKS][16];
    float mn = 0.0f, mx = 0.0f;
#pragma unroll
    for (int c = 0; c < DQ_MAX_CHUNKS; ++c) {
        const long ch = t + (long)c * DQ_THREADS;
        if (ch < chunks) {
            const long col = ch * 16;
            const T *src = xr + col;
            if (col + 16 <= p.K && p.vec_ok) {
                if (sizeof(T) == 2) {
                    const v8us a = *reinterpret_cast<const v8us *>(src);
                    const v8us b = *reinterpret_cast<const v8us *>(src + 8);
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        v[c][i] = Elem<DT>::ld((T)a[i]);
                        v[c][8 + i] = Elem<DT>::ld((T)b[i]);
                    }
                } else {
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const v4f a = *reinterpret_cast<const v4f *>((const float *)src + 4 * j);
#pragma unroll
                        for (int i = 0; i < 4; ++i) v[c][4 * j + i] = a[i];
                    }
                }
            } else {
#pragma unroll
                for (int i = 0; i < 16; ++i) v[c][i] = (col + i < p.K) ? Elem<DT>::ld(src[i]) : 0.0f;
            }
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                if (p.skip_col0 && col + i == 0) continue;
                mn = fminf(mn, v[c][i]);
                mx = fmaxf(mx, v[c][i]);
            }
        }
    }
#pragma unroll
    for (int st = 1; st < 64; st <<= 1) {
        mn = fminf(mn, __shfl_xor(mn, st, 64));
        mx = fmaxf(mx, __shfl_xor(mx, st, 64));
    }
    if (lane == 0) { rmin[wave] = mn; rmax[wave] = mx; }
    __syncthreads();
#pragma unroll
    for (int w = 0; w < DQ_THREADS / 64; ++w) { mn = fminf(mn, rmin[w]); mx = fmaxf(mx, rmax[w]); }
    if (p.range_in) {                                   // act_per_tensor (quant_utils.py:214-237): one range for all rows
        mn = fminf(p.range_in[0], 0.0f);
        mx = fmaxf(p.range_in[1], 0.0f);
    }
    // The per-TENSOR rule is evaluated in x's dtype upstream (quant_utils.py:214-231: torch.tensor(0).to(x), the int64
    // maxq tensor does not promote): range * clip, the scale, the zero point, x / scale and the level sum are 16-bit
    // tensors for a half model, each torch op rounding its fp32 result once.  The per-token rule promotes to fp32 (:239).
    const bool faithful = (DT != MQ_F32) && p.range_in != nullptr;
    auto rd = [&](float v) { return faithful ? Elem<DT>::rnd(v) : v; };
    float xmin = rd(mn * p.clip), xmax0 = rd(mx * p.clip);
    float s, zero = 0.0f, lo, hi;
    if (ASYM) {                                         // quant_utils.py:255-268 + asym_quant :27-31
        if (p.range_in) {                               // the per-tensor rule fixes each bound on its own (:229-232)
            if (xmin == 0.0f) xmin = -1.0f;
            if (xmax0 == 0.0f) xmax0 = 1.0f;
        } else if (xmin == 0.0f && xmax0 == 0.0f) { xmin = -1.0f; xmax0 = 1.0f; }
        s = rd(rd(xmax0 - xmin) / p.maxq);
        zero = rintf(rd(-xmin / s));
        lo = 0.0f;
        hi = p.maxq;
        if (t == 0) {
            if (p.scale_out) p.scale_out[row] = s;
            if (p.zero_out) p.zero_out[row] = zero;
            if (p.shift_out) p.shift_out[row] = s * (p.half - zero);
        }
    } else {
        const float xmax = fmaxf(fabsf(xmin), xmax0);
        s = (xmax == 0.0f) ? 1.0f : rd(xmax / p.maxq);
        if (t == 0 && p.scale_out) p.scale_out[row] = s;
        lo = -(p.maxq + 1.0f);
        hi = p.maxq;
    }

#pragma unroll
    for (int c = 0; c < DQ_MAX_CHUNKS; ++c) {
        const long ch = t + (long)c * DQ_THREADS;
        if (ch < chunks) {
            int q[16];
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                if (ASYM) {      // clamp(rint(x / s) + zero, 0, maxq), stored minus 2^(bits-1); pad columns 0 (their weights are 0)
                    float lv = rd(rintf(rd(v[c][i] / s)) + zero);
                    lv = fminf(fmaxf(lv, lo), hi);
                    q[i] = (ch * 16 + i < p.K) ? (int)(lv - p.half) : 0;
                } else if (faithful) {
                    const float lv = fminf(fmaxf(rintf(Elem<DT>::rnd(v[c][i] / s)), lo), hi);
                    q[i] = (ch * 16 + i < p.K) ? (int)lv : 0;
                } else {
                    q[i] = (ch * 16 + i < p.K) ? quant_level(v[c][i], s, lo, hi) : 0;
                }
            }
            if (p.skip_col0 && ch == 0) {
                if (p.x0_out) p.x0_out[row] = v[c][0];
                q[0] = 0;
            }
            v4i pk;
#pragma unroll
            for (int j = 0; j < 4; ++j)
                pk[j] = (q[4 * j] & 0xff) | ((q[4 * j + 1] & 0xff) << 8) | ((q[4 * j + 2] & 0xff) << 16) |
                        ((q[4 * j + 3] & 0xff) << 24);
            *reinterpret_cast<v4i *>(p.out + act_offset(row, ch * 16, p.K_pad, p.ldo)) = pk;
        }
    }
    for (long k = chunks * 16 + t * 16L; k < p.K_pad; k += DQ_THREADS * 16L)
        *reinterpret_cast<v4i *>(p.out + act_offset(row, k, p.K_pad, p.ldo)) = v4i{0, 0, 0, 0};
}

}  // namespace mq

extern "C" int mq_quantize_act_dyn_i8(const void *x, int x_dtype, long M, long K, long ldx, int bits,
                                      float clip_ratio, int skip_col0, float *x0_out, float *scale_out,
                                      int8_t *out, long K_pad, long ldo, void *stream)
{
    using namespace mq;
    if (M == 0) return MQ_OK;                       // empty input: nothing to do (null pointers allowed)
    MQ_REQUIRE(x && out && scale_out && M >= 0 && K > 0 && ldx >= K, "mq_quantize_act_dyn_i8: bad shape");
    MQ_REQUIRE(bits >= 2 && bits <= 8, "mq_quantize_act_dyn_i8: bits must be 2..8");
    MQ_REQUIRE(K <= 16L * DQ_THREADS * DQ_MAX_CHUNKS, "mq_quantize_act_dyn_i8: K=%ld too large (max %d)", K, 16 * DQ_THREADS * DQ_MAX_CHUNKS);
    MQ_REQUIRE(K_pad >= K && K_pad % 16 == 0 && ((uintptr_t)out) % 16 == 0 &&
                   (ldo == MQ_LD_TILED ? K_pad % 64 == 0 : (ldo >= K_pad && ldo % 16 == 0)),
               "mq_quantize_act_dyn_i8: bad K_pad / ldo / alignment");
    if (M == 0) return MQ_OK;
    DqArgs p;
    p.x = x; p.M = M; p.K = K; p.ldx = ldx; p.clip = clip_ratio; p.maxq = (float)((1 << (bits - 1)) - 1);
    p.skip_col0 = skip_col0; p.x0_out = x0_out; p.scale_out = scale_out; p.out = out; p.K_pad = K_pad; p.ldo = ldo;
    p.zero_out = nullptr; p.shift_out = nullptr; p.half = 0.0f; p.range_in = nullptr;
    const size_t esz = (x_dtype == MQ_F32) ? 4 : 2;
    p.vec_ok = (((uintptr_t)x) % 16 == 0) && ((ldx * esz) % 16 == 0);
    hipStream_t st = (hipStream_t)stream;
    switch (x_dtype) {
    case MQ_F16: hipLaunchKernelGGL(act_quant_dyn_kernel<MQ_F16>, dim3((unsigned)M), dim3(DQ_THREADS), 0, st, p); break;
    case MQ_BF16: hipLaunchKernelGGL(act_quant_dyn_kernel<MQ_BF16>, dim3((unsigned)M), dim3(DQ_THREADS), 0, st, p); break;
    case MQ_F32: hipLaunchKernelGGL(act_quant_dyn_kernel<MQ_F32>, dim3((unsigned)M), dim3(DQ_THREADS), 0, st, p); break;
    default: return fail(MQ_EINVAL, "mq_quantize_act_dyn_i8: unknown dtype %d", x_dtype);
    }
    return check_launch("quantize_act_dyn_i8");
}

/* Dynamic ASYMMETRIC per-token quantizer (--a_asym without --*_static; quant_utils.py:239-268 else-branch):
 *   scale = (xmax - xmin) / (2^bits - 1), zero = rint(-xmin / scale), q = clamp(rint(x / scale) + zero, 0, 2^bits - 1)
 * The int8 GEMM gets q - 2^(bits-1); shift_out[m] = scale * (2^(bits-1) - zero) is the row factor of the
 * rank-1 term that undoes the offset and the zero point: x_hat = scale * stored + shift, so
 *   y[m][n] = ((acc * scale[m]) * s_w[n]) + bias[n] + shift[m] * (s_w[n] * sum_k q_w[n][k])
 * (mq_gemm_w4a8_rowscale_ws with x0 = shift_out and w0 = s_w * column sums of the weight levels). */
extern "C" int mq_quantize_act_dyn_asym_i8(const void *x, int x_dtype, long M, long K, long ldx, int bits,
                                           float clip_ratio, float *scale_out, float *zero_out, float *shift_out,
                                           int8_t *out, long K_pad, long ldo, void *stream)
{
    using namespace mq;
    if (M == 0) return MQ_OK;
    MQ_REQUIRE(x && out && scale_out && shift_out && M >= 0 && K > 0 && ldx >= K, "mq_quantize_act_dyn_asym_i8: bad shape");
    MQ_REQUIRE(bits >= 2 && bits <= 8, "mq_quantize_act_dyn_asym_i8: bits must be 2..8");
    MQ_REQUIRE(K <= 16L * DQ_THREADS * DQ_MAX_CHUNKS, "mq_quantize_act_dyn_asym_i8: K=%ld too large (max %d)", K, 16 * DQ_THREADS * DQ_MAX_CHUNKS);
    MQ_REQUIRE(K_pad >= K && K_pad % 16 == 0 && ((uintptr_t)out) % 16 == 0 &&
                   (ldo == MQ_LD_TILED ? K_pad % 64 == 0 : (ldo >= K_pad && ldo % 16 == 0)),
               "mq_quantize_act_dyn_asym_i8: bad K_pad / ldo / alignment");
    DqArgs p;
    p.x = x; p.M = M; p.K = K; p.ldx = ldx; p.clip = clip_ratio; p.maxq = (float)((1 << bits) - 1);
    p.half = (float)(1 << (bits - 1));
    p.skip_col0 = 0; p.x0_out = nullptr; p.scale_out = scale_out; p.zero_out = zero_out; p.shift_out = shift_out; p.range_in = nullptr;
    p.out = out; p.K_pad = K_pad; p.ldo = ldo;
    const size_t esz = (x_dtype == MQ_F32) ? 4 : 2;
    p.vec_ok = (((uintptr_t)x) % 16 == 0) && ((ldx * esz) % 16 == 0);
    hipStream_t st = (hipStream_t)stream;
    switch (x_dtype) {
    case MQ_F16: hipLaunchKernelGGL((act_quant_dyn_kernel<MQ_F16, true>), dim3((unsigned)M), dim3(DQ_THREADS), 0, st, p); break;
    case MQ_BF16: hipLaunchKernelGGL((act_quant_dyn_kernel<MQ_BF16, true>), dim3((unsigned)M), dim3(DQ_THREADS), 0, st, p); break;
    case MQ_F32: hipLaunchKernelGGL((act_quant_dyn_kernel<MQ_F32, true>), dim3((unsigned)M), dim3(DQ_THREADS), 0, st, p); break;
    default: return fail(MQ_EINVAL, "mq_quantize_act_dyn_asym_i8: unknown dtype %d", x_dtype);
    }
    return check_launch("quantize_act_dyn_asym_i8");
}

/* Dynamic PER-TENSOR quantizer (act_per_tensor = True; quant_utils.py:214-237): the range of the whole
 * tensor comes from mq_minmax_tensor (device [min, max], col_begin = 1 under skip_col0), so nothing
 * returns to the host.  Symmetric or asymmetric as above, except that the asymmetric rule replaces a
 * zero bound on its own (xmin == 0 -> -1; xmax == 0 -> +1).  scale_out / zero_out / shift_out are
 * written per row (all rows equal) for mq_gemm_w4a8_rowscale_ws. */
extern "C" int mq_quantize_act_range_i8(const void *x, int x_dtype, long M, long K, long ldx, int bits,
                                        float clip_ratio, int asym, int skip_col0, const float *minmax,
                                        float *x0_out, float *scale_out, float *zero_out, float *shift_out,
                                        int8_t *out, long K_pad, long ldo, void *stream)
{
    using namespace mq;
    if (M == 0) return MQ_OK;
    MQ_REQUIRE(x && out && scale_out && minmax && M >= 0 && K > 0 && ldx >= K, "mq_quantize_act_range_i8: bad shape");
    MQ_REQUIRE(!asym || (shift_out && !skip_col0), "mq_quantize_act_range_i8: the asymmetric mode needs shift_out and no split column");
    MQ_REQUIRE(bits >= 2 && bits <= 8, "mq_quantize_act_range_i8: bits must be 2..8");
    MQ_REQUIRE(K <= 16L * DQ_THREADS * DQ_MAX_CHUNKS, "mq_quantize_act_range_i8: K=%ld too large (max %d)", K, 16 * DQ_THREADS * DQ_MAX_CHUNKS);
    MQ_REQUIRE(K_pad >= K && K_pad % 16 == 0 && ((uintptr_t)out) % 16 == 0 &&
                   (ldo == MQ_LD_TILED ? K_pad % 64 == 0 : (ldo >= K_pad && ldo % 16 == 0)),
               "mq_quantize_act_range_i8: bad K_pad / ldo / alignment");
    DqArgs p;
    p.x = x; p.M = M; p.K = K; p.ldx = ldx; p.clip = clip_ratio;
    p.maxq = asym ? (float)((1 << bits) - 1) : (float)((1 << (bits - 1)) - 1);
    p.half = asym ? (float)(1 << (bits - 1)) : 0.0f;
    p.skip_col0 = skip_col0; p.x0_out = x0_out; p.scale_out = scale_out; p.zero_out = zero_out; p.shift_out = shift_out;
    p.range_in = minmax; p.out = out; p.K_pad = K_pad; p.ldo = ldo;
    const size_t esz = (x_dtype == MQ_F32) ? 4 : 2;
    p.vec_ok = (((uintptr_t)x) % 16 == 0) && ((ldx * esz) % 16 == 0);
    hipStream_t st = (hipStream_t)stream;
#define MQ_DQ_LAUNCH(DT)                                                                                              \
    do {                                                                                                              \
        if (asym) hipLaunchKernelGGL((act_quant_dyn_kernel<DT, true>), dim3((unsigned)M), dim3(DQ_THREADS), 0, st, p); \
        else hipLaunchKernelGGL((act_quant_dyn_kernel<DT, false>), dim3((unsigned)M), dim3(DQ_THREADS), 0, st, p);    \
    } while (0)
    switch (x_dtype) {
    case MQ_F16: MQ_DQ_LAUNCH(MQ_F16); break;
    case MQ_BF16: MQ_DQ_LAUNCH(MQ_BF16); break;
    case MQ_F32: MQ_DQ_LAUNCH(MQ_F32); break;
    default: return fail(MQ_EINVAL, "mq_quantize_act_range_i8: unknown dtype %d", x_dtype);
    }
#undef MQ_DQ_LAUNCH
    return check_launch("quantize_act_range_i8");
}

/* Dynamic symmetric GROUP-WISE quantizer (--a_groupsize g; quant_utils.py:181-203 + sym_quant :46-50): one scale
 * per row and group of g consecutive channels,
 *     xmax = amax * clip; xmin = amin * clip (no zero inclusion); xmax = max(|xmin|, xmax)
 *     scale = xmax / maxq, 1 where xmax == 0;    q = clamp(rint(x / scale), -(maxq+1), maxq)
 * with every intermediate rounded to x's dtype like the reference's tensors of that dtype (amax * clip, the quotient
 * by maxq and x / scale are half tensors for a half model).  A thread owns 16 consecutive channels, the g / 16 lanes
 * of a group sit side by side in one wave and combine their extrema with shuffles.  scale_out: fp32 [M, K / g]. */
namespace mq {

struct GqArgs {
    const void *x;
    long M, K, ldx;
    int lanes_per_group;           // g / 16: 1 .. 64, a power of two
    float clip, maxq;
    float half = 0.0f;             // asymmetric levels: 2^(bits-1), maxq = 2^bits - 1
    float *scale_out;
    float *zero_out = nullptr, *shift_out = nullptr;
    long n_groups;
    int8_t *out;
    long K_pad, ldo;
    int vec_ok;
};

template <int DT, bool ASYM = false>
__global__ __launch_bounds__(256) void act_quant_group_kernel(GqArgs p)
{
    kernarg_warm<sizeof(GqArgs), true>();
    typedef typename Elem<DT>::T T;
    const long cpr = p.K_pad / 16;                                  // 16-channel chunks per row (a multiple of lanes_per_group)
    const long total = p.M * cpr;
    const long span = ((total + 63) / 64) * 64;                    // whole waves: every lane takes part in the shuffles
    for (long c = (long)blockIdx.x * 256 + threadIdx.x; c < span; c += (long)gridDim.x * 256) {
        const bool live = c < total;
        const long cc = live ? c : total - 1;
        const long row = cc / cpr;
        const long col = (cc - row * cpr) * 16;
        const T *xr = reinterpret_cast<const T *>(p.x) + row * p.ldx + col;
        float v[16];
        if (col + 16 <= p.K && p.vec_ok) {
            if (sizeof(T) == 2) {
                const v8us a = *reinterpret_cast<const v8us *>(xr);
                const v8us b = *reinterpret_cast<const v8us *>(xr + 8);
#pragma unroll
                for (int i = 0; i < 8; ++i) { v[i] = Elem<DT>::ld((T)a[i]); v[8 + i] = Elem<DT>::ld((T)b[i]); }
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
            for (int i = 0; i < 16; ++i) v[i] = (col + i < p.K) ? Elem<DT>::ld(xr[i]) : 0.0f;
        }
        float mn = v[0], mx = v[0];
#pragma unroll
        for (int i = 1; i < 16; ++i) { mn = fminf(mn, v[i]); mx = fmaxf(mx, v[i]); }
        for (int st = 1; st < p.lanes_per_group; st <<= 1) {
            mn = fminf(mn, __shfl_xor(mn, st, 64));
            mx = fmaxf(mx, __shfl_xor(mx, st, 64));
        }
        float xmin = Elem<DT>::rnd(mn * p.clip), xmax0 = Elem<DT>::rnd(mx * p.clip);
        float s, z = 0.0f;
        if (ASYM) {      // quant_utils.py:181-203, sym = False: the range is [amin, amax] * clip (0 need not be inside), (-1, +1) when both are 0
            if (xmin == 0.0f && xmax0 == 0.0f) { xmin = -1.0f; xmax0 = 1.0f; }
            s = Elem<DT>::rnd(Elem<DT>::rnd(xmax0 - xmin) / p.maxq);
            z = rintf(Elem<DT>::rnd(-xmin / s));
        } else {
            const float xmax = fmaxf(fabsf(xmin), xmax0);
            s = (xmax == 0.0f) ? 1.0f : Elem<DT>::rnd(xmax / p.maxq);
        }
        if (!live) continue;
        const long chunk = cc - row * cpr;
        if ((chunk % p.lanes_per_group) == 0 && col < p.K) {
            const long gi = row * p.n_groups + chunk / p.lanes_per_group;
            p.scale_out[gi] = s;
            if (ASYM) {
                if (p.zero_out) p.zero_out[gi] = z;
                p.shift_out[gi] = s * (p.half - z);
            }
        }
        int q[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            float t = rintf(Elem<DT>::rnd(v[i] / s));
            if (ASYM) {
                t = Elem<DT>::rnd(t + z);
                t = fminf(fmaxf(t, 0.0f), p.maxq) - p.half;        // stored for the int8 GEMM: q - 2^(bits-1)
            } else {
                t = fminf(fmaxf(t, -(p.maxq + 1.0f)), p.maxq);
            }
            q[i] = (col + i < p.K) ? (int)t : 0;
        }
        v4i pk;
#pragma unroll
        for (int j = 0; j < 4; ++j)
            pk[j] = (q[4 * j] & 0xff) | ((q[4 * j + 1] & 0xff) << 8) | ((q[4 * j + 2] & 0xff) << 16) | ((q[4 * j + 3] & 0xff) << 24);
        *reinterpret_cast<v4i *>(p.out + act_offset(row, col, p.K_pad, p.ldo)) = pk;
    }
}

}  // namespace mq

static int quantize_act_group(const void *x, int x_dtype, long M, long K, long ldx, int groupsize, int bits, float clip_ratio,
                              bool asym, float *scale_out, float *zero_out, float *shift_out, int8_t *out, long K_pad, long ldo,
                              void *stream)
{
    using namespace mq;
    if (M == 0) return MQ_OK;
    MQ_REQUIRE(x && out && scale_out && (!asym || shift_out) && M >= 0 && K > 0 && ldx >= K, "mq_quantize_act_group_i8: bad shape");
    MQ_REQUIRE(bits >= 2 && bits <= 8, "mq_quantize_act_group_i8: bits must be 2..8");
    MQ_REQUIRE(groupsize >= 16 && groupsize <= 1024 && (groupsize & (groupsize - 1)) == 0 && K % groupsize == 0,
               "mq_quantize_act_group_i8: groupsize=%d must be a power of two in 16..1024 that divides K=%ld", groupsize, K);
    MQ_REQUIRE(K_pad >= K && K_pad % 16 == 0 && (K_pad / 16) % (groupsize / 16) == 0 && ((uintptr_t)out) % 16 == 0 &&
                   (ldo == MQ_LD_TILED ? K_pad % 64 == 0 : (ldo >= K_pad && ldo % 16 == 0)),
               "mq_quantize_act_group_i8: bad K_pad / ldo / alignment (K_pad must hold whole groups)");
    GqArgs p;
    p.x = x; p.M = M; p.K = K; p.ldx = ldx; p.lanes_per_group = groupsize / 16; p.clip = clip_ratio;
    p.maxq = asym ? (float)((1 << bits) - 1) : (float)((1 << (bits - 1)) - 1);
    p.half = (float)(1 << (bits - 1));
    p.zero_out = zero_out; p.shift_out = shift_out;
    p.scale_out = scale_out; p.n_groups = K / groupsize; p.out = out; p.K_pad = K_pad; p.ldo = ldo;
    const size_t esz = (x_dtype == MQ_F32) ? 4 : 2;
    p.vec_ok = (((uintptr_t)x) % 16 == 0) && ((ldx * esz) % 16 == 0);
    long blocks = ceil_div(M * (K_pad / 16), 256);
    if (blocks > 256L * 16) blocks = 256L * 16;
    hipStream_t st = (hipStream_t)stream;
#define MQ_GQ_LAUNCH(DTC) \
    do { \
        if (asym) hipLaunchKernelGGL((act_quant_group_kernel<DTC, true>), dim3((unsigned)blocks), dim3(256), 0, st, p); \
        else hipLaunchKernelGGL((act_quant_group_kernel<DTC, false>), dim3((unsigned)blocks), dim3(256), 0, st, p); \
    } while (0)
    switch (x_dtype) {
    case MQ_F16: MQ_GQ_LAUNCH(MQ_F16); break;
    case MQ_BF16: MQ_GQ_LAUNCH(MQ_BF16); break;
    case MQ_F32: MQ_GQ_LAUNCH(MQ_F32); break;
    default: return fail(MQ_EINVAL, "mq_quantize_act_group_i8: unknown dtype %d", x_dtype);
    }
#undef MQ_GQ_LAUNCH
    return check_launch("quantize_act_group_i8");
}

extern "C" int mq_quantize_act_group_i8(const void *x, int x_dtype, long M, long K, long ldx, int groupsize, int bits,
                                        float clip_ratio, float *scale_out, int8_t *out, long K_pad, long ldo, void *stream)
{
    return quantize_act_group(x, x_dtype, M, K, ldx, groupsize, bits, clip_ratio, false, scale_out, nullptr, nullptr, out, K_pad, ldo, stream);
}

extern "C" int mq_quantize_act_group_asym_i8(const void *x, int x_dtype, long M, long K, long ldx, int groupsize, int bits,
                                             float clip_ratio, float *scale_out, float *zero_out, float *shift_out, int8_t *out,
                                             long K_pad, long ldo, void *stream)
{
    return quantize_act_group(x, x_dtype, M, K, ldx, groupsize, bits, clip_ratio, true, scale_out, zero_out, shift_out, out, K_pad, ldo, stream);
}
