/*
 * mq_oracle.c -- CPU restatement of MQuant's W4A8 static-quant hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product:
 * only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library, and only as the checker.  The product path
 * (the .hip sources in mquant_amd/csrc behind include/mquant_hip.h) never calls it.
 *
 * Parity status: PINNED against outputs of the reference itself
 * (the .npz files in tests/golden, produced by tools/gen_golden.py which imports
 * /root/reference/fake_quant on CPU).  The one third-party boundary,
 * fast_hadamard_transform (Dao-AILab, un-pinned HEAD clone per
 * reference docs/install.md:13-19), is restated from its published
 * algorithm (in-register/warp/block butterflies in ascending stride, fp32,
 * scale applied on store) and cross-checked against the reference's in-tree
 * definition of the same operator, fake_quant/hadamard_utils.py:79-100.
 *
 * Every function cites the reference lines it follows (paths relative to
 * /root/reference).  Floating point is IEEE fp32 with one rounding per
 * written operation: build with -ffp-contract=off (see oracle/Makefile).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------ */
/* fp16 / bf16 round trips (round-to-nearest-even), used wherever the  */
/* reference casts back to x_dtype (quant_utils.py:336-341, base.py:49) */
/* ------------------------------------------------------------------ */
static inline uint32_t f2u(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
static inline float u2f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }

float orc_round_f16(float f)
{
    uint32_t x = f2u(f);
    uint32_t sign = x & 0x80000000u;
    uint32_t ax = x & 0x7fffffffu;
    if (ax >= 0x7f800000u) return f;                 /* inf / nan */
    if (ax >= 0x477ff000u) {                         /* >= 65520 -> inf */
        return u2f(sign | 0x7f800000u);
    }
    if (ax < 0x38800000u) {                          /* < 2^-14: fp16 subnormal grid 2^-24 */
        float a = u2f(ax);
        /* adding 2^-1 * 2^-24 * 2^24 trick: scale so that unit = 1, rint, scale back */
        float r = rintf(a * 16777216.0f) * (1.0f / 16777216.0f);
        return u2f(sign | f2u(r));
    }
    /* normal: keep 10 mantissa bits, RNE on the 13 dropped bits */
    uint32_t lsb = (ax >> 13) & 1u;
    ax += 0x0fffu + lsb;
    ax &= 0xffffe000u;
    return u2f(sign | ax);
}

float orc_round_bf16(float f)
{
    uint32_t x = f2u(f);
    if ((x & 0x7fffffffu) > 0x7f800000u) return f;   /* nan */
    uint32_t lsb = (x >> 16) & 1u;
    x += 0x7fffu + lsb;
    x &= 0xffff0000u;
    return u2f(x);
}

static inline float round_mid(float v, int mode)
{
    if (mode == 1) return orc_round_f16(v);
    if (mode == 2) return orc_round_bf16(v);
    return v;
}

void orc_round_array(float *x, long n, int mode)
{
    for (long i = 0; i < n; ++i) x[i] = round_mid(x[i], mode);
}

/* ------------------------------------------------------------------ */
/* Static activation quantizer                                          */
/*   fake_quant/quantizer/uniform.py:20-33 (quant)                      */
/*   fake_quant/quantizer/base.py:44-50    (x.float() first)            */
/*   q = clamp(round_half_even(x / s + zp), lo, hi)                      */
/* scale / zp are broadcast on the last dim when per_channel != 0        */
/* (base.py:20-24), otherwise they are scalars (layer_wise).            */
/* row_sel (may be NULL) picks scale set 0/1 per row: the MSQ extension  */
/* (two UniformQuantizers applied to masked rows).      */
/* ------------------------------------------------------------------ */
void orc_quant_static(const float *x, long rows, long cols,
                      const float *scale0, const float *zp0,
                      const float *scale1, const float *zp1,
                      const uint8_t *row_sel, int per_channel,
                      int lo, int hi, int8_t *q)
{
    for (long r = 0; r < rows; ++r) {
        const float *sc = (row_sel && row_sel[r]) ? scale1 : scale0;
        const float *zp = (row_sel && row_sel[r]) ? zp1 : zp0;
        for (long c = 0; c < cols; ++c) {
            float s = per_channel ? sc[c] : sc[0];
            float z = per_channel ? zp[c] : zp[0];
            float v = x[r * cols + c] / s + z;
            v = rintf(v);
            if (v < (float)lo) v = (float)lo;
            if (v > (float)hi) v = (float)hi;
            q[r * cols + c] = (int8_t)v;
        }
    }
}

/* ------------------------------------------------------------------ */
/* RMSN: weight-less RMS normalisation left behind by the LayerNorm fusion,
 * module_util.py:42-61:  h = fp32(x);  ms = sum(h*h) / mean_dim;  y = cast(h * rsqrt(ms + eps)).
 * torch leaves the summation order to its reduction kernels; this restatement fixes the order the
 * device kernel uses (mquant_amd/csrc/rmsn_quant.hip): 256 workers, worker t owns the 16-element
 * chunks c with c % 256 == t and adds its squares in ascending k; a 64-wide XOR butterfly
 * (strides 1,2,..,32) inside each group of 64 workers; the four group sums left to right.
 * rsqrt is 1/sqrt with both operations correctly rounded (= torch's CPU kernel).          */
/* ------------------------------------------------------------------ */
void orc_rmsn(const float *x, long rows, long cols, float mean_dim, float eps, int mode, float *y)
{
    for (long r = 0; r < rows; ++r) {
        const float *h = x + r * cols;
        float part[256];
        for (int t = 0; t < 256; ++t) part[t] = 0.0f;
        for (long k = 0; k < cols; ++k) {
            const int t = (int)((k / 16) % 256);
            float sq = h[k] * h[k];
            if (mode == 2) sq = round_mid(sq, 2);          /* bf16: x.pow(2) is a bf16 tensor */
            part[t] = part[t] + sq;
        }
        float wsum[4];
        for (int w = 0; w < 4; ++w) {
            float v[64], n[64];
            for (int l = 0; l < 64; ++l) v[l] = part[w * 64 + l];
            for (int st = 1; st < 64; st <<= 1) {
                for (int l = 0; l < 64; ++l) n[l] = v[l] + v[l ^ st];
                for (int l = 0; l < 64; ++l) v[l] = n[l];
            }
            wsum[w] = v[0];
        }
        const float total = ((wsum[0] + wsum[1]) + wsum[2]) + wsum[3];
        float inv;
        if (mode == 2) {                                   /* no promotion for bf16: one rounding per torch op */
            const float var = round_mid(round_mid(total, 2) / mean_dim, 2);
            const float ve = round_mid(var + eps, 2);
            inv = round_mid(1.0f / sqrtf(ve), 2);
        } else {
            const float ms = total / mean_dim;
            inv = 1.0f / sqrtf(ms + eps);
        }
        for (long k = 0; k < cols; ++k) y[r * cols + k] = round_mid(h[k] * inv, mode);
    }
}

/* Dynamic symmetric per-token quantizer, quant_utils.py:205-268 (act_per_tensor = False,
 * groupsize = -1) + sym_quant :46-50; fp32 arithmetic (the fp32 `tmp` of :239 promotes). */
void orc_quant_dyn(const float *x, long rows, long cols, int bits, float clip, int skip_col0,
                   float *scale, int8_t *q)
{
    const float maxq = (float)((1 << (bits - 1)) - 1);
    for (long r = 0; r < rows; ++r) {
        const float *h = x + r * cols;
        float mn = 0.0f, mx = 0.0f;
        for (long k = skip_col0 ? 1 : 0; k < cols; ++k) { if (h[k] < mn) mn = h[k]; if (h[k] > mx) mx = h[k]; }
        const float xmin = mn * clip, xmax0 = mx * clip;
        const float xmax = fmaxf(fabsf(xmin), xmax0);
        const float s = (xmax == 0.0f) ? 1.0f : xmax / maxq;
        scale[r] = s;
        for (long k = 0; k < cols; ++k) {
            float v = rintf(h[k] / s);
            if (v < -(maxq + 1.0f)) v = -(maxq + 1.0f);
            if (v > maxq) v = maxq;
            q[r * cols + k] = (skip_col0 && k == 0) ? 0 : (int8_t)v;
        }
    }
}

/* Dynamic symmetric GROUP-WISE quantizer (--a_groupsize g), quant_utils.py:181-203
 * (find_params_per_token_groupwise) + sym_quant :46-50.  Per row and group of g consecutive channels:
 *   xmax = amax * clip; xmin = amin * clip        (NO zero inclusion, unlike the per-token rule)
 *   xmax = max(|xmin|, xmax); scale = xmax / maxq, 1 where xmax == 0;   q = clamp(round(x / scale))
 * The reference keeps every one of these tensors in x's dtype (amax, the product with the Python float,
 * the quotient by the int64 maxq tensor, x / scale), torch evaluating each op in fp32 and rounding once:
 * mode 0 = fp32, 1 = fp16, 2 = bf16 inputs. */
void orc_quant_group(const float *x, long rows, long cols, long g, int bits, float clip, int mode,
                     float *scale, int8_t *q)
{
    const float maxq = (float)((1 << (bits - 1)) - 1);
    const long G = cols / g;
    for (long r = 0; r < rows; ++r)
        for (long gi = 0; gi < G; ++gi) {
            const float *h = x + r * cols + gi * g;
            float mn = h[0], mx = h[0];
            for (long k = 1; k < g; ++k) { if (h[k] < mn) mn = h[k]; if (h[k] > mx) mx = h[k]; }
            const float xmin = round_mid(mn * clip, mode), xmax0 = round_mid(mx * clip, mode);
            const float xmax = fmaxf(fabsf(xmin), xmax0);
            const float s = (xmax == 0.0f) ? 1.0f : round_mid(xmax / maxq, mode);
            scale[r * G + gi] = s;
            for (long k = 0; k < g; ++k) {
                float v = rintf(round_mid(h[k] / s, mode));
                if (v < -(maxq + 1.0f)) v = -(maxq + 1.0f);
                if (v > maxq) v = maxq;
                q[r * cols + gi * g + k] = (int8_t)v;
            }
        }
}

/* The same with ASYMMETRIC levels (sym = False branch of quant_utils.py:181-203 + asym_quant :27-31, maxq = 2^bits - 1): per
 * (row, group)  xmax = amax * clip, xmin = amin * clip (the range need not include 0); both 0 -> (-1, +1);
 * scale = (xmax - xmin) / maxq; zero = round(-xmin / scale); q = clamp(round(x / scale) + zero, 0, maxq) -- every operation
 * rounded to x's dtype (mode).  Stored for the int8 GEMM as q - 2^(bits-1); shift = scale * (2^(bits-1) - zero) in fp32. */
void orc_quant_group_asym(const float *x, long rows, long cols, long g, int bits, float clip, int mode,
                          float *scale, float *zero, float *shift, int8_t *q)
{
    const float maxq = (float)((1 << bits) - 1), half = (float)(1 << (bits - 1));
    const long G = cols / g;
    for (long r = 0; r < rows; ++r)
        for (long gi = 0; gi < G; ++gi) {
            const float *h = x + r * cols + gi * g;
            float mn = h[0], mx = h[0];
            for (long k = 1; k < g; ++k) { if (h[k] < mn) mn = h[k]; if (h[k] > mx) mx = h[k]; }
            float xmin = round_mid(mn * clip, mode), xmax = round_mid(mx * clip, mode);
            if (xmin == 0.0f && xmax == 0.0f) { xmin = -1.0f; xmax = 1.0f; }
            const float s = round_mid(round_mid(xmax - xmin, mode) / maxq, mode);
            const float z = rintf(round_mid(-xmin / s, mode));
            scale[r * G + gi] = s;
            zero[r * G + gi] = z;
            shift[r * G + gi] = s * (half - z);
            for (long k = 0; k < g; ++k) {
                float v = round_mid(rintf(round_mid(h[k] / s, mode)) + z, mode);
                if (v < 0.0f) v = 0.0f;
                if (v > maxq) v = maxq;
                q[r * cols + gi * g + k] = (int8_t)(v - half);
            }
        }
}

/* Dynamic ASYMMETRIC per-token quantizer (--a_asym), quant_utils.py:239-268 (else-branch) +
 * asym_quant :27-31, maxq = 2^bits - 1:
 *   xmin = min(min_k x, 0)*clip; xmax = max(max_k x, 0)*clip; both 0 -> (-1, +1)
 *   scale = (xmax - xmin) / maxq;  zero = round(-xmin / scale)
 *   q = clamp(round(x / scale) + zero, 0, maxq)           dequantised: scale * (q - zero)
 * Stored for the int8 GEMM as q - 2^(bits-1); shift[r] = scale * (2^(bits-1) - zero) is the per-row
 * factor of the rank-1 term that restores it: x_hat = scale * q_stored + shift. */
void orc_quant_dyn_asym(const float *x, long rows, long cols, int bits, float clip,
                        float *scale, float *zero, float *shift, int8_t *q)
{
    const float maxq = (float)((1 << bits) - 1), half = (float)(1 << (bits - 1));
    for (long r = 0; r < rows; ++r) {
        const float *h = x + r * cols;
        float mn = 0.0f, mx = 0.0f;
        for (long k = 0; k < cols; ++k) { if (h[k] < mn) mn = h[k]; if (h[k] > mx) mx = h[k]; }
        float xmin = mn * clip, xmax = mx * clip;
        if (xmin == 0.0f && xmax == 0.0f) { xmin = -1.0f; xmax = 1.0f; }
        const float s = (xmax - xmin) / maxq;
        const float z = rintf(-xmin / s);
        scale[r] = s;
        zero[r] = z;
        shift[r] = s * (half - z);
        for (long k = 0; k < cols; ++k) {
            float v = rintf(h[k] / s) + z;
            if (v < 0.0f) v = 0.0f;
            if (v > maxq) v = maxq;
            q[r * cols + k] = (int8_t)(v - half);
        }
    }
}

/* Dynamic PER-TENSOR quantizer (act_per_tensor = True), quant_utils.py:214-237: one range for the
 * whole tensor (column 0 left out and passed through under skip_col0, ActQuantWrapper.split :367-372):
 *   xmin = min(min x, 0)*clip; xmax = max(max x, 0)*clip
 *   sym:  xmax = max(|xmin|, xmax); scale = xmax / maxq (1 if xmax == 0), maxq = 2^(bits-1)-1
 *   asym: xmin == 0 -> -1; xmax == 0 -> +1 (each on its own, unlike the per-token rule);
 *         scale = (xmax - xmin) / maxq, zero = round(-xmin / scale), maxq = 2^bits - 1
 * Levels as in orc_quant_dyn / orc_quant_dyn_asym; params[0..2] = scale, zero, shift. */
void orc_quant_tensor(const float *x, long rows, long cols, int bits, float clip, int asym, int skip_col0, int mode,
                      float *params, int8_t *q)
{
    /* mode = dtype of x (0 fp32, 1 fp16, 2 bf16): the reference keeps the range, the scale, the zero point, x / scale and
     * the level sum in x's dtype here (torch.tensor(0).to(x), the int64 maxq tensor does not promote; quant_utils.py:214-231),
     * torch evaluating each op in fp32 and rounding once -- unlike the per-token rule, whose fp32 `tmp` promotes. */
    float mn = 0.0f, mx = 0.0f;
    for (long r = 0; r < rows; ++r)
        for (long k = skip_col0 ? 1 : 0; k < cols; ++k) {
            const float v = x[r * cols + k];
            if (v < mn) mn = v;
            if (v > mx) mx = v;
        }
    float xmin = round_mid(mn * clip, mode), xmax = round_mid(mx * clip, mode), s, z = 0.0f, lo, hi, half = 0.0f;
    if (asym) {
        const float maxq = (float)((1 << bits) - 1);
        if (xmin == 0.0f) xmin = -1.0f;
        if (xmax == 0.0f) xmax = 1.0f;
        s = round_mid(round_mid(xmax - xmin, mode) / maxq, mode);
        z = rintf(round_mid(-xmin / s, mode));
        lo = 0.0f; hi = maxq; half = (float)(1 << (bits - 1));
    } else {
        const float maxq = (float)((1 << (bits - 1)) - 1);
        xmax = fmaxf(fabsf(xmin), xmax);
        s = (xmax == 0.0f) ? 1.0f : round_mid(xmax / maxq, mode);
        lo = -(maxq + 1.0f); hi = maxq;
    }
    params[0] = s; params[1] = z; params[2] = s * (half - z);
    for (long r = 0; r < rows; ++r)
        for (long k = 0; k < cols; ++k) {
            float v = round_mid(rintf(round_mid(x[r * cols + k] / s, mode)) + z, mode);
            if (v < lo) v = lo;
            if (v > hi) v = hi;
            q[r * cols + k] = (skip_col0 && k == 0) ? 0 : (int8_t)(v - half);
        }
}

/* uniform.py:35-43: x_hat = (q - zp) * s  (fp32) */
void orc_dequant_static(const int8_t *q, long rows, long cols,
                        const float *scale0, const float *zp0,
                        const float *scale1, const float *zp1,
                        const uint8_t *row_sel, int per_channel, float *out)
{
    for (long r = 0; r < rows; ++r) {
        const float *sc = (row_sel && row_sel[r]) ? scale1 : scale0;
        const float *zp = (row_sel && row_sel[r]) ? zp1 : zp0;
        for (long c = 0; c < cols; ++c) {
            float s = per_channel ? sc[c] : sc[0];
            float z = per_channel ? zp[c] : zp[0];
            out[r * cols + c] = ((float)q[r * cols + c] - z) * s;
        }
    }
}

/* ------------------------------------------------------------------ */
/* Min/max observer reduction: fake_quant/observer/minmax.py:13-28.     */
/* Per-channel (last dim) min and max over all rows; the caller applies */
/* the zero-inclusion / running / layer_wise collapse rules.            */
/* ------------------------------------------------------------------ */
void orc_minmax_channels(const float *x, long rows, long cols,
                         float *mn, float *mx)
{
    for (long c = 0; c < cols; ++c) { mn[c] = INFINITY; mx[c] = -INFINITY; }
    for (long r = 0; r < rows; ++r)
        for (long c = 0; c < cols; ++c) {
            float v = x[r * cols + c];
            if (v < mn[c]) mn[c] = v;
            if (v > mx[c]) mx[c] = v;
        }
}

/* ------------------------------------------------------------------ */
/* Online Hadamard, CUDA-path semantics:                                */
/*   fake_quant/hadamard_utils.py:115-128 (matmul_hadU_cuda)            */
/*   fake_quant/utils.py:465-471          (zero pad n_in -> n)          */
/*   y = (H_K (x) H_{n/K}) x / sqrt(n),   index i = k*(n/K) + j          */
/* Steps, each a single fp32 rounding per operation:                    */
/*   1. per contiguous block of m = n/K: Walsh-Hadamard butterflies in  */
/*      ascending stride h = 1,2,4,..  (a+b, a-b) -- the operator of     */
/*      hadamard_utils.py:83-91 and of fast_hadamard_transform;         */
/*   2. multiply by scale = 1.0f / sqrtf((float)n)                       */
/*      (hadamard_utils.py:119,125: 1.0/torch.tensor(n).sqrt(), fp32);   */
/*   3. mid_round: the FHT extension returns x's dtype, so an fp16/bf16  */
/*      input is rounded here (mode 1/2); fp32_had keeps fp32 (mode 0);  */
/*   4. K > 1: out[j*m+i] = sum_k hadK[j][k] * y[k*m+i], k ascending,    */
/*      as one fp32 add/sub chain starting from 0 (hadamard_utils.py:127 */
/*      `hadK @ input`);                                                 */
/*   5. out_round: cast back to x_dtype (quant_utils.py:336-341).        */
/* post_div != 0 selects the pure-torch ordering of hadamard_utils.py:   */
/* 79-100 (matmul_hadU): no scale in step 2, `/ sqrt(n)` after step 4.  */
/* hadK is K*K int8 (+1/-1), row-major, ignored when K == 1.            */
/* ------------------------------------------------------------------ */
void orc_hadamard(const float *x, long rows, long n_in, long n, int K,
                  const int8_t *hadK, int mid_round_mode, int out_round_mode,
                  int post_div, float *out)
{
    const long m = n / K;
    const float root = sqrtf((float)n);
    const float scale = 1.0f / root;
    float *y = (float *)malloc(sizeof(float) * (size_t)n);
    for (long r = 0; r < rows; ++r) {
        for (long i = 0; i < n; ++i) y[i] = (i < n_in) ? x[r * n_in + i] : 0.0f;
        for (int k = 0; k < K; ++k) {
            float *b = y + (long)k * m;
            for (long h = 1; h < m; h <<= 1)
                for (long i = 0; i < m; i += 2 * h)
                    for (long j = i; j < i + h; ++j) {
                        float a0 = b[j], a1 = b[j + h];
                        b[j] = a0 + a1;
                        b[j + h] = a0 - a1;
                    }
        }
        if (!post_div)
            for (long i = 0; i < n; ++i) y[i] = round_mid(y[i] * scale, mid_round_mode);
        float *o = out + r * n;
        if (K == 1) {
            for (long i = 0; i < n; ++i)
                o[i] = round_mid(post_div ? y[i] / root : y[i], out_round_mode);
        } else {
            for (int j = 0; j < K; ++j)
                for (long i = 0; i < m; ++i) {
                    float acc = 0.0f;
                    for (int k = 0; k < K; ++k) {
                        float v = y[(long)k * m + i];
                        acc = (hadK[j * K + k] > 0) ? (acc + v) : (acc - v);
                    }
                    o[(long)j * m + i] = round_mid(post_div ? acc / root : acc, out_round_mode);
                }
        }
    }
    free(y);
}

/* ------------------------------------------------------------------ */
/* int4 wire format: fake_quant/quant_utils.py:61-94                    */
/* two's-complement nibbles, even index -> low nibble, odd -> high,     */
/* along the last dim.                                                  */
/* ------------------------------------------------------------------ */
void orc_pack_i4(const int8_t *q, long rows, long cols, uint8_t *out)
{
    for (long r = 0; r < rows; ++r)
        for (long c = 0; c < cols / 2; ++c) {
            uint8_t lo = (uint8_t)q[r * cols + 2 * c] & 0x0f;
            uint8_t hi = (uint8_t)q[r * cols + 2 * c + 1] & 0x0f;
            out[r * (cols / 2) + c] = (uint8_t)(lo | (hi << 4));
        }
}

void orc_unpack_i4(const uint8_t *p, long rows, long cols, int8_t *out)
{
    for (long r = 0; r < rows; ++r)
        for (long c = 0; c < cols / 2; ++c) {
            uint8_t b = p[r * (cols / 2) + c];
            int lo = b & 0x0f, hi = (b >> 4) & 0x0f;
            out[r * cols + 2 * c] = (int8_t)(lo >= 8 ? lo - 16 : lo);
            out[r * cols + 2 * c + 1] = (int8_t)(hi >= 8 ? hi - 16 : hi);
        }
}

/* ------------------------------------------------------------------ */
/* Integer core of the quantized Linear.                                */
/* The reference evaluates F.linear on dequantized tensors              */
/* (quant_utils.py:384 with uniform.py:42 and quant_utils.py:512-518);  */
/* on the integer grid that is acc[m][n] = sum_k qx[m][k] * qw[n][k].   */
/* ------------------------------------------------------------------ */
void orc_gemm_i8i4_i32(const int8_t *a, const int8_t *w, long M, long N, long K,
                       int32_t *acc)
{
#pragma omp parallel for schedule(static)
    for (long m = 0; m < M; ++m)
        for (long n = 0; n < N; ++n) {
            const int8_t *ar = a + m * K, *wr = w + n * K;
            int32_t s = 0;
            for (long k = 0; k < K; ++k) s += (int32_t)ar[k] * (int32_t)wr[k];
            acc[m * N + n] = s;
        }
}

/* ------------------------------------------------------------------ */
/* Dequant epilogue: y = acc * s_x[row set] * s_w[n] (+ bias) (+ rank-1 */
/* split term x0[m]*w0[n], quant_utils.py:367-376: channel 0 bypasses   */
/* the quantizer and goes through L1 in fp32).                           */
/* One rounding per operation, in this order.                            */
/* ------------------------------------------------------------------ */
void orc_epilogue(const int32_t *acc, long M, long N,
                  float sx0, float sx1, const uint8_t *row_sel,
                  const float *s_w, const float *bias,
                  const float *x0, const float *w0, float *out)
{
    for (long m = 0; m < M; ++m) {
        float sx = (row_sel && row_sel[m]) ? sx1 : sx0;
        for (long n = 0; n < N; ++n) {
            float t = (float)acc[m * N + n] * sx;
            t = t * s_w[n];
            if (bias) t = t + bias[n];
            if (x0) { float p = x0[m] * w0[n]; t = t + p; }
            out[m * N + n] = t;
        }
    }
}

/* ------------------------------------------------------------------ */
/* Group-wise WEIGHT scales (--w_groupsize g, gptq/gptq_utils.py:263-273: */
/* the GPTQ solver re-runs WeightQuantizer.find_params, quant_utils.py:   */
/* 446-509, on every group of g consecutive input channels, so W~[n][k] = */
/* s_w[n][k / g] * q[n][k]).  The integer path evaluates F.linear         */
/* (quant_utils.py:384) group by group:                                   */
/*   acc_g[m][n] = sum_{k in g} a[m][k] q[n][k]          exact int32      */
/*   f += (float(acc_g) * s_xg[m][g]) * s_wg[g][n]       ascending g, one */
/*                                                       fp32 rounding    */
/*                                                       per operation    */
/*   y  = f * s_x(m) + bias[n]                                            */
/* s_xg: group-wise activation scales of the same group size (NULL: the   */
/* factor is absent); s_x(m): sx_rows[m], or sx1 where row_sel[m] else    */
/* sx0.  acc_groups (optional, [M][G][N]) receives the per-group sums.    */
/* ------------------------------------------------------------------ */
void orc_gemm_wgroup(const int8_t *a, const int8_t *w, long M, long N, long K, long g,
                     const float *s_wg, const float *s_xg, float sx0, float sx1, const uint8_t *row_sel,
                     const float *sx_rows, const float *bias, int32_t *acc_groups, float *out)
{
    const long G = K / g;
#pragma omp parallel for schedule(static)
    for (long m = 0; m < M; ++m) {
        const float sx = sx_rows ? sx_rows[m] : ((row_sel && row_sel[m]) ? sx1 : sx0);
        for (long n = 0; n < N; ++n) {
            float f = 0.0f;
            for (long gi = 0; gi < G; ++gi) {
                int32_t acc = 0;
                const int8_t *ar = a + m * K + gi * g, *wr = w + n * K + gi * g;
                for (long k = 0; k < g; ++k) acc += (int32_t)ar[k] * (int32_t)wr[k];
                if (acc_groups) acc_groups[(m * G + gi) * N + n] = acc;
                float t = (float)acc;
                if (s_xg) t = t * s_xg[m * G + gi];
                t = t * s_wg[gi * N + n];
                f = f + t;
            }
            float y = f * sx;
            if (bias) y = y + bias[n];
            out[m * N + n] = y;
        }
    }
}

/* ------------------------------------------------------------------ */
/* Whole fake-quant Linear exactly as the reference evaluates it on CPU */
/* in fp32 (quant_utils.py:378-384): dequantize both operands, then an  */
/* fp32 matmul.  Used as the timed CPU baseline ("port") and as a        */
/* sanity cross-check of the integer path; summation is k-ascending.    */
/* ------------------------------------------------------------------ */
void orc_linear_fakequant_f32(const float *x, long M, long K,
                              float s_x, const float *w_dq, long N,
                              const float *bias, float *out)
{
    float *xq = (float *)malloc(sizeof(float) * (size_t)(M * K));
    for (long i = 0; i < M * K; ++i) {
        float v = rintf(x[i] / s_x + 0.0f);
        if (v < -128.0f) v = -128.0f;
        if (v > 127.0f) v = 127.0f;
        xq[i] = (v - 0.0f) * s_x;
    }
#pragma omp parallel for schedule(static)
    for (long m = 0; m < M; ++m)
        for (long n = 0; n < N; ++n) {
            const float *xr = xq + m * K, *wr = w_dq + n * K;
            float s = 0.0f;
            for (long k = 0; k < K; ++k) s += xr[k] * wr[k];
            out[m * N + n] = bias ? s + bias[n] : s;
        }
    free(xq);
}

/* ------------------------------------------------------------------ */
/* Symmetric per-output-channel weight quantizer (RTN):                 */
/*   fake_quant/quant_utils.py:446-518, sym branch.                     */
/*   xmax = max(|min(row,0)|, max(row,0)).clamp(1e-5); s = xmax/maxq    */
/*   optional MSE shrink search (:473-500): p = 1 - i/grid, i < 0.8*grid,*/
/*   err = sum |q - x|^norm; keep the best.                             */
/*   levels = clamp(round(x/s), -(maxq+1), maxq)  (quant_utils.py:42-45)*/
/* ------------------------------------------------------------------ */
/* d^norm for d >= 0, evaluated in double with explicitly ordered +,*,/ only (no libm
 * transcendental, no contraction), then rounded once to fp32.  The reference calls torch's
 * fp32 pow (quant_utils.py:490), whose last bit depends on the vector math library in use;
 * this form is within 1 ulp(fp32) of it and, being plain IEEE arithmetic, is reproduced bit for
 * bit by the device kernel (mquant_amd/csrc/wquant.hip carries its own copy). */
static inline double orc_log2_pos(double x)
{
    int e;
    double m = frexp(x, &e);                       /* x = m * 2^e, m in [0.5, 1) */
    if (m < 0.70710678118654752440) { m = m * 2.0; e -= 1; }
    const double f = (m - 1.0) / (m + 1.0);        /* |f| <= 0.1716 */
    const double f2 = f * f;
    double t = 1.0 / 23.0;                         /* atanh series: ln m = 2 f (1 + f^2/3 + ...) */
    t = t * f2 + 1.0 / 21.0;
    t = t * f2 + 1.0 / 19.0;
    t = t * f2 + 1.0 / 17.0;
    t = t * f2 + 1.0 / 15.0;
    t = t * f2 + 1.0 / 13.0;
    t = t * f2 + 1.0 / 11.0;
    t = t * f2 + 1.0 / 9.0;
    t = t * f2 + 1.0 / 7.0;
    t = t * f2 + 1.0 / 5.0;
    t = t * f2 + 1.0 / 3.0;
    t = t * f2 + 1.0;
    return (double)e + (2.0 * f) * t * 1.44269504088896340736;
}

static inline double orc_exp2(double y)
{
    const double yi = floor(y + 0.5);
    const double r = (y - yi) * 0.69314718055994530942;   /* |r| <= 0.3466 */
    double t = 1.0 / 6227020800.0;                 /* Taylor of e^r to r^13 */
    t = t * r + 1.0 / 479001600.0;
    t = t * r + 1.0 / 39916800.0;
    t = t * r + 1.0 / 3628800.0;
    t = t * r + 1.0 / 362880.0;
    t = t * r + 1.0 / 40320.0;
    t = t * r + 1.0 / 5040.0;
    t = t * r + 1.0 / 720.0;
    t = t * r + 1.0 / 120.0;
    t = t * r + 1.0 / 24.0;
    t = t * r + 1.0 / 6.0;
    t = t * r + 0.5;
    t = t * r + 1.0;
    t = t * r + 1.0;
    return ldexp(t, (int)yi);
}

float orc_pow_pos(float d, float norm)
{
    if (d == 0.0f) return 0.0f;
    return (float)orc_exp2((double)norm * orc_log2_pos((double)d));
}

/* Activations in front of a rotated Linear, as torch evaluates them on tensors of the given dtype
 * (mode 0 fp32, 1 fp16, 2 bf16): F.silu(gate) * up and QuickGELUActivation (x * sigmoid(1.702 x));
 * these live in the HF model code, not in MQuant.  exp through orc_exp2 (library-free). */
static inline float orc_exp_neg(float z) { return (float)orc_exp2((double)(-z) * 1.44269504088896340736); }

void orc_silu_mul(const float *g, const float *u, long n, int mode, float *out)
{
    for (long i = 0; i < n; ++i) {
        const float den = 1.0f + orc_exp_neg(g[i]);
        const float sl = round_mid(g[i] / den, mode);
        out[i] = round_mid(sl * u[i], mode);
    }
}

void orc_quick_gelu(const float *x, long n, int mode, float *out)
{
    for (long i = 0; i < n; ++i) {
        const float z = round_mid(1.702f * x[i], mode);
        const float den = 1.0f + orc_exp_neg(z);
        const float sg = round_mid(1.0f / den, mode);
        out[i] = round_mid(x[i] * sg, mode);
    }
}

void orc_wquant_sym(const float *w, long N, long K, int bits, int mse,
                    float norm, int grid, float maxshrink,
                    float *scale, int8_t *levels)
{
    const float maxq = (float)((1 << (bits - 1)) - 1);
    for (long n = 0; n < N; ++n) {
        const float *r = w + n * K;
        float mn = 0.0f, mx = 0.0f;
        for (long k = 0; k < K; ++k) { if (r[k] < mn) mn = r[k]; if (r[k] > mx) mx = r[k]; }
        float xmax = fmaxf(fabsf(mn), mx);
        if (xmax < 1e-5f) xmax = 1e-5f;
        float s = xmax / maxq;
        if (mse) {
            float best = INFINITY;
            int steps = (int)(maxshrink * (float)grid);
            for (int i = 0; i < steps; ++i) {
                /* python: p = 1 - i / grid in double, times an fp32 tensor -> fp32 */
                float p = (float)(1.0 - (double)i / (double)grid);
                float xmax1 = p * xmax;
                float s1 = xmax1 / maxq;
                float err = 0.0f;
                for (long k = 0; k < K; ++k) {
                    float q = rintf(r[k] / s1);
                    if (q < -(maxq + 1.0f)) q = -(maxq + 1.0f);
                    if (q > maxq) q = maxq;
                    float d = fabsf(s1 * q - r[k]);
                    err += orc_pow_pos(d, norm);
                }
                if (err < best) { best = err; s = s1; }
            }
        }
        scale[n] = s;
        if (levels)
            for (long k = 0; k < K; ++k) {
                float q = rintf(r[k] / s);
                if (q < -(maxq + 1.0f)) q = -(maxq + 1.0f);
                if (q > maxq) q = maxq;
                levels[n * K + k] = (int8_t)q;
            }
    }
}

/* Asymmetric per-channel weight quantizer (--w_asym), quant_utils.py:446-509 with sym = False:
 *   xmin = min(min_k w, 0); xmax = max(max_k w, 0); both 0 -> (-1, +1)
 *   scale = max(xmax - xmin, 1e-5) / maxq; zero = round(-xmin / scale); maxq = 2^bits - 1
 *   mse: for i < maxshrink*grid: p = 1 - i/grid; scale1 = (p xmax - p xmin) / maxq; zero1 = round(-p xmin / scale1);
 *        err = sum_k |scale1 (clamp(round(w/scale1) + zero1, 0, maxq) - zero1) - w|^norm; keep the first minimum
 * levels (optional): q - 2^(bits-1) with q = clamp(round(w / scale) + zero, 0, maxq) -- the form the int GEMM stores. */
void orc_wquant_asym(const float *w, long N, long K, int bits, int mse, float norm, int grid, float maxshrink,
                     float *scale, float *zero, int8_t *levels)
{
    const float maxq = (float)((1 << bits) - 1), half = (float)(1 << (bits - 1));
    for (long n = 0; n < N; ++n) {
        const float *r = w + n * K;
        float mn = 0.0f, mx = 0.0f;
        for (long k = 0; k < K; ++k) { if (r[k] < mn) mn = r[k]; if (r[k] > mx) mx = r[k]; }
        if (mn == 0.0f && mx == 0.0f) { mn = -1.0f; mx = 1.0f; }
        float d = mx - mn;
        if (d < 1e-5f) d = 1e-5f;
        float s = d / maxq;
        float z = rintf(-mn / s);
        if (mse) {
            float best = INFINITY;
            const int steps = (int)(maxshrink * (float)grid);
            for (int i = 0; i < steps; ++i) {
                const float p = (float)(1.0 - (double)i / (double)grid);
                const float mn1 = p * mn, mx1 = p * mx;
                const float s1 = (mx1 - mn1) / maxq;
                const float z1 = rintf(-mn1 / s1);
                float err = 0.0f;
                for (long k = 0; k < K; ++k) {
                    float q = rintf(r[k] / s1) + z1;
                    if (q < 0.0f) q = 0.0f;
                    if (q > maxq) q = maxq;
                    const float dq = s1 * (q - z1);
                    err += orc_pow_pos(fabsf(dq - r[k]), norm);
                }
                if (err < best) { best = err; s = s1; z = z1; }
            }
        }
        scale[n] = s;
        zero[n] = z;
        if (levels)
            for (long k = 0; k < K; ++k) {
                float q = rintf(r[k] / s) + z;
                if (q < 0.0f) q = 0.0f;
                if (q > maxq) q = maxq;
                levels[n * K + k] = (int8_t)(q - half);
            }
    }
}

/* GPTQ lazy-batch block, per-column loop -- reference fake_quant/gptq/gptq_utils.py:249-286 with
 * the symmetric per-channel quantizer without groups (quant_utils.py sym_quant_dequant: scale *
 * clamp(round(w / scale), -(maxq+1), maxq)):
 *     W1 = W[:, i1:i2].clone(); Hinv1 = Hinv[i1:i2, i1:i2]
 *     for i in range(count):
 *         w = W1[:, i]; d = Hinv1[i, i]
 *         q = quantize(w); Q1[:, i] = q
 *         err1 = (w - q) / d
 *         W1[:, i:] -= err1.unsqueeze(1).matmul(Hinv1[i, i:].unsqueeze(0))   (K = 1: one product)
 *         Err1[:, i] = err1
 * Rows never interact; every statement is one fp32 operation with one rounding (the file is
 * compiled with -ffp-contract=off).  W1 [N, ldw] is read only, H points at Hinv[i1, i1]. */
void orc_gptq_block(const float *W1, long N, long cols, long ldw,
                    const float *H, long ldh, const float *scale, int bits,
                    float *Q1, long ldq, float *E1, long lde)
{
    const float hi = (float)((1 << (bits - 1)) - 1), lo = -(hi + 1.0f);
    float *w = (float *)malloc(sizeof(float) * (size_t)(cols > 0 ? cols : 1));
    for (long n = 0; n < N; ++n) {
        const float s = scale[n];
        for (long j = 0; j < cols; ++j) w[j] = W1[n * ldw + j];
        for (long i = 0; i < cols; ++i) {
            const float *hrow = H + i * ldh;
            float lv = rintf(w[i] / s);
            if (lv < lo) lv = lo;
            if (lv > hi) lv = hi;
            const float q = s * lv;
            const float err = (w[i] - q) / hrow[i];
            for (long j = i + 1; j < cols; ++j) {
                const float prod = err * hrow[j];
                w[j] = w[j] - prod;
            }
            Q1[n * ldq + i] = q;
            E1[n * lde + i] = err;
        }
    }
    free(w);
}

/* ---- fp8 KV cache: NO reference implementation, parity unpinned -------------------
 * OCP e4m3fn: 1 sign, 4 exponent (bias 7), 3 mantissa bits; no infinities, 0x7F / 0xFF = NaN,
 * largest finite 448 (0x7E), smallest subnormal 2^-9.  Round to nearest even; finite values that
 * round beyond 448 saturate to 448 here (the callers clamp first, so this never decides anything);
 * NaN -> 0x7F.  The codec is pinned to torch.float8_e4m3fn in tests/test_oracle_golden.py. */
uint8_t orc_fp8_e4m3fn_encode(float x)
{
    if (x != x) return 0x7F;
    const uint8_t sign = signbit(x) ? 0x80 : 0x00;
    float a = fabsf(x);
    if (a >= 464.0f) return (uint8_t)(sign | 0x7E);           /* 464 = midpoint of 448 and the absent 480 */
    if (a < 0.015625f) {                                      /* below the smallest normal 2^-6: step 2^-9 */
        const float q = rintf(a * 512.0f);                    /* exact scaling, rint = round half to even */
        return (uint8_t)(sign | (uint8_t)q);                  /* q = 8 is the encoding of 2^-6 itself */
    }
    int e;
    const float m = frexpf(a, &e);                            /* a = m * 2^e, m in [0.5, 1) */
    float q = rintf(m * 16.0f);                               /* 4 significant bits: 8 .. 16 */
    int be = e - 1 + 7;                                       /* biased exponent of 1.xxx * 2^(e-1) */
    if (q == 16.0f) { q = 8.0f; be += 1; }
    if (be > 15 || (be == 15 && q > 14.0f)) return (uint8_t)(sign | 0x7E);
    return (uint8_t)(sign | (be << 3) | ((int)q - 8));
}

float orc_fp8_e4m3fn_decode(uint8_t b)
{
    const int e = (b >> 3) & 15, m = b & 7;
    float v;
    if (e == 15 && m == 7) return NAN;
    if (e == 0) v = ldexpf((float)m, -9);
    else v = ldexpf((float)(8 + m), e - 7 - 3);
    return (b & 0x80) ? -v : v;
}

/* write side: x [T][H][D] (already rounded to the cache's source dtype), scale [H] */
void orc_kv_quant_fp8(const float *x, long T, long H, long D, const float *scale, uint8_t *out)
{
    for (long t = 0; t < T; ++t)
        for (long h = 0; h < H; ++h)
            for (long d = 0; d < D; ++d) {
                float q = x[(t * H + h) * D + d] / scale[h];
                q = fminf(fmaxf(q, -448.0f), 448.0f);
                out[(t * H + h) * D + d] = orc_fp8_e4m3fn_encode(q);
            }
}

/* read side: fp32 product, the caller rounds to the output dtype */
void orc_kv_dequant_fp8(const uint8_t *q, long T, long H, long D, const float *scale, float *out)
{
    for (long t = 0; t < T; ++t)
        for (long h = 0; h < H; ++h)
            for (long d = 0; d < D; ++d)
                out[(t * H + h) * D + d] = orc_fp8_e4m3fn_decode(q[(t * H + h) * D + d]) * scale[h];
}
