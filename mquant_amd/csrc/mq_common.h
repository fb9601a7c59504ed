// mq_common.h -- shared host/device helpers for the gfx950 kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <hip/hip_bf16.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "../../include/mquant_hip.h"

namespace mq {

// ---- error plumbing -------------------------------------------------------
char *last_error_buf();
int fail(int code, const char *fmt, ...);
int check_launch(const char *what);
int ensure_dynamic_lds(const void *kernel, int bytes);   // per device, thread-safe (runtime.hip)
int device_cu_count();                                   // CUs of the current device, cached per device (runtime.hip); 256 on an MI355X

#define MQ_REQUIRE(cond, ...)                                   \
    do {                                                        \
        if (!(cond)) return ::mq::fail(MQ_EINVAL, __VA_ARGS__); \
    } while (0)

// ---- vector types -----------------------------------------------------------
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));
typedef int v2i __attribute__((ext_vector_type(2)));
typedef float v4f __attribute__((ext_vector_type(4)));
typedef unsigned short v8us __attribute__((ext_vector_type(8)));
typedef unsigned short v4us __attribute__((ext_vector_type(4)));

// ---- stores of kernel RESULTS ------------------------------------------------------------------
// What a kernel writes here is read next by ANOTHER launch (on all XCDs: through the Infinity Cache / HBM, never through the
// writer's L2).  A plain store leaves the lines dirty in the XCD's L2 and the launch ends with their write-back; the
// non-temporal form streams them out while the kernel still runs: 4-9 % of a small GEMM launch
// (profiles/r4_streaming_stores.txt).  For WHOLE lines only (the GEMM epilogues: 16 bytes per lane, rows of 128+ bytes): the
// 2-byte level stores of the Hadamard kernel became slower that way (partial lines go out one by one instead of merging
// in L2) and the quantizer did not change, so both keep plain stores.  MQ_PLAIN_STORES: A/B builds.
template <typename V>
__device__ __forceinline__ void store_out(V *ptr, V v)
{
#ifdef MQ_PLAIN_STORES
    *ptr = v;
#else
    __builtin_nontemporal_store(v, ptr);
#endif
}

// ---- dtype conversion (round-to-nearest-even everywhere) --------------------
__device__ __forceinline__ float f16_bits_to_f32(unsigned short h)
{
    return __half2float(__ushort_as_half(h));
}
__device__ __forceinline__ unsigned short f32_to_f16_bits(float f)
{
    // The empty asm makes `f` opaque: without it the backend folds a preceding fp32 multiply
    // into V_FMA_MIXLO_F16 (ONE rounding, product -> f16), whereas the reference rounds the
    // product to fp32 first and then casts (two roundings); they differ on ~0.1 % of values.
    asm volatile("" : "+v"(f));
    return __half_as_ushort(__float2half_rn(f));
}
__device__ __forceinline__ float bf16_bits_to_f32(unsigned short h)
{
    return __uint_as_float(((unsigned)h) << 16);
}
__device__ __forceinline__ unsigned short f32_to_bf16_bits(float f)
{
    unsigned x = __float_as_uint(f);
    if ((x & 0x7fffffffu) > 0x7f800000u) return (unsigned short)((x >> 16) | 0x40);
    unsigned lsb = (x >> 16) & 1u;
    x += 0x7fffu + lsb;
    return (unsigned short)(x >> 16);
}

// two fp32 -> two half-precision numbers in one dword (low half = a), round-to-nearest-even: V_CVT_PK_F16_F32 /
// V_CVT_PK_BF16_F32 (gfx950).  The empty asm keeps the pair opaque (see f32_to_f16_bits: no fused multiply-convert).
typedef float v2f_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned pack2_f16(float a, float b)
{
    typedef _Float16 v2h __attribute__((ext_vector_type(2)));
    v2f_t t = {a, b};
    asm volatile("" : "+v"(t));
    const v2h h = __builtin_convertvector(t, v2h);
    return *reinterpret_cast<const unsigned *>(&h);
}
__device__ __forceinline__ unsigned pack2_bf16(float a, float b)
{
    typedef __bf16 v2b __attribute__((ext_vector_type(2)));
    v2f_t t = {a, b};
    asm volatile("" : "+v"(t));
    const v2b h = __builtin_convertvector(t, v2b);
    return *reinterpret_cast<const unsigned *>(&h);
}

// Kernel arguments: warm the scalar cache in ONE round trip.  hipcc loads a by-value argument block lazily, field by field next to
// the first use, with an s_waitcnt in front of every dependent step -- three to six SERIALIZED scalar-load round trips at the top of
// the quantizer, Hadamard and GEMM kernels (the argument block of a launch is never in the scalar cache: ~1 us between a workgroup's
// entry and its first useful instruction, profiles/r5_ws_fixed_cost_timeline.txt).  One dword per 64-byte line of the kernarg segment,
// all requested at once and waited for once: the compiler's own loads behind it hit the cache.  BYTES = sizeof(argument struct).
template <int ARG_BYTES, bool READS_GRID = false>
__device__ __forceinline__ void kernarg_warm()
{
#ifndef MQ_LAZY_ARGS
    // One dword per 64-byte line of the argument block; no load leaves it (the last line's dword is the block's last one, whatever
    // follows in the segment), outputs early-clobber so that none is placed on the base pair.  READS_GRID: the kernel reads
    // gridDim, so the three implicit block counts (12 bytes at the next 8-byte boundary) exist behind the struct and are warmed too.
    static_assert(ARG_BYTES >= 4 && ARG_BYTES % 4 == 0, "argument block: whole dwords");
    constexpr int BYTES = READS_GRID ? (ARG_BYTES + 7) / 8 * 8 + 12 : ARG_BYTES;
    constexpr int LINES = (BYTES + 63) / 64 > 8 ? 8 : (BYTES + 63) / 64;
#define MQ_KA_OFF(i) ((i) * 64 + 4 <= BYTES ? (i) * 64 : BYTES - 4)
    const unsigned long long ka = (unsigned long long)__builtin_amdgcn_kernarg_segment_ptr();
    unsigned d0, d1, d2, d3, d4, d5, d6, d7;
    (void)d0; (void)d1; (void)d2; (void)d3; (void)d4; (void)d5; (void)d6; (void)d7;
    if constexpr (LINES == 1) asm volatile("s_load_dword %0, %1, %2\n\ts_waitcnt lgkmcnt(0)" : "=&s"(d0) : "s"(ka), "n"(MQ_KA_OFF(0)));
    if constexpr (LINES == 2) asm volatile("s_load_dword %0, %2, %3\n\ts_load_dword %1, %2, %4\n\ts_waitcnt lgkmcnt(0)" : "=&s"(d0), "=&s"(d1) : "s"(ka), "n"(MQ_KA_OFF(0)), "n"(MQ_KA_OFF(1)));
    if constexpr (LINES == 3) asm volatile("s_load_dword %0, %3, %4\n\ts_load_dword %1, %3, %5\n\ts_load_dword %2, %3, %6\n\ts_waitcnt lgkmcnt(0)" : "=&s"(d0), "=&s"(d1), "=&s"(d2) : "s"(ka), "n"(MQ_KA_OFF(0)), "n"(MQ_KA_OFF(1)), "n"(MQ_KA_OFF(2)));
    if constexpr (LINES == 4) asm volatile("s_load_dword %0, %4, %5\n\ts_load_dword %1, %4, %6\n\ts_load_dword %2, %4, %7\n\ts_load_dword %3, %4, %8\n\ts_waitcnt lgkmcnt(0)" : "=&s"(d0), "=&s"(d1), "=&s"(d2), "=&s"(d3) : "s"(ka), "n"(MQ_KA_OFF(0)), "n"(MQ_KA_OFF(1)), "n"(MQ_KA_OFF(2)), "n"(MQ_KA_OFF(3)));
    if constexpr (LINES == 5) asm volatile("s_load_dword %0, %5, %6\n\ts_load_dword %1, %5, %7\n\ts_load_dword %2, %5, %8\n\ts_load_dword %3, %5, %9\n\ts_load_dword %4, %5, %10\n\ts_waitcnt lgkmcnt(0)" : "=&s"(d0), "=&s"(d1), "=&s"(d2), "=&s"(d3), "=&s"(d4) : "s"(ka), "n"(MQ_KA_OFF(0)), "n"(MQ_KA_OFF(1)), "n"(MQ_KA_OFF(2)), "n"(MQ_KA_OFF(3)), "n"(MQ_KA_OFF(4)));
    if constexpr (LINES == 6) asm volatile("s_load_dword %0, %6, %7\n\ts_load_dword %1, %6, %8\n\ts_load_dword %2, %6, %9\n\ts_load_dword %3, %6, %10\n\ts_load_dword %4, %6, %11\n\ts_load_dword %5, %6, %12\n\ts_waitcnt lgkmcnt(0)" : "=&s"(d0), "=&s"(d1), "=&s"(d2), "=&s"(d3), "=&s"(d4), "=&s"(d5) : "s"(ka), "n"(MQ_KA_OFF(0)), "n"(MQ_KA_OFF(1)), "n"(MQ_KA_OFF(2)), "n"(MQ_KA_OFF(3)), "n"(MQ_KA_OFF(4)), "n"(MQ_KA_OFF(5)));
    if constexpr (LINES == 7) asm volatile("s_load_dword %0, %7, %8\n\ts_load_dword %1, %7, %9\n\ts_load_dword %2, %7, %10\n\ts_load_dword %3, %7, %11\n\ts_load_dword %4, %7, %12\n\ts_load_dword %5, %7, %13\n\ts_load_dword %6, %7, %14\n\ts_waitcnt lgkmcnt(0)" : "=&s"(d0), "=&s"(d1), "=&s"(d2), "=&s"(d3), "=&s"(d4), "=&s"(d5), "=&s"(d6) : "s"(ka), "n"(MQ_KA_OFF(0)), "n"(MQ_KA_OFF(1)), "n"(MQ_KA_OFF(2)), "n"(MQ_KA_OFF(3)), "n"(MQ_KA_OFF(4)), "n"(MQ_KA_OFF(5)), "n"(MQ_KA_OFF(6)));
    if constexpr (LINES == 8) asm volatile("s_load_dword %0, %8, %9\n\ts_load_dword %1, %8, %10\n\ts_load_dword %2, %8, %11\n\ts_load_dword %3, %8, %12\n\ts_load_dword %4, %8, %13\n\ts_load_dword %5, %8, %14\n\ts_load_dword %6, %8, %15\n\ts_load_dword %7, %8, %16\n\ts_waitcnt lgkmcnt(0)" : "=&s"(d0), "=&s"(d1), "=&s"(d2), "=&s"(d3), "=&s"(d4), "=&s"(d5), "=&s"(d6), "=&s"(d7) : "s"(ka), "n"(MQ_KA_OFF(0)), "n"(MQ_KA_OFF(1)), "n"(MQ_KA_OFF(2)), "n"(MQ_KA_OFF(3)), "n"(MQ_KA_OFF(4)), "n"(MQ_KA_OFF(5)), "n"(MQ_KA_OFF(6)), "n"(MQ_KA_OFF(7)));
#undef MQ_KA_OFF
#endif
}

template <int DT> struct Elem;
template <> struct Elem<MQ_F16> {
    typedef unsigned short T;
    static __device__ __forceinline__ float ld(T v) { return f16_bits_to_f32(v); }
    static __device__ __forceinline__ T st(float f) { return f32_to_f16_bits(f); }
    static __device__ __forceinline__ float rnd(float f) { return ld(st(f)); }
};
template <> struct Elem<MQ_BF16> {
    typedef unsigned short T;
    static __device__ __forceinline__ float ld(T v) { return bf16_bits_to_f32(v); }
    static __device__ __forceinline__ T st(float f) { return f32_to_bf16_bits(f); }
    static __device__ __forceinline__ float rnd(float f) { return ld(st(f)); }
};
template <> struct Elem<MQ_F32> {
    typedef float T;
    static __device__ __forceinline__ float ld(T v) { return v; }
    static __device__ __forceinline__ T st(float f) { return f; }
    static __device__ __forceinline__ float rnd(float f) { return f; }
};

// q = clamp(rint(x / s), lo, hi): IEEE-correct division (hipcc default:
// -fhip-fp32-correctly-rounded-divide-sqrt), v_rndne_f32 for round-half-even.
__device__ __forceinline__ int quant_level(float x, float s, float lo, float hi)
{
    float v = rintf(x / s);
    v = fmaxf(v, lo);   // NaN-free inputs assumed; fmaxf(NaN, lo) = lo like the clamp of a NaN-free oracle
    v = fminf(v, hi);
    return (int)v;
}

// The same levels without the ~10-instruction IEEE division on (almost) every element.  t = x * fl(1/s) is
// within 2^-23 |x/s| of the quotient and fl(x/s) within 2^-24 |x/s|: for |x/s| <= 256 the two lie less than
// 4.6e-5 apart, so rint() of either is the same integer unless t sits within that distance of a half-integer
// -- then, and only then (about 2 elements in 10^4), the exact quotient is computed; beyond 256 both saturate
// (|lo|, |hi| <= 255).  Bit-identical to quant_level by construction; the caller passes inv_s = 1.0f / s
// (IEEE) and uses this form only when s and inv_s are normal numbers (quant_rcp_ok).
__host__ __device__ inline bool quant_rcp_ok(float s)
{
    const float r = 1.0f / s;
    return s >= 1.17549435e-38f && r >= 1.17549435e-38f && r <= 3.0e38f;
}

// N levels at once: ONE (rarely taken) branch instead of N.  rcp = false: plain quant_level.
template <int N>
__device__ __forceinline__ void quant_levels(const float (&x)[N], float s, float inv_s, bool rcp, float lo, float hi, int (&q)[N])
{
    float t[N];
    if (rcp) {
        bool risky = false;
#pragma unroll
        for (int e = 0; e < N; ++e) {
            t[e] = x[e] * inv_s;
            risky |= fabsf(__builtin_amdgcn_fractf(t[e]) - 0.5f) < 1.0e-4f;
        }
        if (risky) {
#pragma unroll
            for (int e = 0; e < N; ++e) t[e] = x[e] / s;
        }
    } else {
#pragma unroll
        for (int e = 0; e < N; ++e) t[e] = x[e] / s;
    }
#pragma unroll
    for (int e = 0; e < N; ++e) {
        float v = rintf(t[e]);
        v = fmaxf(v, lo);
        v = fminf(v, hi);
        q[e] = (int)v;
    }
}

// N levels (N % 4 == 0) in [-128, 127], packed four per dword (element e in byte e & 3 of word e >> 2): the same levels as
// quant_levels by construction, in fewer instructions --
//  * the exact quotient is taken only where it can matter: t = x * fl(1/s) is within 1.5 * 2^-23 |t| of fl(x / s), so the two
//    round to different integers only if t lies that close to a half-integer; the test |frac(t) - 0.5| < 2^-21 |t| (one fma
//    per element, a running minimum) sends about 1 wave in 40 through the IEEE divisions where the absolute 1e-4 of
//    quant_levels sent 1 in 5 (a wave takes the branch if ANY of its 64 x N elements does);
//  * clamp first, then add 1.5 * 2^23: the sum is rounded to an integer, ties to even, exactly like rint (the bounds are
//    integers, so clamp and rint commute), and its low mantissa byte IS the two's complement level: no v_rndne, no
//    v_cvt_i32, and three v_perm per four levels instead of masks, shifts and ors.
template <int N>
__device__ __forceinline__ void quant_levels_i8_packed(const float (&x)[N], float s, float inv_s, bool rcp, unsigned (&w)[N / 4])
{
    float t[N];
    if (rcp) {
        float worst = 1.0f;
#pragma unroll
        for (int e = 0; e < N; ++e) {
            t[e] = x[e] * inv_s;
            const float g = __builtin_amdgcn_fractf(t[e]) - 0.5f;
            worst = fminf(worst, __builtin_fmaf(fabsf(t[e]), -4.76837158203125e-7f, fabsf(g)));
        }
        if (worst < 0.0f) {
#pragma unroll
            for (int e = 0; e < N; ++e) t[e] = x[e] / s;
        }
    } else {
#pragma unroll
        for (int e = 0; e < N; ++e) t[e] = x[e] / s;
    }
    unsigned u[N];
#pragma unroll
    for (int e = 0; e < N; ++e) {
        float v = fmaxf(t[e], -128.0f);
        v = fminf(v, 127.0f);
        u[e] = __float_as_uint(v + 12582912.0f);
    }
#pragma unroll
    for (int j = 0; j < N / 4; ++j) {
        const unsigned lo = __builtin_amdgcn_perm(u[4 * j + 1], u[4 * j], 0x0c0c0400u);
        const unsigned hi = __builtin_amdgcn_perm(u[4 * j + 3], u[4 * j + 2], 0x0c0c0400u);
        w[j] = __builtin_amdgcn_perm(hi, lo, 0x05040100u);
    }
}

// ---- library-free transcendental pieces --------------------------------------------------------
// log2 / exp2 in double from ordered +,*,/ only (no libm, no contraction): every caller rounds the
// result once to fp32, which makes device results reproducible bit for bit by the C oracle
// (orc_log2_pos / orc_exp2) and <= 1 ulp(fp32) from the correctly rounded value.
__device__ __forceinline__ double log2_pos(double x)
{
    int e;
    double m = frexp(x, &e);
    if (m < 0.70710678118654752440) { m = m * 2.0; e -= 1; }
    const double f = (m - 1.0) / (m + 1.0);
    const double f2 = f * f;
    double t = 1.0 / 23.0;
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

__device__ __forceinline__ double exp2_d(double y)
{
    const double yi = floor(y + 0.5);
    const double r = (y - yi) * 0.69314718055994530942;
    double t = 1.0 / 6227020800.0;
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

// exp(-z) in fp32 for the fused activations: the device math library's expf, i.e. the very
// function torch's silu / sigmoid kernels call on this GPU (the fused kernels are checked bit for
// bit against the torch ops they replace).  The library-free double form above costs ~40 fp64
// operations per element (+34 us on the down_proj call) and is kept for the offline quantizer only.
__device__ __forceinline__ float exp_neg(float z)
{
    return expf(-z);
}

// Activations that sit in front of a rotated Linear, evaluated the way torch evaluates them on
// DT tensors: fp32 arithmetic inside one op, one rounding to DT per op.
enum { MQ_ACT_NONE = 0, MQ_ACT_SILU_MUL = 1, MQ_ACT_QUICK_GELU = 2 };

// ---- the reference forms: the device library's expf and IEEE divisions (what torch's kernels execute) ----
template <int DT> __device__ __forceinline__ float act_silu_ref(float g)
{
    const float den = 1.0f + exp_neg(g);              // F.silu: x / (1 + exp(-x))
    return Elem<DT>::rnd(g / den);
}
template <int DT> __device__ __forceinline__ float act_sigmoid_ref(float z)
{
    const float den = 1.0f + exp_neg(z);
    return Elem<DT>::rnd(1.0f / den);
}

// ---- 16-bit dtypes: the same VALUES in half the instructions (round 6) ----
// silu and sigmoid of a half-precision tensor are functions of 65 536 inputs, so ANY arithmetic that returns the reference's
// rounded result on every one of them IS the reference: tests/test_gpu_act_exhaustive.py runs both forms over all 2^16 bit
// patterns of fp16 and bf16 (mq_debug_act_table) and requires identical bits (NaNs: NaN for NaN).  What is cheaper here:
//  * exp(-x): the library's own reduction (x log2(e) as a 49-bit product, 2^fraction by V_EXP_F32, V_LDEXP_F32) without its
//    range selects -- the argument is clamped to +-128 instead (one V_MED3_F32): the sum 1 + exp(-x) and the quotients below come
//    out the same where the library would have returned 0 or inf;
//  * x / den and 1 / den: V_RCP_F32 and one Newton step (the quotient's error stays far inside the half-precision rounding
//    interval it lands in, on every input), V_DIV_FIXUP_F32 for den = inf -- 5 instructions for the 10 of the IEEE sequence.
// ~14 vector-ALU instructions per element instead of ~25; fp32 tensors keep the reference forms.
template <int DT> __device__ __forceinline__ float exp_neg_16(float v)
{
    // (+-inf and bf16's values beyond fp32's exponent range: the reduction below would form inf - inf; 2^-128 and 2^128 give the same
    //  sums 1 + exp(-x) and quotients as the library's 0 and inf)
    const float x = __builtin_amdgcn_fmed3f(-v, -128.0f, 128.0f);
    const float c = 0x1.715476p+0f, cc = 0x1.4ae0bep-26f;
    const float ph = x * c;
    const float pl = __builtin_fmaf(x, cc, __builtin_fmaf(x, c, -ph));
    const float e = __builtin_rintf(ph);
    const float a = (ph - e) + pl;
    return __builtin_ldexpf(__builtin_amdgcn_exp2f(a), (int)e);
}
template <int DT> __device__ __forceinline__ float silu_raw_16(float g)        // g: a DT value; the fp32 quotient BEFORE its rounding to DT
{
    const float den = 1.0f + exp_neg_16<DT>(g);
    const float r = __builtin_amdgcn_rcpf(den);
    float q = g * r;
    q = __builtin_fmaf(__builtin_fmaf(-q, den, g), r, q);
    q = __builtin_amdgcn_div_fixupf(q, den, g);      // den = inf, g = +-inf, NaN: the division's own special cases
    // V_RCP_F32 flushes a denormal reciprocal to 0 (den > 2^126, g below -87): fp16 rounds those quotients to -0 anyway, bf16 has the
    // exponent range to represent them -- there (never, on real activations) the IEEE sequence runs
    if (DT == MQ_BF16 && __builtin_expect(den > 0x1p+125f, 0)) q = g / den;
    return q;
}
template <int DT> __device__ __forceinline__ float sigmoid_raw_16(float z)
{
    const float den = 1.0f + exp_neg_16<DT>(z);
    float r = __builtin_amdgcn_rcpf(den);
    r = __builtin_fmaf(__builtin_fmaf(-den, r, 1.0f), r, r);
    r = __builtin_amdgcn_div_fixupf(r, den, 1.0f);
    if (DT == MQ_BF16 && __builtin_expect(den > 0x1p+125f, 0)) r = 1.0f / den;
    return (z != z) ? z : r;                          // (the clamp in exp_neg_16 swallows a NaN argument; silu's numerator carries it)
}
template <int DT> __device__ __forceinline__ float act_silu_16(float g) { return Elem<DT>::rnd(silu_raw_16<DT>(g)); }
template <int DT> __device__ __forceinline__ float act_sigmoid_16(float z) { return Elem<DT>::rnd(sigmoid_raw_16<DT>(z)); }
template <int DT> __device__ __forceinline__ float act_silu(float g)
{
    if constexpr (DT == MQ_F32) return act_silu_ref<DT>(g);
    else return act_silu_16<DT>(g);
}
template <int DT> __device__ __forceinline__ float act_sigmoid(float z)
{
    if constexpr (DT == MQ_F32) return act_sigmoid_ref<DT>(z);
    else return act_sigmoid_16<DT>(z);
}

template <int DT> __device__ __forceinline__ float act_silu_mul(float g, float u)
{
    return Elem<DT>::rnd(act_silu<DT>(g) * u);          // silu(gate) * up
}

template <int DT> __device__ __forceinline__ float act_quick_gelu(float x)
{
    const float z = Elem<DT>::rnd(1.702f * x);         // QuickGELUActivation: x * sigmoid(1.702 x)
    return Elem<DT>::rnd(x * act_sigmoid<DT>(z));
}

// Two outputs at once from the fp32 values of two Linear outputs each (the producer GEMM's act epilogues): the roundings of the
// scalar forms above with the packed converts (V_CVT_PK_F16_F32 / V_CVT_PK_BF16_F32) and, for fp16, V_PK_MUL_F16 for the last product
// -- a product of two halves is exact in fp32, so its rounding to half IS the correctly rounded half product.  Returns the two
// results packed in one dword (low half = first); fp32: two floats through `out`.
template <int DT> __device__ __forceinline__ unsigned act_silu_mul_pk(float g0, float g1, float u0, float u1)
{
    static_assert(DT == MQ_F16 || DT == MQ_BF16, "packed forms are for the 16-bit dtypes");
    if constexpr (DT == MQ_F16) {
        typedef _Float16 h2 __attribute__((ext_vector_type(2)));
        const unsigned gp = pack2_f16(g0, g1), up = pack2_f16(u0, u1);        // the two Linear outputs, rounded to the model's dtype
        h2 gh, uh;
        __builtin_memcpy(&gh, &gp, 4);
        __builtin_memcpy(&uh, &up, 4);
        const unsigned sp = pack2_f16(silu_raw_16<DT>((float)gh[0]), silu_raw_16<DT>((float)gh[1]));
        h2 sh;
        __builtin_memcpy(&sh, &sp, 4);
        const h2 r = sh * uh;
        unsigned out;
        __builtin_memcpy(&out, &r, 4);
        return out;
    } else {
        const unsigned gp = pack2_bf16(g0, g1), up = pack2_bf16(u0, u1);
        const float ga = __uint_as_float(gp << 16), gb = __uint_as_float(gp & 0xffff0000u);
        const float ua = __uint_as_float(up << 16), ub = __uint_as_float(up & 0xffff0000u);
        return pack2_bf16(act_silu_16<DT>(ga) * ua, act_silu_16<DT>(gb) * ub);
    }
}
template <int DT> __device__ __forceinline__ unsigned act_quick_gelu_pk(float x0, float x1)
{
    static_assert(DT == MQ_F16 || DT == MQ_BF16, "packed forms are for the 16-bit dtypes");
    if constexpr (DT == MQ_F16) {
        typedef _Float16 h2 __attribute__((ext_vector_type(2)));
        const unsigned xp = pack2_f16(x0, x1);
        h2 xh;
        __builtin_memcpy(&xh, &xp, 4);
        const float xa = (float)xh[0], xb = (float)xh[1];
        const unsigned zp = pack2_f16(1.702f * xa, 1.702f * xb);
        h2 zh;
        __builtin_memcpy(&zh, &zp, 4);
        const unsigned sp = pack2_f16(sigmoid_raw_16<DT>((float)zh[0]), sigmoid_raw_16<DT>((float)zh[1]));
        h2 sh;
        __builtin_memcpy(&sh, &sp, 4);
        const h2 r = xh * sh;
        unsigned out;
        __builtin_memcpy(&out, &r, 4);
        return out;
    } else {
        const unsigned xp = pack2_bf16(x0, x1);
        const float xa = __uint_as_float(xp << 16), xb = __uint_as_float(xp & 0xffff0000u);
        const unsigned zp = pack2_bf16(1.702f * xa, 1.702f * xb);
        const float za = __uint_as_float(zp << 16), zb = __uint_as_float(zp & 0xffff0000u);
        return pack2_bf16(xa * act_sigmoid_16<DT>(za), xb * act_sigmoid_16<DT>(zb));
    }
}

__host__ __device__ inline long ceil_div(long a, long b) { return (a + b - 1) / b; }

// Byte offset of element (row, col) of the int8 activation matrix handed from the quantizers to
// the GEMM: row-major with leading dimension ld, or the TILED layout (ld == MQ_LD_TILED,
// include/mquant_hip.h): [row/16][col/64] pieces of 1 KiB, 16-byte chunk ((col/16)%4, row%16) inside.
// A 4-, 8- or 16-byte group that starts at a multiple of its size stays inside one chunk.
__host__ __device__ inline long act_offset(long row, long col, long K_pad, long ld)
{
    if (ld != MQ_LD_TILED) return row * ld + col;
    return ((row >> 4) * (K_pad >> 6) + (col >> 6)) * 1024 + ((((col >> 4) & 3) << 4) + (row & 15)) * 16 + (col & 15);
}

// Rows -> workgroups for kernels that own whole rows and write the tiled layout: the 16 rows of a
// piece row go to workgroups on ONE XCD (block b runs on XCD b % 8, gridDim % 8 == 0), so the eight
// 16-byte chunks that share a 128-byte line meet in one L2 instead of being written back by eight.
// v = virtual row index (block + iteration * grid); returns the row (may be >= M: skip it).
__device__ __forceinline__ long tiled_row_of(long v)
{
    const long xcd = v & 7, i = v >> 3;
    return (((i >> 4) << 3) + xcd) * 16 + (i & 15);
}

}  // namespace mq
