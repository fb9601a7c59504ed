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

#define MQ_REQUIRE(cond, ...)                                   \
    do {                                                        \
        if (!(cond)) return ::mq::fail(MQ_EINVAL, __VA_ARGS__); \
    } while (0)

// ---- vector types -----------------------------------------------------------
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v2i __attribute__((ext_vector_type(2)));
typedef float v4f __attribute__((ext_vector_type(4)));
typedef unsigned short v8us __attribute__((ext_vector_type(8)));
typedef unsigned short v4us __attribute__((ext_vector_type(4)));

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

__host__ __device__ inline long ceil_div(long a, long b) { return (a + b - 1) / b; }

}  // namespace mq
