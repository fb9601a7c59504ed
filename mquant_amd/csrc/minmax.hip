// minmax.hip -- observer reductions (calibration only), HBM-bound: one read of x.
// Reference: fake_quant/observer/minmax.py:13-28 after observer/base.py:15-28.
//
// Per-channel: thread t owns 8 adjacent channels (one 16-byte load per row for half
// inputs), blockIdx.y slices the rows; slices are merged with order-independent integer
// atomics on the fp32 bit patterns, so the result does not depend on scheduling.
#include "mq_common.h"

namespace mq {

__device__ __forceinline__ void atomic_min_f32(float *addr, float v)
{
    v = v + 0.0f;  // -0.0 -> +0.0 so the sign test below orders zeros consistently
    if (v >= 0.0f) atomicMin(reinterpret_cast<int *>(addr), __float_as_int(v));
    else atomicMax(reinterpret_cast<unsigned *>(addr), __float_as_uint(v));
}
__device__ __forceinline__ void atomic_max_f32(float *addr, float v)
{
    v = v + 0.0f;
    if (v >= 0.0f) atomicMax(reinterpret_cast<int *>(addr), __float_as_int(v));
    else atomicMin(reinterpret_cast<unsigned *>(addr), __float_as_uint(v));
}

__global__ void minmax_init_kernel(float *mn, float *mx, long n)
{
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        mn[i] = __int_as_float(0x7f800000);
        mx[i] = __int_as_float(0xff800000);
    }
}

template <int DT>
__global__ __launch_bounds__(256) void minmax_channels_kernel(
    const typename Elem<DT>::T *__restrict__ x, long M, long C, long ldx, long col_begin,
    long rows_per_slice, float *__restrict__ mn, float *__restrict__ mx, int vec_ok)
{
    typedef typename Elem<DT>::T T;
    const long c0 = col_begin + ((long)blockIdx.x * blockDim.x + threadIdx.x) * 8;
    if (c0 >= C) return;
    const long r0 = (long)blockIdx.y * rows_per_slice;
    long r1 = r0 + rows_per_slice;
    if (r1 > M) r1 = M;
    float lo[8], hi[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        lo[i] = __int_as_float(0x7f800000);
        hi[i] = __int_as_float(0xff800000);
    }
    const bool whole = (c0 + 8 <= C) && vec_ok && (sizeof(T) == 2) && (((c0 * sizeof(T)) % 16) == 0);
    for (long r = r0; r < r1; ++r) {
        const T *xr = x + r * ldx + c0;
        if (whole) {
            const v8us a = *reinterpret_cast<const v8us *>(xr);
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const float v = Elem<DT>::ld((T)a[i]);
                lo[i] = fminf(lo[i], v);
                hi[i] = fmaxf(hi[i], v);
            }
        } else {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (c0 + i < C) {
                    const float v = Elem<DT>::ld(xr[i]);
                    lo[i] = fminf(lo[i], v);
                    hi[i] = fmaxf(hi[i], v);
                }
            }
        }
    }
    if (r1 > r0) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (c0 + i < C) {
                atomic_min_f32(mn + (c0 + i - col_begin), lo[i]);
                atomic_max_f32(mx + (c0 + i - col_begin), hi[i]);
            }
        }
    }
}

template <int DT>
__global__ __launch_bounds__(256) void minmax_tensor_kernel(
    const typename Elem<DT>::T *__restrict__ x, long M, long C, long ldx, long col_begin,
    float *__restrict__ out2)
{
    typedef typename Elem<DT>::T T;
    const long W = C - col_begin;
    const long total = M * W;
    float lo = __int_as_float(0x7f800000), hi = __int_as_float(0xff800000);
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (long)gridDim.x * blockDim.x) {
        const long r = i / W, c = i - r * W + col_begin;
        const float v = Elem<DT>::ld(x[r * ldx + c]);
        lo = fminf(lo, v);
        hi = fmaxf(hi, v);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        lo = fminf(lo, __shfl_xor(lo, o));
        hi = fmaxf(hi, __shfl_xor(hi, o));
    }
    __shared__ float slo[4], shi[4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) { slo[wave] = lo; shi[wave] = hi; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 4; ++w) { lo = fminf(lo, slo[w]); hi = fmaxf(hi, shi[w]); }
        atomic_min_f32(out2, lo);
        atomic_max_f32(out2 + 1, hi);
    }
}

}  // namespace mq

extern "C" int mq_minmax_channels(const void *x, int x_dtype, long M, long C, long ldx,
                                  long col_begin, float *mn, float *mx, void *stream)
{
    using namespace mq;
    MQ_REQUIRE(M > 0 && C > 0 && col_begin >= 0 && col_begin < C && ldx >= C, "mq_minmax_channels: bad shape");
    MQ_REQUIRE(x && mn && mx, "mq_minmax_channels: null buffer");
    hipStream_t st = (hipStream_t)stream;
    const long W = C - col_begin;
    hipLaunchKernelGGL(minmax_init_kernel, dim3((unsigned)ceil_div(W, 256)), dim3(256), 0, st, mn, mx, W);
    const long col_threads = ceil_div(W, 8);
    const unsigned gx = (unsigned)ceil_div(col_threads, 256);
    long slices = 2048 / (gx ? gx : 1);
    if (slices < 1) slices = 1;
    if (slices > M) slices = M;
    const long rows_per_slice = ceil_div(M, slices);
    slices = ceil_div(M, rows_per_slice);
    const dim3 g(gx, (unsigned)slices), b(256);
    switch (x_dtype) {
    case MQ_F16: {
        const int vec_ok = (((uintptr_t)x) % 16 == 0) && ((ldx * 2) % 16 == 0);
        hipLaunchKernelGGL(minmax_channels_kernel<MQ_F16>, g, b, 0, st, (const unsigned short *)x, M, C, ldx, col_begin, rows_per_slice, mn, mx, vec_ok);
        break;
    }
    case MQ_BF16: {
        const int vec_ok = (((uintptr_t)x) % 16 == 0) && ((ldx * 2) % 16 == 0);
        hipLaunchKernelGGL(minmax_channels_kernel<MQ_BF16>, g, b, 0, st, (const unsigned short *)x, M, C, ldx, col_begin, rows_per_slice, mn, mx, vec_ok);
        break;
    }
    case MQ_F32:
        hipLaunchKernelGGL(minmax_channels_kernel<MQ_F32>, g, b, 0, st, (const float *)x, M, C, ldx, col_begin, rows_per_slice, mn, mx, 0);
        break;
    default: return fail(MQ_EINVAL, "mq_minmax_channels: unknown dtype %d", x_dtype);
    }
    return check_launch("minmax_channels");
}

extern "C" int mq_minmax_tensor(const void *x, int x_dtype, long M, long C, long ldx,
                                long col_begin, float *out2, void *stream)
{
    using namespace mq;
    MQ_REQUIRE(M > 0 && C > 0 && col_begin >= 0 && col_begin < C && ldx >= C, "mq_minmax_tensor: bad shape");
    MQ_REQUIRE(x && out2, "mq_minmax_tensor: null buffer");
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(minmax_init_kernel, dim3(1), dim3(64), 0, st, out2, out2 + 1, 1L);
    long blocks = ceil_div(M * (C - col_begin), 256 * 16);
    if (blocks > 2048) blocks = 2048;
    if (blocks < 1) blocks = 1;
    const dim3 g((unsigned)blocks), b(256);
    switch (x_dtype) {
    case MQ_F16: hipLaunchKernelGGL(minmax_tensor_kernel<MQ_F16>, g, b, 0, st, (const unsigned short *)x, M, C, ldx, col_begin, out2); break;
    case MQ_BF16: hipLaunchKernelGGL(minmax_tensor_kernel<MQ_BF16>, g, b, 0, st, (const unsigned short *)x, M, C, ldx, col_begin, out2); break;
    case MQ_F32: hipLaunchKernelGGL(minmax_tensor_kernel<MQ_F32>, g, b, 0, st, (const float *)x, M, C, ldx, col_begin, out2); break;
    default: return fail(MQ_EINVAL, "mq_minmax_tensor: unknown dtype %d", x_dtype);
    }
    return check_launch("minmax_tensor");
}
