// rope.hip -- rotary position embedding (rotate-half convention) applied in place to the q and k
// column blocks of a fused q|k|v GEMM output.  NOT part of MQuant (the reference never touches
// RoPE); it exists because the whole-prefill TTFT report chains the W4A8 Linears through the
// model's glue and the torch composition of RoPE costs ~8 launches per layer.  Semantics = the HF
// formula on tensors of dtype DT, one rounding per torch op:
//     out = cast(cast(x * cos) + cast(rotate_half(x) * sin)),  rotate_half(x) = cat(-x2, x1)
#include "mq_common.h"

namespace mq {

template <int DT>
__global__ __launch_bounds__(256) void rope_kernel(void *x_, long T, int heads, int head_dim, long ldx,
                                                   const void *cos_, const void *sin_)
{
    typedef typename Elem<DT>::T E;
    E *x = reinterpret_cast<E *>(x_);
    const E *cs = reinterpret_cast<const E *>(cos_), *sn = reinterpret_cast<const E *>(sin_);
    const int half = head_dim / 2;
    const long total = T * heads * half;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int d = (int)(i % half);
        const long th = i / half;
        const int h = (int)(th % heads);
        const long t = th / heads;
        E *p = x + t * ldx + (long)h * head_dim;
        const float a = Elem<DT>::ld(p[d]), b = Elem<DT>::ld(p[d + half]);
        const float c0 = Elem<DT>::ld(cs[t * head_dim + d]), c1 = Elem<DT>::ld(cs[t * head_dim + d + half]);
        const float s0 = Elem<DT>::ld(sn[t * head_dim + d]), s1 = Elem<DT>::ld(sn[t * head_dim + d + half]);
        const float lo = Elem<DT>::rnd(a * c0) + Elem<DT>::rnd(-b * s0);
        const float hi = Elem<DT>::rnd(b * c1) + Elem<DT>::rnd(a * s1);
        p[d] = Elem<DT>::st(lo);
        p[d + half] = Elem<DT>::st(hi);
    }
}

// The same arithmetic, eight pairs per thread with 16-byte accesses and 32-bit index arithmetic (16-bit dtypes, head_dim a
// multiple of 16, 16-byte aligned rows and tables, fewer than 2^31 groups): the scalar form above spends three 64-bit
// divisions and eight 2-byte accesses per pair and took 7.4 us on the prefill's 1.6 M pairs.
template <int DT>
__global__ __launch_bounds__(256) void rope_vec_kernel(void *x_, unsigned groups, unsigned heads, unsigned gph, int head_dim,
                                                       long ldx, const void *cos_, const void *sin_)
{
    typedef typename Elem<DT>::T E;
    E *x = reinterpret_cast<E *>(x_);
    const E *cs = reinterpret_cast<const E *>(cos_), *sn = reinterpret_cast<const E *>(sin_);
    const int half = head_dim / 2;
    const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;     // group of 8 pairs: (t, h, g), g < gph = half / 8
    if (i >= groups) return;
    const unsigned th = i / gph, g = i - th * gph;
    const unsigned t = th / heads, h = th - t * heads;
    E *p = x + (long)t * ldx + (long)h * head_dim + g * 8;
    const E *ct = cs + (long)t * head_dim + g * 8, *st = sn + (long)t * head_dim + g * 8;
    const v8us a8 = *reinterpret_cast<const v8us *>(p), b8 = *reinterpret_cast<const v8us *>(p + half);
    const v8us c0 = *reinterpret_cast<const v8us *>(ct), c1 = *reinterpret_cast<const v8us *>(ct + half);
    const v8us s0 = *reinterpret_cast<const v8us *>(st), s1 = *reinterpret_cast<const v8us *>(st + half);
    v8us lo8, hi8;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const float a = Elem<DT>::ld((E)a8[e]), b = Elem<DT>::ld((E)b8[e]);
        const float lo = Elem<DT>::rnd(a * Elem<DT>::ld((E)c0[e])) + Elem<DT>::rnd(-b * Elem<DT>::ld((E)s0[e]));
        const float hi = Elem<DT>::rnd(b * Elem<DT>::ld((E)c1[e])) + Elem<DT>::rnd(a * Elem<DT>::ld((E)s1[e]));
        lo8[e] = (unsigned short)Elem<DT>::st(lo);
        hi8[e] = (unsigned short)Elem<DT>::st(hi);
    }
    *reinterpret_cast<v8us *>(p) = lo8;
    *reinterpret_cast<v8us *>(p + half) = hi8;
}

}  // namespace mq

extern "C" int mq_rope_inplace(void *x, int x_dtype, long T, int heads, int head_dim, long ldx,
                               const void *cos, const void *sin, void *stream)
{
    using namespace mq;
    if (T == 0 || heads == 0) return MQ_OK;
    MQ_REQUIRE(x && cos && sin && T > 0 && heads > 0 && head_dim > 0 && head_dim % 2 == 0 && ldx >= (long)heads * head_dim,
               "mq_rope_inplace: bad shape");
    const long total = T * heads * (head_dim / 2);
    hipStream_t st = (hipStream_t)stream;
    if ((x_dtype == MQ_F16 || x_dtype == MQ_BF16) && head_dim % 16 == 0 && total / 8 < (1L << 31) &&
        ((uintptr_t)x) % 16 == 0 && (ldx * 2) % 16 == 0 && ((uintptr_t)cos) % 16 == 0 && ((uintptr_t)sin) % 16 == 0) {
        const unsigned groups = (unsigned)(total / 8), gph = (unsigned)(head_dim / 16);
        const unsigned vblocks = (groups + 255) / 256;
        if (x_dtype == MQ_F16) hipLaunchKernelGGL(rope_vec_kernel<MQ_F16>, dim3(vblocks), dim3(256), 0, st, x, groups, (unsigned)heads, gph, head_dim, ldx, cos, sin);
        else hipLaunchKernelGGL(rope_vec_kernel<MQ_BF16>, dim3(vblocks), dim3(256), 0, st, x, groups, (unsigned)heads, gph, head_dim, ldx, cos, sin);
        return check_launch("rope_inplace");
    }
    long blocks = ceil_div(total, 256);
    if (blocks > 4096) blocks = 4096;
    switch (x_dtype) {
    case MQ_F16: hipLaunchKernelGGL(rope_kernel<MQ_F16>, dim3((unsigned)blocks), dim3(256), 0, st, x, T, heads, head_dim, ldx, cos, sin); break;
    case MQ_BF16: hipLaunchKernelGGL(rope_kernel<MQ_BF16>, dim3((unsigned)blocks), dim3(256), 0, st, x, T, heads, head_dim, ldx, cos, sin); break;
    case MQ_F32: hipLaunchKernelGGL(rope_kernel<MQ_F32>, dim3((unsigned)blocks), dim3(256), 0, st, x, T, heads, head_dim, ldx, cos, sin); break;
    default: return fail(MQ_EINVAL, "mq_rope_inplace: unknown dtype %d", x_dtype);
    }
    return check_launch("rope_inplace");
}
