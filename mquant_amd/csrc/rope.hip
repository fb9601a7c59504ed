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

}  // namespace mq

extern "C" int mq_rope_inplace(void *x, int x_dtype, long T, int heads, int head_dim, long ldx,
                               const void *cos, const void *sin, void *stream)
{
    using namespace mq;
    if (T == 0 || heads == 0) return MQ_OK;
    MQ_REQUIRE(x && cos && sin && T > 0 && heads > 0 && head_dim > 0 && head_dim % 2 == 0 && ldx >= (long)heads * head_dim,
               "mq_rope_inplace: bad shape");
    const long total = T * heads * (head_dim / 2);
    long blocks = ceil_div(total, 256);
    if (blocks > 4096) blocks = 4096;
    hipStream_t st = (hipStream_t)stream;
    switch (x_dtype) {
    case MQ_F16: hipLaunchKernelGGL(rope_kernel<MQ_F16>, dim3((unsigned)blocks), dim3(256), 0, st, x, T, heads, head_dim, ldx, cos, sin); break;
    case MQ_BF16: hipLaunchKernelGGL(rope_kernel<MQ_BF16>, dim3((unsigned)blocks), dim3(256), 0, st, x, T, heads, head_dim, ldx, cos, sin); break;
    case MQ_F32: hipLaunchKernelGGL(rope_kernel<MQ_F32>, dim3((unsigned)blocks), dim3(256), 0, st, x, T, heads, head_dim, ldx, cos, sin); break;
    default: return fail(MQ_EINVAL, "mq_rope_inplace: unknown dtype %d", x_dtype);
    }
    return check_launch("rope_inplace");
}
