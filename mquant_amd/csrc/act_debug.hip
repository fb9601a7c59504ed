// act_debug.hip -- TEST-ONLY: the fused activations of the 16-bit dtypes, fast form against reference form, element by element.
//
// silu / sigmoid of a half-precision tensor are functions of 65 536 inputs; mq_common.h evaluates them with fewer instructions than
// the device library's expf + IEEE division and claims the SAME rounded results.  tests/test_gpu_act_exhaustive.py holds that claim
// to every bit pattern of fp16 and bf16 through this entry point (and the packed two-at-a-time forms of the GEMM act epilogues to the
// scalar ones on random operand pairs).  Reference semantics: torch's silu / sigmoid / mul kernels on half tensors (fp32 arithmetic,
// one rounding per op) as the HF modules around fake_quant/quant_utils.py:330-391 call them.
#include "mq_common.h"

namespace mq {

template <int DT>
__global__ __launch_bounds__(256) void act_table_kernel(int which, const unsigned short *in, const unsigned short *in2, long n,
                                                        unsigned short *out_fast, unsigned short *out_ref)
{
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float x = Elem<DT>::ld(in[i]);
    const float u = in2 ? Elem<DT>::ld(in2[i]) : 1.0f;
    float f = 0.0f, r = 0.0f;
    switch (which) {
    case 0: f = act_silu_16<DT>(x); r = act_silu_ref<DT>(x); break;
    case 1: f = act_sigmoid_16<DT>(x); r = act_sigmoid_ref<DT>(x); break;
    case 2: f = act_silu_mul<DT>(x, u); r = Elem<DT>::rnd(act_silu_ref<DT>(x) * u); break;
    case 3: {
        f = act_quick_gelu<DT>(x);
        const float z = Elem<DT>::rnd(1.702f * x);
        r = Elem<DT>::rnd(x * act_sigmoid_ref<DT>(z));
        break;
    }
    case 4: {      // packed forms: this element and its neighbour (i ^ 1) as the pair; the fp32 inputs are the half values themselves
        const long j = i ^ 1;
        const float xj = j < n ? Elem<DT>::ld(in[j]) : x, uj = (in2 && j < n) ? Elem<DT>::ld(in2[j]) : u;
        const unsigned pk = (i & 1) ? act_silu_mul_pk<DT>(xj, x, uj, u) : act_silu_mul_pk<DT>(x, xj, u, uj);
        out_fast[i] = (unsigned short)((i & 1) ? (pk >> 16) : (pk & 0xffffu));
        out_ref[i] = Elem<DT>::st(Elem<DT>::rnd(act_silu_ref<DT>(x) * u));
        return;
    }
    default: {
        const long j = i ^ 1;
        const float xj = j < n ? Elem<DT>::ld(in[j]) : x;
        const unsigned pk = (i & 1) ? act_quick_gelu_pk<DT>(xj, x) : act_quick_gelu_pk<DT>(x, xj);
        out_fast[i] = (unsigned short)((i & 1) ? (pk >> 16) : (pk & 0xffffu));
        const float z = Elem<DT>::rnd(1.702f * x);
        out_ref[i] = Elem<DT>::st(Elem<DT>::rnd(x * act_sigmoid_ref<DT>(z)));
        return;
    }
    }
    out_fast[i] = Elem<DT>::st(f);
    out_ref[i] = Elem<DT>::st(r);
}

}  // namespace mq

extern "C" int mq_debug_act_table(int dtype, int which, const void *in_bits, const void *in2_bits, long n, void *out_fast, void *out_ref, void *stream)
{
    using namespace mq;
    MQ_REQUIRE(dtype == MQ_F16 || dtype == MQ_BF16, "mq_debug_act_table: fp16 or bf16");
    MQ_REQUIRE(which >= 0 && which <= 5 && in_bits && out_fast && out_ref && n >= 0, "mq_debug_act_table: bad arguments");
    if (n == 0) return MQ_OK;
    const dim3 grid((unsigned)ceil_div(n, 256));
    hipStream_t st = (hipStream_t)stream;
    if (dtype == MQ_F16)
        hipLaunchKernelGGL(act_table_kernel<MQ_F16>, grid, dim3(256), 0, st, which, (const unsigned short *)in_bits, (const unsigned short *)in2_bits, n,
                           (unsigned short *)out_fast, (unsigned short *)out_ref);
    else
        hipLaunchKernelGGL(act_table_kernel<MQ_BF16>, grid, dim3(256), 0, st, which, (const unsigned short *)in_bits, (const unsigned short *)in2_bits, n,
                           (unsigned short *)out_fast, (unsigned short *)out_ref);
    return check_launch("act_table");
}
