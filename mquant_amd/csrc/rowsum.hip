// rowsum.hip -- scaled row sums of the int8 activation levels: the per-row factor of the rank-1 epilogue
// term that carries the zero points of ASYMMETRIC weights (--w_asym; WeightQuantizer with sym = False,
// fake_quant/quant_utils.py:446-509).  With stored weight levels q' = q - 2^(b-1) the fake-quantized
// weight is W~[n][k] = s_w[n] (q'[n][k] + 2^(b-1) - z_w[n]), so
//     y[m][n] = ((acc * s_x) * s_w[n]) + bias[n] + (s_x * sum_k a[m][k]) * (s_w[n] (2^(b-1) - z_w[n]))
// and the GEMM's x0 / w0 slot takes the two factors.  out[m] = s_x(m) * float(sum_k a[m][k]); the integer
// sum is exact, one rounding in the product.  One wave per row, 16 bytes per lane per access.
#include "mq_common.h"

namespace mq {

struct RsArgs {
    const int8_t *a;
    long M, K_pad, lda;
    float sx0, sx1;
    const uint8_t *row_sel;
    const float *sx_vec;
    float *out;
};

__global__ __launch_bounds__(256) void act_rowsum_kernel(RsArgs p)
{
    kernarg_warm<sizeof(RsArgs)>();
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= p.M) return;
    int acc = 0;
    for (long k = lane * 16L; k < p.K_pad; k += 64 * 16L) {
        const v4i v = *reinterpret_cast<const v4i *>(p.a + act_offset(row, k, p.K_pad, p.lda));
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int w = v[j];
            acc += (int)(signed char)(w & 0xff) + (int)(signed char)((w >> 8) & 0xff) +
                   (int)(signed char)((w >> 16) & 0xff) + (w >> 24);
        }
    }
#pragma unroll
    for (int st = 1; st < 64; st <<= 1) acc += __shfl_xor(acc, st, 64);
    if (lane == 0) {
        float sx = p.sx0;
        if (p.sx_vec) sx = p.sx_vec[row];
        else if (p.row_sel && p.row_sel[row]) sx = p.sx1;
        p.out[row] = sx * (float)acc;
    }
}

// out[m][n] = cast(y32[m][n] + x[m] * w[n]): the THIRD rank-1 term of a layer that uses the split column, asymmetric weights and
// asymmetric dynamic activations at once (the GEMM epilogue has two slots).  The GEMM runs with an fp32 output, so this pass
// continues the epilogue's sum in fp32 and rounds to the output dtype once -- the same arithmetic a third slot would do.
struct R1Args {
    const float *y;
    long M, N, ldy;
    const float *x, *w;
    void *out;
    long ldo;
};

template <int DT>
__global__ __launch_bounds__(256) void rank1_add_cast_kernel(R1Args p)
{
    kernarg_warm<sizeof(R1Args), true>();
    typedef typename Elem<DT>::T T;
    const long quads = (p.N + 3) / 4;
    const long total = p.M * quads;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long m = i / quads, n = (i - m * quads) * 4;
        const float xm = p.x[m];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            if (n + r >= p.N) break;
            const float pr = xm * p.w[n + r];
            const float v = p.y[m * p.ldy + n + r] + pr;
            reinterpret_cast<T *>(p.out)[m * p.ldo + n + r] = Elem<DT>::st(v);
        }
    }
}

}  // namespace mq

extern "C" int mq_rank1_add_cast(const float *y32, long M, long N, long ldy, const float *x, const float *w, void *out, int out_dtype,
                                 long ldo, void *stream)
{
    using namespace mq;
    if (M == 0 || N == 0) return MQ_OK;
    MQ_REQUIRE(y32 && x && w && out && M > 0 && N > 0 && ldy >= N && ldo >= N, "mq_rank1_add_cast: bad arguments");
    R1Args p{y32, M, N, ldy, x, w, out, ldo};
    long blocks = ceil_div(M * ((N + 3) / 4), 256);
    if (blocks > 4096) blocks = 4096;
    hipStream_t st = (hipStream_t)stream;
    switch (out_dtype) {
    case MQ_F16: hipLaunchKernelGGL(rank1_add_cast_kernel<MQ_F16>, dim3((unsigned)blocks), dim3(256), 0, st, p); break;
    case MQ_BF16: hipLaunchKernelGGL(rank1_add_cast_kernel<MQ_BF16>, dim3((unsigned)blocks), dim3(256), 0, st, p); break;
    case MQ_F32: hipLaunchKernelGGL(rank1_add_cast_kernel<MQ_F32>, dim3((unsigned)blocks), dim3(256), 0, st, p); break;
    default: return fail(MQ_EINVAL, "mq_rank1_add_cast: unknown dtype %d", out_dtype);
    }
    return check_launch("rank1_add_cast");
}

extern "C" int mq_act_rowsum_scaled(const int8_t *a, long lda, long M, long K_pad, float s_x0, float s_x1,
                                    const uint8_t *row_sel, const float *s_x_rows, float *out, void *stream)
{
    using namespace mq;
    if (M == 0) return MQ_OK;
    MQ_REQUIRE(a && out && M > 0 && K_pad > 0 && K_pad % 16 == 0, "mq_act_rowsum_scaled: bad shape");
    MQ_REQUIRE(((uintptr_t)a) % 16 == 0 && (lda == MQ_LD_TILED ? K_pad % 64 == 0 : (lda >= K_pad && lda % 16 == 0)),
               "mq_act_rowsum_scaled: bad lda / alignment");
    RsArgs p{a, M, K_pad, lda, s_x0, s_x1, row_sel, s_x_rows, out};
    hipLaunchKernelGGL(act_rowsum_kernel, dim3((unsigned)((M + 3) / 4)), dim3(256), 0, (hipStream_t)stream, p);
    return check_launch("act_rowsum_scaled");
}
