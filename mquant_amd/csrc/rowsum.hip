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

}  // namespace mq

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
