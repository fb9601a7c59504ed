// gptq_block.hip -- the column loop of GPTQ's lazy-batch block ("GPTQ's column loop
// is latency-bound").  Reference fake_quant/gptq/gptq_utils.py:258-279, symmetric per-channel
// quantizer without groups:
//     for i in 0..cols-1:
//         w   = W1[:, i];  d = Hinv1[i, i]
//         q   = scale * clamp(rint(w / scale), -(maxq+1), maxq)        (sym_quant_dequant)
//         err = (w - q) / d
//         W1[:, i:] -= err (x) Hinv1[i, i:]                              (rank-1, K = 1 matmul)
//         Q1[:, i] = q;  Err1[:, i] = err
// In torch that is ~8 launches per column (1000 per block).  The rows of W1 never interact, so one
// LANE owns one output row and walks the columns sequentially with exactly the reference's
// operation order (one rounding per written fp32 operation; -ffp-contract=off): results are bit
// identical to the torch loop.  A wave stages its 64 x cols tile in LDS (column-major, padded:
// conflict-free), Hinv1 rows are wave-uniform (scalar loads).
#include "mq_common.h"

namespace mq {

constexpr int GB_COLS = 128;   // GPTQ blocksize upper bound
constexpr int GB_LD = 65;      // 64 rows + 1 pad

struct GbArgs {
    const float *W1;
    long N, ldw;
    int cols;
    const float *H;
    long ldh;
    const float *scale;
    float lo, hi;
    float *Q1;
    long ldq;
    float *E1;
    long lde;
};

__global__ __launch_bounds__(64) void gptq_block_kernel(GbArgs p)
{
    __shared__ float wl[GB_COLS * GB_LD];   // working weights, then q
    __shared__ float el[GB_COLS * GB_LD];   // err
    const int lane = threadIdx.x;
    const long n0 = (long)blockIdx.x * 64;
    const int rows = (int)((p.N - n0) < 64 ? (p.N - n0) : 64);

    // coalesced tile load: row r, columns lane and lane + 64
    for (int r = 0; r < rows; ++r) {
        const float *src = p.W1 + (n0 + r) * p.ldw;
        if (lane < p.cols) wl[lane * GB_LD + r] = src[lane];
        if (lane + 64 < p.cols) wl[(lane + 64) * GB_LD + r] = src[lane + 64];
    }
    __syncthreads();

    if (lane < rows) {
        const float s = p.scale[n0 + lane];
        for (int i = 0; i < p.cols; ++i) {
            const float *hrow = p.H + (long)i * p.ldh;
            const float w = wl[i * GB_LD + lane];
            float lv = rintf(w / s);
            lv = fminf(fmaxf(lv, p.lo), p.hi);
            const float q = s * lv;
            const float err = (w - q) / hrow[i];
            for (int j = i + 1; j < p.cols; ++j) {
                const float prod = err * hrow[j];
                wl[j * GB_LD + lane] = wl[j * GB_LD + lane] - prod;
            }
            wl[i * GB_LD + lane] = q;
            el[i * GB_LD + lane] = err;
        }
    }
    __syncthreads();

    for (int r = 0; r < rows; ++r) {
        float *qd = p.Q1 + (n0 + r) * p.ldq;
        float *ed = p.E1 + (n0 + r) * p.lde;
        if (lane < p.cols) {
            qd[lane] = wl[lane * GB_LD + r];
            ed[lane] = el[lane * GB_LD + r];
        }
        if (lane + 64 < p.cols) {
            qd[lane + 64] = wl[(lane + 64) * GB_LD + r];
            ed[lane + 64] = el[(lane + 64) * GB_LD + r];
        }
    }
}

}  // namespace mq

extern "C" int mq_gptq_block(const float *W1, long N, int cols, long ldw, const float *Hinv1, long ldh,
                             const float *scale, int bits, float *Q1, long ldq, float *Err1, long lde,
                             void *stream)
{
    using namespace mq;
    if (N == 0) return MQ_OK;                       // empty input: nothing to do (null pointers allowed)
    MQ_REQUIRE(W1 && Hinv1 && scale && Q1 && Err1, "mq_gptq_block: null argument");
    MQ_REQUIRE(N >= 0 && cols >= 1 && cols <= GB_COLS, "mq_gptq_block: cols must be 1..%d (got %d)", GB_COLS, cols);
    MQ_REQUIRE(ldw >= cols && ldh >= cols && ldq >= cols && lde >= cols, "mq_gptq_block: leading dimension < cols");
    MQ_REQUIRE(bits >= 2 && bits <= 8, "mq_gptq_block: bits must be 2..8");
    if (N == 0) return MQ_OK;
    GbArgs p;
    p.W1 = W1; p.N = N; p.ldw = ldw; p.cols = cols; p.H = Hinv1; p.ldh = ldh; p.scale = scale;
    p.hi = (float)((1 << (bits - 1)) - 1);
    p.lo = -(p.hi + 1.0f);
    p.Q1 = Q1; p.ldq = ldq; p.E1 = Err1; p.lde = lde;
    hipLaunchKernelGGL(gptq_block_kernel, dim3((unsigned)ceil_div(N, 64)), dim3(64), 0, (hipStream_t)stream, p);
    return check_launch("gptq_block");
}
