// gemv_f16.hip -- out[m][n] = sum_k x[m][k] * W[n][k] for a handful of rows x (M <= 8) against a LARGE 16-bit matrix W:
// the unquantized lm_head on the last position of a prefill (the reference keeps lm_head in 16 bits: every exam/quant_*.py
// skips it, e.g. exam/quant_qwen2vl.py's skip list; HF computes logits = lm_head(hidden[:, -1:])).  152 064 x 3584 fp16 weights
// are 1.09 GB that are read once: an HBM stream, nothing else.  hipBLASLt's kernel for this shape moves them at 5.0 TB/s (219 us);
// this one at the rate the chip streams (guide: 6.3 TB/s copy, 6.5-6.8 with non-temporal loads).
//
//   * x ([M][K], a few KB) sits in LDS; a wave owns whole rows of W, four at a time (four independent 16-byte loads per lane
//     and step in flight), lane l takes the 16-byte chunks l, l + 64, ... of each row;
//   * products and sums in fp32 (V_DOT2_F32_F16 / fp32 FMA for bf16), one butterfly over the 64 lanes per output, one rounding
//     to the output dtype -- hipBLASLt's accumulation is fp32 as well, the summation ORDER differs (glue of the whole-prefill
//     report, not the W4A8 path: checked against an fp64 product with a half-precision tolerance, tests/test_gpu_gemv.py);
//   * rows are handed out wave by wave in one pass (grid = every wave slot of the chip), no tail round.
#include "mq_common.h"

namespace mq {

struct GemvArgs {
    const void *x;      // [M][K], row stride ldx elements
    const void *w;      // [N][K], row stride ldw elements
    void *out;          // [M][N], row stride ldo elements
    long N, K, ldx, ldw, ldo;
    int M;
};

typedef _Float16 gv_h2 __attribute__((ext_vector_type(2)));

template <int DT>
__device__ __forceinline__ float dot8(const v4i a, const v4i b, float acc)
{
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int aj = a[j], bj = b[j];      // (a bit_cast straight from a vector ELEMENT reads element 0: copy to a scalar first)
        if (DT == MQ_F16) {
            acc = __builtin_amdgcn_fdot2(__builtin_bit_cast(gv_h2, aj), __builtin_bit_cast(gv_h2, bj), acc, false);
        } else {
            const unsigned ua = (unsigned)aj, ub = (unsigned)bj;
            acc = fmaf(__uint_as_float(ua << 16), __uint_as_float(ub << 16), acc);
            acc = fmaf(__uint_as_float(ua & 0xffff0000u), __uint_as_float(ub & 0xffff0000u), acc);
        }
    }
    return acc;
}

constexpr int GV_ROWS = 4;      // rows of W a wave has in flight
constexpr int GV_THREADS = 256;

template <int DT, int M>
__global__ __launch_bounds__(GV_THREADS) void gemv_f16_kernel(GemvArgs p)
{
    kernarg_warm<sizeof(GemvArgs), true>();
    extern __shared__ __attribute__((aligned(16))) char smem[];       // x: [M][K] 16-bit
    const int tid = threadIdx.x, lane = tid & 63;
    const long chunks = p.K / 8;
    for (int c = tid; c < (int)chunks * M; c += GV_THREADS) {
        const int m = c / (int)chunks, cc = c - m * (int)chunks;
        *reinterpret_cast<v4i *>(smem + ((long)m * p.K + cc * 8) * 2) =
            *reinterpret_cast<const v4i *>(reinterpret_cast<const unsigned short *>(p.x) + (long)m * p.ldx + cc * 8);
    }
    __syncthreads();
    const long wave = (long)blockIdx.x * (GV_THREADS / 64) + (tid >> 6), waves = (long)gridDim.x * (GV_THREADS / 64);
    const unsigned short *w = reinterpret_cast<const unsigned short *>(p.w);
    for (long n0 = wave * GV_ROWS; n0 < p.N; n0 += waves * GV_ROWS) {
        float acc[GV_ROWS][M];
#pragma unroll
        for (int r = 0; r < GV_ROWS; ++r)
#pragma unroll
            for (int m = 0; m < M; ++m) acc[r][m] = 0.0f;
        // the last group of a matrix whose N is not a multiple of four re-reads its last row (never stored)
        const unsigned short *wr[GV_ROWS];
#pragma unroll
        for (int r = 0; r < GV_ROWS; ++r) wr[r] = w + (n0 + r < p.N ? n0 + r : p.N - 1) * p.ldw;
        for (long c = lane; c < chunks; c += 64) {
            v4i wv[GV_ROWS];
#pragma unroll
            for (int r = 0; r < GV_ROWS; ++r) wv[r] = __builtin_nontemporal_load(reinterpret_cast<const v4i *>(wr[r] + c * 8));
#pragma unroll
            for (int m = 0; m < M; ++m) {
                const v4i xv = *reinterpret_cast<const v4i *>(smem + ((long)m * p.K + c * 8) * 2);
#pragma unroll
                for (int r = 0; r < GV_ROWS; ++r) acc[r][m] = dot8<DT>(wv[r], xv, acc[r][m]);
            }
        }
#pragma unroll
        for (int r = 0; r < GV_ROWS; ++r)
#pragma unroll
            for (int m = 0; m < M; ++m) {
                float v = acc[r][m];
#pragma unroll
                for (int st = 1; st < 64; st <<= 1) v += __shfl_xor(v, st, 64);
                acc[r][m] = v;
            }
        if (lane < GV_ROWS * M) {                                     // lane (m, r) stores out[m][n0 + r]
            const int r = lane % GV_ROWS, m = lane / GV_ROWS;
            float v = 0.0f;
#pragma unroll
            for (int rr = 0; rr < GV_ROWS; ++rr)
#pragma unroll
                for (int mm = 0; mm < M; ++mm)
                    if (rr == r && mm == m) v = acc[rr][mm];
            if (n0 + r < p.N)
                reinterpret_cast<unsigned short *>(p.out)[(long)m * p.ldo + n0 + r] = (unsigned short)Elem<DT>::st(v);
        }
    }
}

}  // namespace mq

template <int DT>
static void gemv_launch(const mq::GemvArgs &a, hipStream_t st)
{
    using namespace mq;
    const size_t lds = (size_t)a.M * a.K * 2;
    // every wave slot of the chip that the rows can fill: 8 workgroups of 4 waves per CU
    long blocks = (long)device_cu_count() * 8;
    const long need = ceil_div(ceil_div(a.N, GV_ROWS), GV_THREADS / 64);
    if (blocks > need) blocks = need;
    switch (a.M) {
    case 1: hipLaunchKernelGGL((gemv_f16_kernel<DT, 1>), dim3((unsigned)blocks), dim3(GV_THREADS), lds, st, a); break;
    case 2: hipLaunchKernelGGL((gemv_f16_kernel<DT, 2>), dim3((unsigned)blocks), dim3(GV_THREADS), lds, st, a); break;
    case 3: hipLaunchKernelGGL((gemv_f16_kernel<DT, 3>), dim3((unsigned)blocks), dim3(GV_THREADS), lds, st, a); break;
    case 4: hipLaunchKernelGGL((gemv_f16_kernel<DT, 4>), dim3((unsigned)blocks), dim3(GV_THREADS), lds, st, a); break;
    case 5: hipLaunchKernelGGL((gemv_f16_kernel<DT, 5>), dim3((unsigned)blocks), dim3(GV_THREADS), lds, st, a); break;
    case 6: hipLaunchKernelGGL((gemv_f16_kernel<DT, 6>), dim3((unsigned)blocks), dim3(GV_THREADS), lds, st, a); break;
    case 7: hipLaunchKernelGGL((gemv_f16_kernel<DT, 7>), dim3((unsigned)blocks), dim3(GV_THREADS), lds, st, a); break;
    default: hipLaunchKernelGGL((gemv_f16_kernel<DT, 8>), dim3((unsigned)blocks), dim3(GV_THREADS), lds, st, a); break;
    }
}

extern "C" int mq_gemv_f16(const void *x, int dtype, int M, long K, long ldx, const void *w, long N, long ldw, void *out, long ldo,
                           void *stream)
{
    using namespace mq;
    MQ_REQUIRE(dtype == MQ_F16 || dtype == MQ_BF16, "mq_gemv_f16: dtype must be fp16 or bf16 (got %d)", dtype);
    MQ_REQUIRE(M >= 0 && M <= 8 && N >= 0 && K > 0 && K % 8 == 0, "mq_gemv_f16: M must be 0..8 and K a positive multiple of 8 (M %d, N %ld, K %ld)", M, N, K);
    if (M == 0 || N == 0) return MQ_OK;
    MQ_REQUIRE(x && w && out, "mq_gemv_f16: null pointer");
    MQ_REQUIRE(ldx >= K && ldw >= K && ldo >= N, "mq_gemv_f16: row strides too short");
    MQ_REQUIRE(((uintptr_t)x) % 16 == 0 && (ldx * 2) % 16 == 0 && ((uintptr_t)w) % 16 == 0 && (ldw * 2) % 16 == 0,
               "mq_gemv_f16: x / W rows must be 16-byte aligned");
    MQ_REQUIRE((long)M * K * 2 <= 64 * 1024, "mq_gemv_f16: x (%d x %ld) must fit 64 KiB of LDS", M, K);
    GemvArgs a{x, w, out, N, K, ldx, ldw, ldo, M};
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (dtype == MQ_F16) gemv_launch<MQ_F16>(a, st);
    else gemv_launch<MQ_BF16>(a, st);
    return check_launch("gemv_f16");
}
