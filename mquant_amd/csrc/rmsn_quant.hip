// rmsn_quant.hip -- weight-less RMS normalisation fused with the static activation quantizer
// ("RMSN fused into the act-quantize, norm -> scale -> int8 in one read").
//
// Reference: module_util.py:42-61 (RMSN.forward: fp16 is promoted to fp32, the result is cast back
// to the input dtype) followed by uniform.py:20-33 (q = clamp(rint(y / s), -128, 127)).  In the
// models every norm in front of a wrapped Linear is such an RMSN after the LayerNorm fusion
// (qwen2vl_rotation.py:46-71, internvl_rotation.py:177-206), so norm + quantize are always adjacent.
//
// One workgroup per row, the row stays in registers between the two passes (one HBM read, one
// int8 write; optionally the normalised row in the input dtype as well).  The sum of squares uses
// (bf16 rows are not promoted upstream -- only fp16 is, module_util.py:56-57 -- so every step of the
// bf16 path rounds to bf16 like the torch ops do) a FIXED order that the oracle restates (orc_rmsn): thread t adds the squares of its 16-element
// chunks c = t, t + 256, ... in ascending k; XOR butterfly (1..32) inside each wave; the four wave
// sums left to right.  1/sqrt with correctly rounded sqrt and divide (torch's CPU rsqrt).
#include "mq_common.h"

namespace mq {

constexpr int RQ_THREADS = 256;
constexpr int RQ_MAX_CHUNKS = 4;   // 16-element chunks per thread: rows up to 16384 elements

struct RqArgs {
    const void *x;
    long M, K, ldx;
    float mean_dim, eps, scale0, scale1;
    const uint8_t *row_sel;
    void *y;
    long ldy;
    int8_t *out;
    long K_pad, ldo;
};

template <int DT>
__global__ __launch_bounds__(RQ_THREADS) void rmsn_quant_kernel(RqArgs p)
{
    kernarg_warm<sizeof(RqArgs)>();                 // one scalar-load round trip for the argument block (mq_common.h)
    typedef typename Elem<DT>::T T;
    __shared__ float wsum[4];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    // tiled int8 output: the 16 rows of a piece row are handled on ONE XCD (tiled_row_of, mq_common.h), so the eight
    // 16-byte chunks that share a 128-byte line meet in one L2 instead of being written back by eight
    const long row = (p.ldo == MQ_LD_TILED) ? tiled_row_of(blockIdx.x) : (long)blockIdx.x;
    if (row >= p.M) return;                         // uniform over the workgroup
    const T *xr = reinterpret_cast<const T *>(p.x) + row * p.ldx;
    const long chunks = p.K / 16;

    float v[RQ_MAX_CHUNKS][16];
    float part = 0.0f;
#pragma unroll
    for (int c = 0; c < RQ_MAX_CHUNKS; ++c) {
        const long ch = t + (long)c * RQ_THREADS;
        if (ch < chunks) {
            const T *src = xr + ch * 16;
            if (sizeof(T) == 2) {
                const v8us a = *reinterpret_cast<const v8us *>(src);
                const v8us b = *reinterpret_cast<const v8us *>(src + 8);
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    v[c][i] = Elem<DT>::ld((T)a[i]);
                    v[c][8 + i] = Elem<DT>::ld((T)b[i]);
                }
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const v4f a = *reinterpret_cast<const v4f *>((const float *)src + 4 * j);
#pragma unroll
                    for (int i = 0; i < 4; ++i) v[c][4 * j + i] = a[i];
                }
            }
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                float sq = v[c][i] * v[c][i];
                if (DT == MQ_BF16) sq = Elem<DT>::rnd(sq);      // bf16 rows are NOT promoted upstream: x.pow(2) is bf16
                part = part + sq;
            }
        }
    }
#pragma unroll
    for (int st = 1; st < 64; st <<= 1) part = part + __shfl_xor(part, st, 64);
    if (lane == 0) wsum[wave] = part;
    __syncthreads();
    const float total = ((wsum[0] + wsum[1]) + wsum[2]) + wsum[3];
    float inv;
    if (DT == MQ_BF16) {
        // module_util.py:58-60 on a bf16 tensor: every torch op rounds its fp32 result to bf16
        const float var = Elem<DT>::rnd(Elem<DT>::rnd(total) / p.mean_dim);
        const float ve = Elem<DT>::rnd(var + p.eps);
        inv = Elem<DT>::rnd(1.0f / sqrtf(ve));
    } else {
        const float ms = total / p.mean_dim;
        inv = 1.0f / sqrtf(ms + p.eps);
    }
    const float s = (p.row_sel && p.row_sel[row]) ? p.scale1 : p.scale0;
    const float s_inv = 1.0f / s;
    const bool s_rcp = quant_rcp_ok(s);

#pragma unroll
    for (int c = 0; c < RQ_MAX_CHUNKS; ++c) {
        const long ch = t + (long)c * RQ_THREADS;
        if (ch < chunks) {
            int q[16];
            float y[16];
#pragma unroll
            for (int i = 0; i < 16; ++i) y[i] = Elem<DT>::rnd(v[c][i] * inv);
            quant_levels<16>(y, s, s_inv, s_rcp, -128.0f, 127.0f, q);      // the IEEE quotient only next to a half-integer (mq_common.h)
            v4i pk;
#pragma unroll
            for (int j = 0; j < 4; ++j)
                pk[j] = (q[4 * j] & 0xff) | ((q[4 * j + 1] & 0xff) << 8) | ((q[4 * j + 2] & 0xff) << 16) |
                        ((q[4 * j + 3] & 0xff) << 24);
            *reinterpret_cast<v4i *>(p.out + act_offset(row, ch * 16, p.K_pad, p.ldo)) = pk;
            if (p.y) {
                T *yd = reinterpret_cast<T *>(p.y) + row * p.ldy + ch * 16;
#pragma unroll
                for (int i = 0; i < 16; ++i) yd[i] = Elem<DT>::st(y[i]);
            }
        }
    }
    // zero the K .. K_pad tail of the int8 row (the GEMM reads whole 128-byte k-steps)
    for (long k = p.K + t * 16L; k < p.K_pad; k += RQ_THREADS * 16L)
        *reinterpret_cast<v4i *>(p.out + act_offset(row, k, p.K_pad, p.ldo)) = v4i{0, 0, 0, 0};
}

}  // namespace mq

extern "C" int mq_rmsn_quantize_i8(const void *x, int x_dtype, long M, long K, long ldx, float mean_dim,
                                   float eps, float scale0, float scale1, const uint8_t *row_sel,
                                   void *y_out, long ldy, int8_t *out, long K_pad, long ldo, void *stream)
{
    using namespace mq;
    if (M == 0) return MQ_OK;                       // empty input: nothing to do (null pointers allowed)
    MQ_REQUIRE(x && out && M >= 0 && K > 0 && ldx >= K, "mq_rmsn_quantize_i8: bad shape");
    MQ_REQUIRE(K % 16 == 0 && K <= 16L * RQ_THREADS * RQ_MAX_CHUNKS,
               "mq_rmsn_quantize_i8: K must be a multiple of 16 and <= %d (got %ld)", 16 * RQ_THREADS * RQ_MAX_CHUNKS, K);
    MQ_REQUIRE(K_pad >= K && K_pad % 16 == 0 && (ldo == MQ_LD_TILED ? K_pad % 64 == 0 : (ldo >= K_pad && ldo % 16 == 0)),
               "mq_rmsn_quantize_i8: bad K_pad / ldo");
    MQ_REQUIRE(!y_out || ldy >= K, "mq_rmsn_quantize_i8: ldy < K");
    MQ_REQUIRE(scale0 > 0.0f && mean_dim > 0.0f, "mq_rmsn_quantize_i8: scale and mean_dim must be positive");
    MQ_REQUIRE(((uintptr_t)x) % 16 == 0 && (ldx * (x_dtype == MQ_F32 ? 4 : 2)) % 16 == 0,
               "mq_rmsn_quantize_i8: x rows must be 16-byte aligned");
    if (M == 0) return MQ_OK;
    RqArgs p;
    p.x = x; p.M = M; p.K = K; p.ldx = ldx; p.mean_dim = mean_dim; p.eps = eps;
    p.scale0 = scale0; p.scale1 = row_sel ? scale1 : scale0; p.row_sel = row_sel;
    p.y = y_out; p.ldy = ldy; p.out = out; p.K_pad = K_pad; p.ldo = ldo;
    hipStream_t st = (hipStream_t)stream;
    const unsigned grid = (unsigned)(ldo == MQ_LD_TILED ? ceil_div(M, 128) * 128 : M);      // tiled: whole groups of 8 XCDs x 16 rows
    switch (x_dtype) {
    case MQ_F16: hipLaunchKernelGGL(rmsn_quant_kernel<MQ_F16>, dim3(grid), dim3(RQ_THREADS), 0, st, p); break;
    case MQ_F32: hipLaunchKernelGGL(rmsn_quant_kernel<MQ_F32>, dim3(grid), dim3(RQ_THREADS), 0, st, p); break;
    case MQ_BF16: hipLaunchKernelGGL(rmsn_quant_kernel<MQ_BF16>, dim3(grid), dim3(RQ_THREADS), 0, st, p); break;
    default: return fail(MQ_EINVAL, "mq_rmsn_quantize_i8: unknown dtype %d", x_dtype);
    }
    return check_launch("rmsn_quantize_i8");
}
