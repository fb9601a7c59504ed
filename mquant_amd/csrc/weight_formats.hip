// weight_formats.hip -- int4 wire format (reference quant_utils.py:61-94), weight levels,
// and the pre-tiled images mq_gemm_w4a8 streams into LDS with lane-linear 16-byte DMA.
//
// W4 image  [ntp = N_pad/32][kt = K_pad/64][lane = 0..63][16 B]      (N_pad = ceil32(N))
//   one 1 KiB piece = the B-fragments of TWO adjacent 16-channel tiles for ONE 64-wide k-tile of
//   V_MFMA_I32_16X16X64_I8: bytes 8h..8h+7 of lane l belong to channel n = 32*ntp + 16*h + (l & 15)
//   and hold the 16 reduction indices k0..k0+15, k0 = 64*kt + 16*(l >> 4).
//   Inside a half, little-endian words P0,P1; nibble i = bits [4i,4i+4):
//     P0 nibble 2j   -> k0 + j        P0 nibble 2j+1 -> k0 + 4 + j
//     P1 nibble 2j   -> k0 + 8 + j    P1 nibble 2j+1 -> k0 + 12 + j     (j = 0..3)
//   so (P << 4) & 0xF0F0F0F0 and P & 0xF0F0F0F0 are the four int8 operand words with the
//   level in the HIGH nibble (value = 16 * level; the epilogue shifts the factor out).
// W8 image  [nt][kt = K_pad/64][lane][16 B], lane l: n as above, k0 = 64*kt + 16*(l >> 4).
#include "mq_common.h"

namespace mq {

__global__ void pack_i4_kernel(const int8_t *__restrict__ q, long rows, long cols,
                               uint8_t *__restrict__ out)
{
    const long half = cols / 2;
    const long total = rows * half;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (long)gridDim.x * blockDim.x) {
        const long r = i / half, c = i - r * half;
        const unsigned lo = (unsigned)q[r * cols + 2 * c] & 0xf;
        const unsigned hi = (unsigned)q[r * cols + 2 * c + 1] & 0xf;
        out[i] = (uint8_t)(lo | (hi << 4));
    }
}

__global__ void unpack_i4_kernel(const uint8_t *__restrict__ p, long rows, long cols,
                                 int8_t *__restrict__ out)
{
    const long half = cols / 2;
    const long total = rows * half;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (long)gridDim.x * blockDim.x) {
        const long r = i / half, c = i - r * half;
        const int b = p[i];
        const int lo = b & 0xf, hi = (b >> 4) & 0xf;
        out[r * cols + 2 * c] = (int8_t)(lo >= 8 ? lo - 16 : lo);
        out[r * cols + 2 * c + 1] = (int8_t)(hi >= 8 ? hi - 16 : hi);
    }
}

template <int DT>
__global__ void weight_levels_kernel(const typename Elem<DT>::T *__restrict__ w, long N, long K,
                                     long ldw, const float *__restrict__ scale, float lo,
                                     float hi, int8_t *__restrict__ q)
{
    const long total = N * K;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (long)gridDim.x * blockDim.x) {
        const long n = i / K, k = i - n * K;
        q[i] = (int8_t)quant_level(Elem<DT>::ld(w[n * ldw + k]), scale[n], lo, hi);
    }
}

__device__ __forceinline__ int level_at(const int8_t *q, long N, long K, long n, long k,
                                        int zero_col0)
{
    if (n >= N || k >= K) return 0;
    if (zero_col0 && k == 0) return 0;
    return q[n * K + k];
}

__global__ void prepack_w4_kernel(const int8_t *__restrict__ q, long N, long K, long N_pad,
                                  long K_pad, int zero_col0, uint8_t *__restrict__ out)
{
    const long kts = K_pad / 64;
    const long total = (N_pad / 32) * kts * 64;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (long)gridDim.x * blockDim.x) {
        const int lane = (int)(i & 63);
        const long frag = i >> 6;
        const long ntp = frag / kts, kt = frag - ntp * kts;
        const long k0 = kt * 64 + (lane >> 4) * 16;
        unsigned words[4];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const long n = ntp * 32 + h * 16 + (lane & 15);
#pragma unroll
            for (int p = 0; p < 2; ++p) {
                unsigned wv = 0;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const unsigned a = (unsigned)level_at(q, N, K, n, k0 + 8 * p + j, zero_col0) & 0xf;
                    const unsigned b = (unsigned)level_at(q, N, K, n, k0 + 8 * p + 4 + j, zero_col0) & 0xf;
                    wv |= a << (8 * j);
                    wv |= b << (8 * j + 4);
                }
                words[2 * h + p] = wv;
            }
        }
        v4i o = {(int)words[0], (int)words[1], (int)words[2], (int)words[3]};
        reinterpret_cast<v4i *>(out)[i] = o;
    }
}

__global__ void prepack_w8_kernel(const int8_t *__restrict__ q, long N, long K, long N_pad,
                                  long K_pad, int zero_col0, int8_t *__restrict__ out)
{
    const long kts = K_pad / 64;
    const long total = (N_pad / 16) * kts * 64;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (long)gridDim.x * blockDim.x) {
        const int lane = (int)(i & 63);
        const long frag = i >> 6;
        const long nt = frag / kts, kt = frag - nt * kts;
        const long n = nt * 16 + (lane & 15);
        const long k0 = kt * 64 + (lane >> 4) * 16;
        unsigned words[4];
#pragma unroll
        for (int wd = 0; wd < 4; ++wd) {
            unsigned wv = 0;
#pragma unroll
            for (int j = 0; j < 4; ++j)
                wv |= ((unsigned)level_at(q, N, K, n, k0 + 4 * wd + j, zero_col0) & 0xff) << (8 * j);
            words[wd] = wv;
        }
        v4i o = {(int)words[0], (int)words[1], (int)words[2], (int)words[3]};
        reinterpret_cast<v4i *>(out)[i] = o;
    }
}

static unsigned grid_for(long total)
{
    long b = ceil_div(total, 256);
    if (b > 4096) b = 4096;
    if (b < 1) b = 1;
    return (unsigned)b;
}

}  // namespace mq

extern "C" size_t mq_prepacked_bytes(long N, long K, int w_bits)
{
    const long N_pad = mq::ceil_div(N, w_bits == 4 ? 32 : 16) * (w_bits == 4 ? 32 : 16);
    const long K_pad = mq::ceil_div(K, 128) * 128;
    return (size_t)(N_pad * K_pad) * (size_t)w_bits / 8;
}

extern "C" int mq_pack_i4(const int8_t *q, long rows, long cols, uint8_t *out, void *stream)
{
    using namespace mq;
    MQ_REQUIRE(rows >= 0 && cols >= 0 && cols % 2 == 0, "mq_pack_i4: cols must be even");
    if (rows * cols == 0) return MQ_OK;
    hipLaunchKernelGGL(pack_i4_kernel, dim3(grid_for(rows * cols / 2)), dim3(256), 0,
                       (hipStream_t)stream, q, rows, cols, out);
    return check_launch("pack_i4");
}

extern "C" int mq_unpack_i4(const uint8_t *packed, long rows, long cols, int8_t *out, void *stream)
{
    using namespace mq;
    MQ_REQUIRE(rows >= 0 && cols >= 0 && cols % 2 == 0, "mq_unpack_i4: cols must be even");
    if (rows * cols == 0) return MQ_OK;
    hipLaunchKernelGGL(unpack_i4_kernel, dim3(grid_for(rows * cols / 2)), dim3(256), 0,
                       (hipStream_t)stream, packed, rows, cols, out);
    return check_launch("unpack_i4");
}

extern "C" int mq_weight_levels(const void *w, int w_dtype, long N, long K, long ldw,
                                const float *scale, int lo, int hi, int8_t *q, void *stream)
{
    using namespace mq;
    MQ_REQUIRE(N >= 0 && K >= 0 && ldw >= K, "mq_weight_levels: bad shape");
    MQ_REQUIRE(lo >= -128 && hi <= 127 && lo <= hi, "mq_weight_levels: level range must fit int8");
    if (N * K == 0) return MQ_OK;
    hipStream_t st = (hipStream_t)stream;
    const dim3 g(grid_for(N * K)), b(256);
    switch (w_dtype) {
    case MQ_F16: hipLaunchKernelGGL(weight_levels_kernel<MQ_F16>, g, b, 0, st, (const unsigned short *)w, N, K, ldw, scale, (float)lo, (float)hi, q); break;
    case MQ_BF16: hipLaunchKernelGGL(weight_levels_kernel<MQ_BF16>, g, b, 0, st, (const unsigned short *)w, N, K, ldw, scale, (float)lo, (float)hi, q); break;
    case MQ_F32: hipLaunchKernelGGL(weight_levels_kernel<MQ_F32>, g, b, 0, st, (const float *)w, N, K, ldw, scale, (float)lo, (float)hi, q); break;
    default: return fail(MQ_EINVAL, "mq_weight_levels: unknown dtype %d", w_dtype);
    }
    return check_launch("weight_levels");
}

extern "C" int mq_prepack_w4(const int8_t *q, long N, long K, int zero_col0, uint8_t *out, void *stream)
{
    using namespace mq;
    MQ_REQUIRE(N > 0 && K > 0 && q && out, "mq_prepack_w4: bad arguments");
    const long N_pad = ceil_div(N, 32) * 32, K_pad = ceil_div(K, 128) * 128;
    const long total = (N_pad / 32) * (K_pad / 64) * 64;
    hipLaunchKernelGGL(prepack_w4_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream,
                       q, N, K, N_pad, K_pad, zero_col0, out);
    return check_launch("prepack_w4");
}

extern "C" int mq_prepack_w8(const int8_t *q, long N, long K, int zero_col0, int8_t *out, void *stream)
{
    using namespace mq;
    MQ_REQUIRE(N > 0 && K > 0 && q && out, "mq_prepack_w8: bad arguments");
    const long N_pad = ceil_div(N, 16) * 16, K_pad = ceil_div(K, 128) * 128;
    const long total = (N_pad / 16) * (K_pad / 64) * 64;
    hipLaunchKernelGGL(prepack_w8_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream,
                       q, N, K, N_pad, K_pad, zero_col0, out);
    return check_launch("prepack_w8");
}
