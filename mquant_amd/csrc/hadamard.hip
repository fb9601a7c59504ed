// hadamard.hip -- online Hadamard rotation y = (H_K (x) H_m) [x;0] / sqrt(n), m = n/K = 2^p,
// optionally fused with the static int8 quantizer so the rotated activations never reach HBM.
//
// Reference semantics: fake_quant/utils.py:465-471 (zero pad), fake_quant/hadamard_utils.py:
// 115-128 (matmul_hadU_cuda: third-party FHT over the last m elements, then hadK @ .),
// fake_quant/quant_utils.py:334-341 (casts), and uniform.py:20-33 when quantizing.
//
// One workgroup (4 waves) owns one activation row, held in LDS as fp32:
//   A. butterflies in ascending stride, exactly the (a+b, a-b) order of the reference:
//      strides 1,2,4 inside a lane's 8 registers, strides 8..256 with wavefront shuffles
//      (lane ^ stride/8), strides >= 512 through LDS; then * 1/sqrt(n) (fp32 scalar) and the
//      cast to x's dtype that the FHT extension performs for half inputs;
//   B. the K x K +-1 stage on the matrix core: V_MFMA_F32_16X16X4_F32 is an exact k-ordered
//      fp32 fma chain (fma(+-1, y, acc) == acc +- y), so ascending k-steps reproduce the
//      oracle's sequential add/sub chain bit for bit.  hadK lives in LDS as sign bits;
//   C. cast to x's dtype, then either store or quantize (IEEE divide, rint, clamp) to int8.
#include <math.h>

#include "mq_common.h"

namespace mq {

struct HadArgs {
    const void *x;
    long M, n_in, ldx, n;
    int K, m;
    const uint8_t *had_bits;
    int fp32_had;
    void *out;
    long ldo;
    float s0, s1;
    const uint8_t *row_sel;
    int skip_col0;
    float *x0_out;
    int8_t *qout;
    long K_pad, ldq;
    int vec_ok;
    float inv_sqrt_n;  // 1.0f / sqrtf((float)n), computed on the host in IEEE fp32
};

constexpr int HAD_THREADS = 256;
constexpr int HAD_WAVES = HAD_THREADS / 64;
constexpr int HAD_G = 4;  // 16-wide column tiles accumulated per A-operand fetch

template <int DT, bool QUANT>
__device__ __forceinline__ void had_emit(const HadArgs &p, long row, long col, float v, float s)
{
    v = Elem<DT>::rnd(v);
    if (QUANT) {
        int q;
        if (p.skip_col0 && col == 0) {
            if (p.x0_out) p.x0_out[row] = v;
            q = 0;
        } else {
            q = quant_level(v, s, -128.0f, 127.0f);
        }
        p.qout[row * p.ldq + col] = (int8_t)q;
    } else {
        typedef typename Elem<DT>::T T;
        reinterpret_cast<T *>(p.out)[row * p.ldo + col] = Elem<DT>::st(v);
    }
}

template <int DT, bool QUANT>
__global__ __launch_bounds__(HAD_THREADS) void hadamard_kernel(HadArgs p)
{
    typedef typename Elem<DT>::T T;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float *y = reinterpret_cast<float *>(smem);
    const uint8_t *hb = reinterpret_cast<const uint8_t *>(smem + (size_t)p.n * 4);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const long n = p.n;
    const int K = p.K, m = p.m;
    const float scale = p.inv_sqrt_n;
    const bool mid_round = (DT != MQ_F32) && !p.fp32_had;

    if (K > 1) {
        uint8_t *hbw = reinterpret_cast<uint8_t *>(smem + (size_t)n * 4);
        const int nbytes = (K * K + 7) / 8;
        for (int i = tid; i < nbytes; i += HAD_THREADS) hbw[i] = p.had_bits[i];
    }

    for (long row = blockIdx.x; row < p.M; row += gridDim.x) {
        const T *xr = reinterpret_cast<const T *>(p.x) + row * p.ldx;
        const float s = (p.row_sel && p.row_sel[row]) ? p.s1 : p.s0;

        // ---------------- A: butterflies ------------------------------------------------
        if (m >= 8) {
            const long nchunks = ceil_div(n, 512);
            for (long c = wave; c < nchunks; c += HAD_WAVES) {
                const long idx = c * 512 + lane * 8;
                float v[8];
                if (idx + 8 <= p.n_in && p.vec_ok) {
                    if (sizeof(T) == 2) {
                        const v8us a = *reinterpret_cast<const v8us *>(xr + idx);
#pragma unroll
                        for (int i = 0; i < 8; ++i) v[i] = Elem<DT>::ld((T)a[i]);
                    } else {
                        const v4f a = *reinterpret_cast<const v4f *>((const float *)xr + idx);
                        const v4f b = *reinterpret_cast<const v4f *>((const float *)xr + idx + 4);
#pragma unroll
                        for (int i = 0; i < 4; ++i) { v[i] = a[i]; v[4 + i] = b[i]; }
                    }
                } else {
#pragma unroll
                    for (int i = 0; i < 8; ++i) v[i] = (idx + i < p.n_in) ? Elem<DT>::ld(xr[idx + i]) : 0.0f;
                }
#pragma unroll
                for (int h = 1; h < 8; h <<= 1) {
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        if ((i & h) == 0) {
                            const float a0 = v[i], a1 = v[i + h];
                            v[i] = a0 + a1;
                            v[i + h] = a0 - a1;
                        }
                    }
                }
                for (int h = 8; h < m && h < 512; h <<= 1) {
                    const int lm = h >> 3;
                    const bool upper = (lane & lm) != 0;
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        const float o = __shfl_xor(v[i], lm);
                        v[i] = upper ? (o - v[i]) : (v[i] + o);
                    }
                }
                if (m <= 512) {
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        float t = v[i] * scale;
                        if (mid_round) t = Elem<DT>::rnd(t);
                        v[i] = t;
                    }
                }
                if (idx + 8 <= n) {
                    *reinterpret_cast<v4f *>(y + idx) = v4f{v[0], v[1], v[2], v[3]};
                    *reinterpret_cast<v4f *>(y + idx + 4) = v4f{v[4], v[5], v[6], v[7]};
                } else {
#pragma unroll
                    for (int i = 0; i < 8; ++i)
                        if (idx + i < n) y[idx + i] = v[i];
                }
            }
        } else {
            for (long i = tid; i < n; i += HAD_THREADS) y[i] = (i < p.n_in) ? Elem<DT>::ld(xr[i]) : 0.0f;
        }
        __syncthreads();
        if (m > 512 || m < 8) {
            for (long h = (m < 8) ? 1 : 512; h < m; h <<= 1) {
                for (long b = tid; b < n / 2; b += HAD_THREADS) {
                    const long i = (b / h) * 2 * h + (b % h);
                    const float a0 = y[i], a1 = y[i + h];
                    y[i] = a0 + a1;
                    y[i + h] = a0 - a1;
                }
                __syncthreads();
            }
            for (long i = tid; i < n; i += HAD_THREADS) {
                float t = y[i] * scale;
                if (mid_round) t = Elem<DT>::rnd(t);
                y[i] = t;
            }
            __syncthreads();
        }

        // ---------------- B/C: K x K stage, cast, store / quantize -----------------------
        if (K == 1) {
            for (long idx = (long)tid * 8; idx < n; idx += HAD_THREADS * 8) {
#pragma unroll
                for (int i = 0; i < 8; ++i)
                    if (idx + i < n) had_emit<DT, QUANT>(p, row, idx + i, y[idx + i], s);
            }
        } else if (m >= 16) {
            const int JT = (K + 15) / 16;
            const int IT = m / 16;
            const int IG = (IT + HAD_G - 1) / HAD_G;
            const int ksteps = K / 4;
            for (int u = wave; u < JT * IG; u += HAD_WAVES) {
                const int jt = u / IG, ig = u - jt * IG;
                v4f acc[HAD_G];
#pragma unroll
                for (int g = 0; g < HAD_G; ++g) acc[g] = v4f{0.f, 0.f, 0.f, 0.f};
                const int ja = jt * 16 + (lane & 15);
                const bool jvalid = ja < K;
                int bit = ja * K + (lane >> 4);
                const float *yb = y + (long)(lane >> 4) * m + ig * (HAD_G * 16) + (lane & 15);
                for (int ks = 0; ks < ksteps; ++ks) {
                    float a = 0.0f;
                    if (jvalid) {
                        const unsigned byte = hb[bit >> 3];
                        a = ((byte >> (7 - (bit & 7))) & 1u) ? 1.0f : -1.0f;
                    }
                    bit += 4;
                    const float *yk = yb + (long)ks * 4 * m;
#pragma unroll
                    for (int g = 0; g < HAD_G; ++g) {
                        if (ig * HAD_G + g < IT) {
                            const float b = yk[g * 16];
                            acc[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[g], 0, 0, 0);
                        }
                    }
                }
#pragma unroll
                for (int g = 0; g < HAD_G; ++g) {
                    const int it = ig * HAD_G + g;
                    if (it >= IT) continue;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int j = jt * 16 + (lane >> 4) * 4 + r;
                        if (j < K) had_emit<DT, QUANT>(p, row, (long)j * m + it * 16 + (lane & 15), acc[g][r], s);
                    }
                }
            }
        } else {
            for (long o = tid; o < n; o += HAD_THREADS) {
                const int j = (int)(o / m), i = (int)(o - (long)j * m);
                float acc = 0.0f;
                for (int k = 0; k < K; ++k) {
                    const int bit = j * K + k;
                    const float v = y[(long)k * m + i];
                    acc = ((hb[bit >> 3] >> (7 - (bit & 7))) & 1u) ? (acc + v) : (acc - v);
                }
                had_emit<DT, QUANT>(p, row, o, acc, s);
            }
        }
        if (QUANT) {
            for (long c = n + tid; c < p.K_pad; c += HAD_THREADS) p.qout[row * p.ldq + c] = 0;
        }
        __syncthreads();  // y is reused by the next row
    }
}

template <int DT, bool QUANT>
static int launch_hadamard(const HadArgs &p, hipStream_t st)
{
    const size_t lds = (size_t)p.n * 4 + (((size_t)p.K * p.K + 7) / 8 + 15) / 16 * 16;
    if (lds > 160 * 1024) return fail(MQ_EUNSUPPORTED, "mq_hadamard: n=%ld needs %zu B of LDS (> 160 KiB)", p.n, lds);
    auto kern = hadamard_kernel<DT, QUANT>;
    hipError_t e = hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return fail((int)e, "hadamard: set smem attr: %s", hipGetErrorString(e));
    int cus = 256;
    long per_cu = (160 * 1024) / (long)lds;
    if (per_cu > 8) per_cu = 8;
    if (per_cu < 1) per_cu = 1;
    long blocks = (long)cus * per_cu;
    if (blocks > p.M) blocks = p.M;
    hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(HAD_THREADS), lds, st, p);
    return check_launch("hadamard");
}

static int hadamard_common(HadArgs p, int x_dtype, bool quant, void *stream)
{
    MQ_REQUIRE(p.M >= 0 && p.n > 0 && p.n_in > 0 && p.n_in <= p.n, "mq_hadamard: bad shape (n_in=%ld, n=%ld)", p.n_in, p.n);
    if (p.M == 0) return MQ_OK;
    MQ_REQUIRE(p.K >= 1 && p.n % p.K == 0, "mq_hadamard: K=%d does not divide n=%ld", p.K, p.n);
    p.m = (int)(p.n / p.K);
    p.inv_sqrt_n = 1.0f / sqrtf((float)p.n);
    MQ_REQUIRE((p.m & (p.m - 1)) == 0, "mq_hadamard: n/K=%d is not a power of two", p.m);
    MQ_REQUIRE(p.K == 1 || (p.K % 4 == 0 && p.had_bits), "mq_hadamard: K=%d needs had_bits and K %% 4 == 0", p.K);
    MQ_REQUIRE(p.ldx >= p.n_in, "mq_hadamard: ldx < n_in");
    const size_t esz = (x_dtype == MQ_F32) ? 4 : 2;
    p.vec_ok = (((uintptr_t)p.x) % 16 == 0) && ((p.ldx * esz) % 16 == 0);
    hipStream_t st = (hipStream_t)stream;
    if (quant) {
        MQ_REQUIRE(p.qout && p.K_pad >= p.n && p.ldq >= p.K_pad, "mq_hadamard_quant_i8: bad output geometry");
        switch (x_dtype) {
        case MQ_F16: return launch_hadamard<MQ_F16, true>(p, st);
        case MQ_BF16: return launch_hadamard<MQ_BF16, true>(p, st);
        case MQ_F32: return launch_hadamard<MQ_F32, true>(p, st);
        }
    } else {
        MQ_REQUIRE(p.out && p.ldo >= p.n, "mq_hadamard: bad output geometry");
        switch (x_dtype) {
        case MQ_F16: return launch_hadamard<MQ_F16, false>(p, st);
        case MQ_BF16: return launch_hadamard<MQ_BF16, false>(p, st);
        case MQ_F32: return launch_hadamard<MQ_F32, false>(p, st);
        }
    }
    return fail(MQ_EINVAL, "mq_hadamard: unknown dtype %d", x_dtype);
}

}  // namespace mq

extern "C" int mq_hadamard(const void *x, int x_dtype, long M, long n_in, long ldx, long n, int K,
                           const uint8_t *had_bits, int fp32_had, void *out, long ldo, void *stream)
{
    mq::HadArgs p;
    memset(&p, 0, sizeof(p));
    p.x = x; p.M = M; p.n_in = n_in; p.ldx = ldx; p.n = n; p.K = K; p.had_bits = had_bits;
    p.fp32_had = fp32_had; p.out = out; p.ldo = ldo; p.s0 = p.s1 = 1.0f;
    return mq::hadamard_common(p, x_dtype, false, stream);
}

extern "C" int mq_hadamard_quant_i8(const void *x, int x_dtype, long M, long n_in, long ldx, long n,
                                    int K, const uint8_t *had_bits, int fp32_had, float scale0,
                                    float scale1, const uint8_t *row_sel, int skip_col0,
                                    float *x0_out, int8_t *out, long K_pad, long ldo, void *stream)
{
    mq::HadArgs p;
    memset(&p, 0, sizeof(p));
    p.x = x; p.M = M; p.n_in = n_in; p.ldx = ldx; p.n = n; p.K = K; p.had_bits = had_bits;
    p.fp32_had = fp32_had; p.s0 = scale0; p.s1 = scale1; p.row_sel = row_sel;
    p.skip_col0 = skip_col0; p.x0_out = x0_out; p.qout = out; p.K_pad = K_pad; p.ldq = ldo;
    return mq::hadamard_common(p, x_dtype, true, stream);
}
