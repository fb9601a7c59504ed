// wquant.hip -- the weight quantizer on the device:
//   WeightQuantizer.find_params + quantize, symmetric, per output channel
//   (reference fake_quant/quant_utils.py:446-518, sym_quant_dequant :46-58),
// emitting in ONE launch the per-channel scale, the integer levels, the int4 wire format
// (reference pack_i4, quant_utils.py:61-69) and/or the fake-quantized weights W~ in W's dtype.
//
// Arithmetic = the reference's on ANY weight dtype: a half weight tensor is promoted by the fp32
// `tmp` it is compared with (:458-460), so min/max, scales, the clip-search errors and s*q are
// fp32 on the exactly converted values; only W~ is cast back.
//
// Clip search (mse): candidate i shrinks the range by p = 1 - i/grid, error = sum_k |s1*q - x|^norm,
// first strict minimum wins.  One THREAD per (channel, candidate) walks the row in ascending k
// with a plain fp32 add chain, so the error -- and therefore the argmin -- is reproducible bit for
// bit (oracle: orc_wquant_sym).  d^norm is evaluated in double with ordered +,*,/ only and rounded
// once to fp32 (orc_pow_pos is the same code): within 1 ulp of torch's fp32 pow, and independent of
// any math library.  Candidates of one channel sit in adjacent lanes and read the same address
// (one broadcast transaction per element).
#include "mq_common.h"

namespace mq {

__device__ __forceinline__ float pow_pos(float d, float norm)
{
    if (d == 0.0f) return 0.0f;
    return (float)exp2_d((double)norm * log2_pos((double)d));
}

struct WqArgs {
    const void *w;
    long N, K, ldw;
    float maxq;          // 2^(bits-1) - 1
    int mse, grid, steps;
    float norm;
    float *scale;
    int8_t *levels;
    uint8_t *packed;
    void *wq;
    long ldq;
};

constexpr int WQ_THREADS = 256;

// CPB channels per workgroup, TPC = 256 / CPB threads each.  Dynamic LDS: float err[CPB][steps].
template <int DT, int CPB>
__global__ __launch_bounds__(WQ_THREADS) void wquant_sym_kernel(WqArgs p)
{
    typedef typename Elem<DT>::T T;
    constexpr int TPC = WQ_THREADS / CPB;
    extern __shared__ float errs[];
    __shared__ float red_min[WQ_THREADS], red_max[WQ_THREADS];
    __shared__ float s_xmax[CPB], s_scale[CPB];

    const int c = threadIdx.x / TPC, j = threadIdx.x - c * TPC;
    const long n = (long)blockIdx.x * CPB + c;
    const bool live = c < CPB && n < p.N;
    const T *row = reinterpret_cast<const T *>(p.w) + (live ? n : 0) * p.ldw;
    const float lo = -(p.maxq + 1.0f), hi = p.maxq;

    // ---- min / max including 0 (:458-460), |.|, clamp 1e-5 (:463) ---------------------------
    float mn = 0.0f, mx = 0.0f;
    if (live)
        for (long k = j; k < p.K; k += TPC) {
            const float v = Elem<DT>::ld(row[k]);
            mn = fminf(mn, v);
            mx = fmaxf(mx, v);
        }
    red_min[threadIdx.x] = mn;
    red_max[threadIdx.x] = mx;
    __syncthreads();
    if (live && j == 0) {
        for (int t = 1; t < TPC; ++t) {
            mn = fminf(mn, red_min[threadIdx.x + t]);
            mx = fmaxf(mx, red_max[threadIdx.x + t]);
        }
        float xmax = fmaxf(fabsf(mn), mx);
        xmax = fmaxf(xmax, 1e-5f);
        s_xmax[c] = xmax;
        s_scale[c] = xmax / p.maxq;
    }
    __syncthreads();

    // ---- clip search (:468-497) -----------------------------------------------------------------
    if (p.mse) {
        if (live) {
            const float xmax = s_xmax[c];
            for (int i = j; i < p.steps; i += TPC) {
                const float pf = (float)(1.0 - (double)i / (double)p.grid);
                const float s1 = (pf * xmax) / p.maxq;
                float err = 0.0f;
                for (long k = 0; k < p.K; ++k) {
                    const float x = Elem<DT>::ld(row[k]);
                    float q = rintf(x / s1);
                    q = fminf(fmaxf(q, lo), hi);
                    const float d = fabsf(s1 * q - x);
                    err = err + pow_pos(d, p.norm);
                }
                errs[c * p.steps + i] = err;
            }
        }
        __syncthreads();
        if (live && j == 0) {
            const float xmax = s_xmax[c];
            float best = INFINITY, s = s_scale[c];
            for (int i = 0; i < p.steps; ++i) {
                const float e = errs[c * p.steps + i];
                if (e < best) {
                    best = e;
                    s = ((float)(1.0 - (double)i / (double)p.grid) * xmax) / p.maxq;
                }
            }
            s_scale[c] = s;
        }
        __syncthreads();
    }
    if (!live) return;
    const float s = s_scale[c];
    if (j == 0) p.scale[n] = s;

    // ---- levels, wire format, W~ (:504-509; pack_i4 :61-69) ----------------------------------------
    T *wq_row = p.wq ? reinterpret_cast<T *>(p.wq) + n * p.ldq : nullptr;
    const long pairs = p.K / 2;
    for (long kp = j; kp < pairs; kp += TPC) {
        const int a = quant_level(Elem<DT>::ld(row[2 * kp]), s, lo, hi);
        const int b = quant_level(Elem<DT>::ld(row[2 * kp + 1]), s, lo, hi);
        if (p.levels) {
            p.levels[n * p.K + 2 * kp] = (int8_t)a;
            p.levels[n * p.K + 2 * kp + 1] = (int8_t)b;
        }
        if (p.packed) p.packed[n * pairs + kp] = (uint8_t)(((unsigned)a & 0xf) | (((unsigned)b & 0xf) << 4));
        if (wq_row) {
            wq_row[2 * kp] = Elem<DT>::st(s * (float)a);
            wq_row[2 * kp + 1] = Elem<DT>::st(s * (float)b);
        }
    }
    if ((p.K & 1) && j == 0) {
        const long k = p.K - 1;
        const int a = quant_level(Elem<DT>::ld(row[k]), s, lo, hi);
        if (p.levels) p.levels[n * p.K + k] = (int8_t)a;
        if (wq_row) wq_row[k] = Elem<DT>::st(s * (float)a);
    }
}

template <int DT>
static int launch_wquant(const WqArgs &p, hipStream_t st)
{
    if (p.mse) {
        // 3 channels x 85 threads: 80 candidates (the default grid) keep 240 of 256 lanes busy
        constexpr int CPB = 3;
        const size_t smem = (size_t)CPB * p.steps * sizeof(float);
        hipLaunchKernelGGL((wquant_sym_kernel<DT, CPB>), dim3((unsigned)ceil_div(p.N, CPB)), dim3(WQ_THREADS),
                           smem, st, p);
    } else {
        constexpr int CPB = 4;
        hipLaunchKernelGGL((wquant_sym_kernel<DT, CPB>), dim3((unsigned)ceil_div(p.N, CPB)), dim3(WQ_THREADS),
                           0, st, p);
    }
    return check_launch("wquant_sym");
}

}  // namespace mq

extern "C" int mq_wquant_sym(const void *w, int w_dtype, long N, long K, long ldw, int bits, int mse,
                             float norm, int grid, float maxshrink, float *scale, int8_t *levels,
                             uint8_t *packed, void *wq, long ldq, void *stream)
{
    using namespace mq;
    if (N == 0) return MQ_OK;                       // empty input: nothing to do (null pointers allowed)
    MQ_REQUIRE(w && scale && N >= 0 && K > 0 && ldw >= K, "mq_wquant_sym: bad shape");
    MQ_REQUIRE(bits >= 2 && bits <= 8, "mq_wquant_sym: bits must be 2..8 (got %d)", bits);
    MQ_REQUIRE(!packed || (bits == 4 && K % 2 == 0), "mq_wquant_sym: the int4 wire format needs bits == 4 and an even K");
    MQ_REQUIRE(!wq || ldq >= K, "mq_wquant_sym: ldq < K");
    if (N == 0) return MQ_OK;
    WqArgs p;
    p.w = w; p.N = N; p.K = K; p.ldw = ldw;
    p.maxq = (float)((1 << (bits - 1)) - 1);
    p.mse = mse ? 1 : 0;
    p.grid = grid;
    p.steps = mse ? (int)(maxshrink * (float)grid) : 0;
    p.norm = norm;
    p.scale = scale; p.levels = levels; p.packed = packed; p.wq = wq; p.ldq = ldq;
    if (mse) MQ_REQUIRE(grid > 0 && p.steps >= 1 && p.steps <= 4096, "mq_wquant_sym: grid/maxshrink give %d candidates (1..4096)", p.steps);
    hipStream_t st = (hipStream_t)stream;
    switch (w_dtype) {
    case MQ_F16: return launch_wquant<MQ_F16>(p, st);
    case MQ_BF16: return launch_wquant<MQ_BF16>(p, st);
    case MQ_F32: return launch_wquant<MQ_F32>(p, st);
    default: return fail(MQ_EINVAL, "mq_wquant_sym: unknown dtype %d", w_dtype);
    }
}
