// rotate_f64.hip -- offline rotation of weight rows by Q = diag(s) . (H_K (x) H_{n/K}) / sqrt(n) in
// fp64.  The reference builds Q as a dense fp64 matrix (hadamard_utils.py:107-112:
// random_hadamard_matrix = matmul_hadU(diag(+-1))) and evaluates W <- (W.double() @ Q).to(dtype) for
// every Linear of the model (rotation_utils.py / qwen2vl_rotation.py:232-332,
// internvl_rotation.py:223-303): 2 n flops per element.  Here a row is read once, sign-flipped,
// transformed in LDS (log2(n/K) butterfly stages + the K x K sign stage, K + log2(n/K) flops per
// element), scaled and written back in its own dtype: HBM-bound, one read + one write of the weight.
//     y[k'*P + j'] = ( sum_k hadK[k'][k] * sum_j H_P[j'][j] * s[k*P + j] * x[k*P + j] ) / c
// with P = n / K and c = (double)sqrtf(n) (the reference divides by torch.tensor(n).sqrt(), an fp32
// scalar).  Q^T W is the same transform of the rows of W^T (the caller transposes).
#include "mq_common.h"

namespace mq {

struct RotArgs {
    void *x;
    long M, ld;
    int n, K, P, wpr;
    const double *signs;
    const uint32_t *words;
    double c;
};

// torch's double -> half / bfloat16 cast goes through float (static_cast<float> first): two roundings
template <int DT> __device__ __forceinline__ double ld64(const void *p, long i)
{
    if constexpr (DT == MQ_F64) return reinterpret_cast<const double *>(p)[i];
    else return (double)Elem<DT>::ld(reinterpret_cast<const typename Elem<DT>::T *>(p)[i]);
}
template <int DT> __device__ __forceinline__ void st64(void *p, long i, double v)
{
    if constexpr (DT == MQ_F64) reinterpret_cast<double *>(p)[i] = v;
    else reinterpret_cast<typename Elem<DT>::T *>(p)[i] = Elem<DT>::st((float)v);
}

template <int DT>
__global__ __launch_bounds__(256) void rotate_f64_kernel(RotArgs p)
{
    extern __shared__ __attribute__((aligned(16))) double row[];
    const int tid = threadIdx.x, n = p.n, P = p.P;
    for (long r = blockIdx.x; r < p.M; r += gridDim.x) {
        const long base = r * p.ld;
        for (int i = tid; i < n; i += 256) {
            double v = ld64<DT>(p.x, base + i);
            if (p.signs) v = v * p.signs[i];
            row[i] = v;
        }
        __syncthreads();
        // Walsh-Hadamard over j (the minor index), strides ascending like the reference's loop
        for (int h = 1; h < P; h <<= 1) {
            for (int t = tid; t < n / 2; t += 256) {
                const int lo = t & (h - 1);
                const int i0 = ((t - lo) << 1) + lo;       // k*P is a multiple of 2h: the pair stays inside its block
                const double a = row[i0], b = row[i0 + h];
                row[i0] = a + b;
                row[i0 + h] = a - b;
            }
            __syncthreads();
        }
        // K x K sign stage over k (the major index), ascending k
        if (p.K > 1) {
            for (int o = tid; o < n; o += 256) {
                const int ko = o / P, j = o - ko * P;
                const uint32_t *w = p.words + (long)ko * p.wpr;
                double acc = 0.0;
                for (int k0 = 0; k0 < p.K; k0 += 32) {
                    const uint32_t bits = w[k0 >> 5];
                    const int kn = (p.K - k0) < 32 ? (p.K - k0) : 32;
                    for (int b = 0; b < kn; ++b) {
                        const double v = row[(k0 + b) * P + j];
                        acc = ((bits >> b) & 1u) ? acc + v : acc - v;
                    }
                }
                st64<DT>(p.x, base + o, acc / p.c);
            }
        } else {
            for (int o = tid; o < n; o += 256) st64<DT>(p.x, base + o, row[o] / p.c);
        }
        __syncthreads();
    }
}

template <int DT> static int launch_rot(RotArgs &a, hipStream_t st)
{
    const int bytes = a.n * (int)sizeof(double);
    const int rc = ensure_dynamic_lds(reinterpret_cast<const void *>(&rotate_f64_kernel<DT>), bytes);
    if (rc != MQ_OK) return rc;
    const long blocks = a.M < (1L << 20) ? a.M : (1L << 20);
    hipLaunchKernelGGL(rotate_f64_kernel<DT>, dim3((unsigned)blocks), dim3(256), bytes, st, a);
    return check_launch("rotate_f64_kernel");
}

}  // namespace mq

extern "C" int mq_rotate_f64(void *x, int dtype, long M, long n, long ld, const double *signs,
                             int K, const uint32_t *had_words, void *stream)
{
    using namespace mq;
    MQ_REQUIRE(x != nullptr && M >= 0 && n >= 1 && ld >= n, "mq_rotate_f64: bad shape M=%ld n=%ld ld=%ld", M, n, ld);
    MQ_REQUIRE(K >= 1 && n % K == 0, "mq_rotate_f64: n=%ld is not a multiple of K=%d", n, K);
    const long P = n / K;
    MQ_REQUIRE((P & (P - 1)) == 0, "mq_rotate_f64: n / K = %ld is not a power of two", P);
    MQ_REQUIRE(K == 1 || had_words != nullptr, "mq_rotate_f64: K=%d needs had_words", K);
    MQ_REQUIRE(n * 8 <= 160 * 1024, "mq_rotate_f64: a row of n=%ld doubles does not fit the 160 KiB LDS", n);
    if (M == 0) return MQ_OK;
    RotArgs a;
    a.x = x; a.M = M; a.ld = ld; a.n = (int)n; a.K = K; a.P = (int)P; a.wpr = (K + 31) / 32;
    a.signs = signs; a.words = had_words;
    a.c = (double)sqrtf((float)n);
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    switch (dtype) {
    case MQ_F16: return launch_rot<MQ_F16>(a, st);
    case MQ_BF16: return launch_rot<MQ_BF16>(a, st);
    case MQ_F32: return launch_rot<MQ_F32>(a, st);
    case MQ_F64: return launch_rot<MQ_F64>(a, st);
    default: return fail(MQ_EINVAL, "mq_rotate_f64: dtype %d", dtype);
    }
}
