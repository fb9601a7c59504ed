// gemm_w4a8.hip -- int8 activations x int4 (or int8) weights -> int32 -> fused dequant.
//
//   acc[m][n] = sum_k a[m][k] * w[n][k]        V_MFMA_I32_16X16X64_I8, exact
//   y[m][n]   = ((float(acc) * s_x[sel(m)]) * s_w[n]) + bias[n] + x0[m] * w0[n]
//
// Structure (one workgroup = BM x BN output tile, K walked in steps of 128):
//   * both operands arrive by LDS-DMA (global_load_lds, 16 B per lane); the LDS image of
//     every 16x64 fragment is lane-linear (lane l at byte 16*l), so fragment reads are
//     conflict-free ds_read_b128 with no swizzle:
//       - weights are stored pre-tiled in HBM in exactly that order (weight_formats.hip);
//       - activations are row-major in HBM, the per-lane SOURCE address does the
//         re-tiling (row = m0 + (l & 15), bytes 16*(l >> 4) .. +16 of the 64-wide k-tile).
//   * int4 weights stay packed in LDS; each wave expands the nibbles it is about to feed to
//     the matrix core into the HIGH nibble of int8 bytes (two VALU ops per 8 weights, no
//     sign-extension needed); the resulting x16 factor is removed by an exact arithmetic
//     shift before dequantisation.
//   * the weight fragment is the MFMA "A" operand and the activation fragment the "B"
//     operand, so every lane ends up with 4 consecutive output channels of one row: the
//     epilogue reads s_w / bias / w0 as float4 and stores 8 B (fp16) per lane.
//   * 2-stage LDS ring: the DMA of step t+1 is issued right after the barrier that
//     publishes step t and flies under the MFMAs of step t.
//
// Reference semantics: fake_quant/quant_utils.py:384 (F.linear on fake-quant tensors) and
// :367-376 (split: channel 0 through L1 in fp32).
#include "mq_common.h"

namespace mq {

typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void gbl_void;

__device__ __forceinline__ void dma16(const void *g, void *lds_wave_base)
{
    __builtin_amdgcn_global_load_lds((gbl_void *)g, (lds_void *)lds_wave_base, 16, 0, 0);
}

enum { EPI_F16 = MQ_F16, EPI_BF16 = MQ_BF16, EPI_F32 = MQ_F32, EPI_I32 = 3 };

struct GemmArgs {
    const int8_t *a;
    long lda;
    const uint8_t *w;
    long M, N, K_pad;
    long n_tiles;  // N_pad / 16
    float sx0, sx1;
    const uint8_t *row_sel;
    const float *s_w, *bias, *x0, *w0;
    void *out;
    long ldo;
};

template <int BM, int BN, int WARPS_M, int WARPS_N, int W_BITS, int EPI>
__global__ __launch_bounds__(WARPS_M *WARPS_N * 64) void gemm_w4a8_kernel(GemmArgs p)
{
    constexpr int NWAVES = WARPS_M * WARPS_N;
    constexpr int TM = BM / WARPS_M / 16;          // activation fragments per wave
    constexpr int TN = BN / WARPS_N / 16;          // weight fragments per wave
    constexpr int X_FRAGS = (BM / 16) * 2;         // 1 KiB DMA pieces per stage (2 k-tiles)
    constexpr int W_PIECES = (W_BITS == 4) ? (BN / 16) : (BN / 16) * 2;
    constexpr int X_BYTES = X_FRAGS * 1024;
    constexpr int STAGE_BYTES = X_BYTES + W_PIECES * 1024;
    static_assert(BM % (WARPS_M * 16) == 0 && BN % (WARPS_N * 16) == 0, "tile shape");

    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WARPS_N, wn = wave % WARPS_N;

    // block -> tile: m-blocks fastest so the blocks sharing one weight panel run together
    const int m_blocks = (int)ceil_div(p.M, BM);
    const int bm = blockIdx.x % m_blocks;
    const int bn = blockIdx.x / m_blocks;
    const long m0 = (long)bm * BM;
    const long nt0 = (long)bn * (BN / 16);

    const long kps = p.K_pad / 128;  // k-steps

    // ---- per-lane DMA source addresses ----------------------------------------------
    // activations: piece f = mt*2 + kt handled by wave (f % NWAVES)
    constexpr int X_PER_WAVE = (X_FRAGS + NWAVES - 1) / NWAVES;
    constexpr int W_PER_WAVE = (W_PIECES + NWAVES - 1) / NWAVES;
    const int8_t *xsrc[X_PER_WAVE];
#pragma unroll
    for (int i = 0; i < X_PER_WAVE; ++i) {
        const int f = wave + i * NWAVES;
        const int mt = f >> 1, kt = f & 1;
        long row = m0 + mt * 16 + (lane & 15);
        if (row >= p.M) row = p.M - 1;
        xsrc[i] = p.a + row * p.lda + kt * 64 + (lane >> 4) * 16;
    }
    const uint8_t *wsrc[W_PER_WAVE];
#pragma unroll
    for (int i = 0; i < W_PER_WAVE; ++i) {
        const int f = wave + i * NWAVES;
        if (W_BITS == 4) {
            long nt = nt0 + f;
            if (nt >= p.n_tiles) nt = p.n_tiles - 1;
            wsrc[i] = p.w + ((nt * kps) * 64 + lane) * 16;            // + kp*1024 per step
        } else {
            long nt = nt0 + (f >> 1);
            if (nt >= p.n_tiles) nt = p.n_tiles - 1;
            wsrc[i] = p.w + ((nt * kps * 2 + (f & 1)) * 64 + lane) * 16;  // + kp*2048 per step
        }
    }

    auto issue_stage = [&](int stage, long kp) {
        char *base = smem + stage * STAGE_BYTES;
#pragma unroll
        for (int i = 0; i < X_PER_WAVE; ++i) {
            const int f = wave + i * NWAVES;
            if (X_FRAGS % NWAVES == 0 || f < X_FRAGS) dma16(xsrc[i] + kp * 128, base + f * 1024);
        }
#pragma unroll
        for (int i = 0; i < W_PER_WAVE; ++i) {
            const int f = wave + i * NWAVES;
            if (W_PIECES % NWAVES == 0 || f < W_PIECES)
                dma16(wsrc[i] + kp * (W_BITS == 4 ? 1024 : 2048), base + X_BYTES + f * 1024);
        }
    };

    v4i acc[TN][TM];
#pragma unroll
    for (int i = 0; i < TN; ++i)
#pragma unroll
        for (int j = 0; j < TM; ++j) acc[i][j] = v4i{0, 0, 0, 0};

    issue_stage(0, 0);
    for (long kp = 0; kp < kps; ++kp) {
        const int cur = (int)(kp & 1);
        __syncthreads();  // hipcc drains vmcnt(0) ahead of the barrier: stage `cur` has landed
        if (kp + 1 < kps) issue_stage(cur ^ 1, kp + 1);

        const char *xs = smem + cur * STAGE_BYTES;
        const char *ws = xs + X_BYTES;
#pragma unroll
        for (int kt = 0; kt < 2; ++kt) {
            v4i xf[TM];
#pragma unroll
            for (int j = 0; j < TM; ++j) {
                const int mt = wm * TM + j;
                xf[j] = *reinterpret_cast<const v4i *>(xs + (mt * 2 + kt) * 1024 + lane * 16);
            }
#pragma unroll
            for (int i = 0; i < TN; ++i) {
                const int nt = wn * TN + i;
                v4i wf;
                if (W_BITS == 4) {
                    const v2i pk = *reinterpret_cast<const v2i *>(ws + nt * 1024 + lane * 16 + kt * 8);
                    wf[0] = (pk[0] << 4) & 0xF0F0F0F0;
                    wf[1] = pk[0] & 0xF0F0F0F0;
                    wf[2] = (pk[1] << 4) & 0xF0F0F0F0;
                    wf[3] = pk[1] & 0xF0F0F0F0;
                } else {
                    wf = *reinterpret_cast<const v4i *>(ws + (nt * 2 + kt) * 1024 + lane * 16);
                }
#pragma unroll
                for (int j = 0; j < TM; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_i32_16x16x64_i8(wf, xf[j], acc[i][j], 0, 0, 0);
            }
        }
    }

    // ---- epilogue ---------------------------------------------------------------------
    // D layout: col = lane & 15 -> m, row = (lane >> 4) * 4 + r -> n
    const int ml = lane & 15, nq = (lane >> 4) * 4;
#pragma unroll
    for (int j = 0; j < TM; ++j) {
        const long m = m0 + (wm * TM + j) * 16 + ml;
        if (m >= p.M) continue;
        float sx = p.sx0, xz = 0.0f;
        if (EPI != EPI_I32) {
            if (p.row_sel && p.row_sel[m]) sx = p.sx1;
            if (p.x0) xz = p.x0[m];
        }
#pragma unroll
        for (int i = 0; i < TN; ++i) {
            const long n = (nt0 + wn * TN + i) * 16 + nq;
            if (n >= p.N) continue;
            v4i a = acc[i][j];
            if (W_BITS == 4) {
#pragma unroll
                for (int r = 0; r < 4; ++r) a[r] >>= 4;
            }
            const bool full = (n + 4 <= p.N);
            if (EPI == EPI_I32) {
                int *o = reinterpret_cast<int *>(p.out) + m * p.ldo + n;
                if (full && (p.ldo % 4 == 0)) {
                    *reinterpret_cast<v4i *>(o) = a;
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (n + r < p.N) o[r] = a[r];
                }
            } else {
                float y[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const long nn = (n + r < p.N) ? n + r : p.N - 1;
                    float t = (float)a[r] * sx;
                    t = t * p.s_w[nn];
                    if (p.bias) t = t + p.bias[nn];
                    if (p.x0) {
                        const float pr = xz * p.w0[nn];
                        t = t + pr;
                    }
                    y[r] = t;
                }
                if (EPI == EPI_F32) {
                    float *o = reinterpret_cast<float *>(p.out) + m * p.ldo + n;
                    if (full && (p.ldo % 4 == 0)) {
                        *reinterpret_cast<v4f *>(o) = v4f{y[0], y[1], y[2], y[3]};
                    } else {
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                            if (n + r < p.N) o[r] = y[r];
                    }
                } else {
                    unsigned short h[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        h[r] = (EPI == EPI_F16) ? f32_to_f16_bits(y[r]) : f32_to_bf16_bits(y[r]);
                    unsigned short *o = reinterpret_cast<unsigned short *>(p.out) + m * p.ldo + n;
                    if (full && (p.ldo % 4 == 0)) {
                        *reinterpret_cast<v4us *>(o) = v4us{h[0], h[1], h[2], h[3]};
                    } else {
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                            if (n + r < p.N) o[r] = h[r];
                    }
                }
            }
        }
    }
}

template <int BM, int BN, int WARPS_M, int WARPS_N, int W_BITS, int EPI>
static int launch_gemm(const GemmArgs &p, hipStream_t st)
{
    constexpr int X_BYTES = (BM / 16) * 2 * 1024;
    constexpr int W_BYTES = ((W_BITS == 4) ? (BN / 16) : (BN / 16) * 2) * 1024;
    constexpr int SMEM = 2 * (X_BYTES + W_BYTES);
    auto kern = gemm_w4a8_kernel<BM, BN, WARPS_M, WARPS_N, W_BITS, EPI>;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void *)kern,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, SMEM);
        if (e != hipSuccess) return fail((int)e, "gemm: set smem attr: %s", hipGetErrorString(e));
        attr_set = true;
    }
    const long m_blocks = ceil_div(p.M, BM);
    const long n_blocks = ceil_div(p.n_tiles * 16, BN);
    hipLaunchKernelGGL(kern, dim3((unsigned)(m_blocks * n_blocks)), dim3(WARPS_M * WARPS_N * 64),
                       SMEM, st, p);
    return check_launch("gemm_w4a8");
}

template <int W_BITS, int EPI>
static int dispatch_tile(const GemmArgs &p, hipStream_t st)
{
    return launch_gemm<128, 128, 2, 2, W_BITS, EPI>(p, st);
}

static int gemm_common(const int8_t *a, long lda, const void *w, int w_bits, long M, long N,
                       long K_pad, float s_x0, float s_x1, const uint8_t *row_sel,
                       const float *s_w, const float *bias, const float *x0, const float *w0,
                       void *out, int epi, long ldo, void *stream)
{
    MQ_REQUIRE(M >= 0 && N >= 0 && K_pad >= 0, "mq_gemm_w4a8: negative shape");
    if (M == 0 || N == 0) return MQ_OK;
    MQ_REQUIRE(a && w && out, "mq_gemm_w4a8: null buffer");
    MQ_REQUIRE(K_pad > 0 && K_pad % 128 == 0, "mq_gemm_w4a8: K_pad=%ld must be a positive multiple of 128", K_pad);
    MQ_REQUIRE(lda >= K_pad && lda % 16 == 0 && ((uintptr_t)a) % 16 == 0,
               "mq_gemm_w4a8: activations must be 16-byte aligned with lda %% 16 == 0 and lda >= K_pad");
    MQ_REQUIRE(((uintptr_t)w) % 16 == 0, "mq_gemm_w4a8: weight image must be 16-byte aligned");
    MQ_REQUIRE(w_bits == 4 || w_bits == 8, "mq_gemm_w4a8: w_bits must be 4 or 8");
    MQ_REQUIRE(ldo >= N, "mq_gemm_w4a8: ldo < N");
    MQ_REQUIRE(epi == EPI_I32 || s_w, "mq_gemm_w4a8: s_w is required");
    MQ_REQUIRE((x0 == nullptr) == (w0 == nullptr), "mq_gemm_w4a8: x0 and w0 go together");
    // int32 headroom: |acc| <= K * 128 * 2^(w_bits-1) (* 16 for the high-nibble trick)
    MQ_REQUIRE(K_pad <= (w_bits == 4 ? 131072L : 131072L), "mq_gemm_w4a8: K too large for int32 accumulation");
    GemmArgs p;
    p.a = a; p.lda = lda; p.w = (const uint8_t *)w; p.M = M; p.N = N; p.K_pad = K_pad;
    p.n_tiles = ceil_div(N, 16);
    p.sx0 = s_x0; p.sx1 = s_x1; p.row_sel = row_sel; p.s_w = s_w; p.bias = bias; p.x0 = x0; p.w0 = w0;
    p.out = out; p.ldo = ldo;
    hipStream_t st = (hipStream_t)stream;
    if (w_bits == 4) {
        switch (epi) {
        case EPI_F16: return dispatch_tile<4, EPI_F16>(p, st);
        case EPI_BF16: return dispatch_tile<4, EPI_BF16>(p, st);
        case EPI_F32: return dispatch_tile<4, EPI_F32>(p, st);
        case EPI_I32: return dispatch_tile<4, EPI_I32>(p, st);
        }
    } else {
        switch (epi) {
        case EPI_F16: return dispatch_tile<8, EPI_F16>(p, st);
        case EPI_BF16: return dispatch_tile<8, EPI_BF16>(p, st);
        case EPI_F32: return dispatch_tile<8, EPI_F32>(p, st);
        case EPI_I32: return dispatch_tile<8, EPI_I32>(p, st);
        }
    }
    return fail(MQ_EINVAL, "mq_gemm_w4a8: unknown output dtype %d", epi);
}

}  // namespace mq

extern "C" int mq_gemm_w4a8(const int8_t *a, long lda, const void *w, int w_bits, long M, long N,
                            long K_pad, float s_x0, float s_x1, const uint8_t *row_sel,
                            const float *s_w, const float *bias, const float *x0, const float *w0,
                            void *out, int out_dtype, long ldo, void *stream)
{
    if (out_dtype != MQ_F16 && out_dtype != MQ_BF16 && out_dtype != MQ_F32)
        return mq::fail(MQ_EINVAL, "mq_gemm_w4a8: unknown output dtype %d", out_dtype);
    return mq::gemm_common(a, lda, w, w_bits, M, N, K_pad, s_x0, s_x1, row_sel, s_w, bias, x0, w0,
                           out, out_dtype, ldo, stream);
}

extern "C" int mq_gemm_w4a8_i32(const int8_t *a, long lda, const void *w, int w_bits, long M,
                                long N, long K_pad, int32_t *acc, long ldacc, void *stream)
{
    return mq::gemm_common(a, lda, w, w_bits, M, N, K_pad, 1.0f, 1.0f, nullptr, nullptr, nullptr,
                           nullptr, nullptr, acc, mq::EPI_I32, ldacc, stream);
}
