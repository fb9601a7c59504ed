// gemm_skinny.hip -- W4A8 GEMM for a FEW activation rows (M <= 64: the generation steps behind a prefill, the reference's
// exam/quant_*.py evaluation loops call ActQuantWrapper.forward -- quant_utils.py:330-384 -- once per token).  Same operands,
// weight image and epilogue arithmetic as the other GEMM kernels; what differs is the shape of the work.
//
// With one to four 16-row activation tiles the output offers N / 128 tiles at most (28 for down_proj): the tiled kernels then
// leave most CUs idle and each workgroup walks the whole reduction alone (10.7 us for q|k|v, 35 us for down_proj at M = 1,
// where the packed weights need 1.3 / 5.7 us of HBM time: profiles/r5_decode_gemm_bench.txt).  Here the WEIGHT STREAM is the
// work: a wave owns one 32-channel pair of the W4 image and a slice of K, pulls its 1 KiB pieces straight from HBM into
// registers (one global_load_dwordx4 per 64-wide k-tile: the piece is already in the MFMA fragment order, no LDS), reads the
// matching activation pieces (L2-resident, shared by every wave), unpacks the nibbles and issues V_MFMA_I32_16X16X64_I8.
// Two kernels: gemm_skinny_wg_kernel (plan id 61, up to two row tiles) -- one workgroup per pair, its EIGHT WAVES are the K slices and
// meet in LDS (short reductions: one launch, no workspace; long ones: a few workgroup slices on top) -- and gemm_skinny_kernel
// (plan id 60, up to four row tiles on long reductions) -- (pairs / 4) x slices workgroups of four waves, one wave per pair and slice.
//
// Reduction over slices of DIFFERENT workgroups: every workgroup parks its exact int32 partial sums in the split-K workspace ([slice][M][N], the
// layout of the tiled kernels' split-K) and splitk_reduce_kernel (gemm_w4a8.hip) adds them and runs the common epilogue in a
// second launch.  (Reducing inside the launch -- the last workgroup to count itself on a per-channel-block counter adds the
// slices -- was built first and is exact, but the agent-scope release / acquire it needs writes back and invalidates a whole
// L2 per workgroup on this chip: 56 us where the two launches take 9, profiles/r5_decode_gemm_bench.txt.)
//
// Reference semantics: fake_quant/quant_utils.py:384 (F.linear on the quantize-dequantized tensors); int32 accumulators exact.
#include "mq_common.h"
#include "gemm_common.h"

namespace mq {

constexpr int SK_MAX_BLOCKS = 65535;                       // channel blocks of 128 (grid.x)

template <int EPI, int TMX>
__global__ __launch_bounds__(256) void gemm_skinny_kernel(GemmArgs p)
{
    kernarg_warm<sizeof(GemmArgs), true>();
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const unsigned pb = blockIdx.x, ks = blockIdx.y;       // channel block (4 pairs = 128 channels), K slice
    const long kts = p.K_pad >> 6;
    long ntp = (long)pb * 4 + wave;
    const bool pair_ok = ntp < p.n_pairs;
    if (!pair_ok) ntp = p.n_pairs - 1;                     // (computed, never stored)
    const int kt0 = (int)ks * p.kq + ((int)ks < p.kr ? (int)ks : p.kr);
    const int nkt = p.kq + ((int)ks < p.kr ? 1 : 0);
    const long MT = (p.M + 15) >> 4;

    const char *wp = reinterpret_cast<const char *>(p.w) + (ntp * kts + kt0) * 1024 + lane * 16;
    const char *xp[TMX];
#pragma unroll
    for (int t = 0; t < TMX; ++t) {
        const long mt = t < MT ? t : MT - 1;               // (row tiles past M: recomputed, never stored)
        xp[t] = reinterpret_cast<const char *>(p.a) + (mt * kts + kt0) * 1024 + lane * 16;
    }
    v4i acc[TMX][2];
#pragma unroll
    for (int t = 0; t < TMX; ++t) acc[t][0] = acc[t][1] = v4i{0, 0, 0, 0};

    auto mac = [&](const v4i wq, const v4i (&x)[TMX]) {
        // lane's 16 bytes of the piece: 8 of the even 16-channel tile, 8 of the odd one; nibbles into the HIGH half (value x 16)
        const v4i w0 = v4i{(wq[0] << 4) & (int)0xF0F0F0F0, wq[0] & (int)0xF0F0F0F0, (wq[1] << 4) & (int)0xF0F0F0F0, wq[1] & (int)0xF0F0F0F0};
        const v4i w1 = v4i{(wq[2] << 4) & (int)0xF0F0F0F0, wq[2] & (int)0xF0F0F0F0, (wq[3] << 4) & (int)0xF0F0F0F0, wq[3] & (int)0xF0F0F0F0};
#pragma unroll
        for (int t = 0; t < TMX; ++t) {
            acc[t][0] = __builtin_amdgcn_mfma_i32_16x16x64_i8(w0, x[t], acc[t][0], 0, 0, 0);
            acc[t][1] = __builtin_amdgcn_mfma_i32_16x16x64_i8(w1, x[t], acc[t][1], 0, 0, 0);
        }
    };
    constexpr int U = TMX >= 4 ? 2 : 4;                    // k-tiles in flight per wave (8 for one row tile: measured, 5-8 % slower)
    int kt = 0;
    for (; kt + U <= nkt; kt += U) {
        v4i wq[U], x[U][TMX];
#pragma unroll
        for (int u = 0; u < U; ++u) wq[u] = __builtin_nontemporal_load(reinterpret_cast<const v4i *>(wp + (long)(kt + u) * 1024));
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int t = 0; t < TMX; ++t) x[u][t] = *reinterpret_cast<const v4i *>(xp[t] + (long)(kt + u) * 1024);
#pragma unroll
        for (int u = 0; u < U; ++u) mac(wq[u], x[u]);
    }
    for (; kt < nkt; ++kt) {
        v4i x[TMX];
        const v4i wq = __builtin_nontemporal_load(reinterpret_cast<const v4i *>(wp + (long)kt * 1024));
#pragma unroll
        for (int t = 0; t < TMX; ++t) x[t] = *reinterpret_cast<const v4i *>(xp[t] + (long)kt * 1024);
        mac(wq, x);
    }

    // D layout: column (lane & 15) = activation row of the tile, register r = channel 16 e + 4 (lane >> 4) + r of the pair's tile e
    const int ml = lane & 15, nq = (lane >> 4) * 4;
    auto row_scales = [&](long m, float &sx, float &xz, float &x1v) {
        sx = p.sx0; xz = 0.0f; x1v = 0.0f;
        if (EPI != EPI_I32) {
            if (p.sx_vec) sx = p.sx_vec[m];
            else if (p.row_sel && p.row_sel[m]) sx = p.sx1;
            if (p.x0) xz = p.x0[m];
            if (p.x1) x1v = p.x1[m];
        }
    };
    const int slices = (int)gridDim.y;
    if (slices == 1) {
#pragma unroll
        for (int t = 0; t < TMX; ++t) {
            const long m = t * 16 + ml;
            if (!pair_ok || m >= p.M) continue;
            float sx, xz, x1v;
            row_scales(m, sx, xz, x1v);
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const long n = (ntp * 2 + e) * 16 + nq;
                if (n < p.N) store_quad<EPI>(p, m, n, acc[t][e] >> 4, sx, xz, x1v);     // (levels x 16: the shift is exact)
            }
        }
        return;
    }
    const bool quads = (p.N % 4) == 0;
#pragma unroll
    for (int t = 0; t < TMX; ++t) {
        const long m = t * 16 + ml;
        if (!pair_ok || m >= p.M) continue;
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const long n = (ntp * 2 + e) * 16 + nq;
            if (n >= p.N) continue;
            const v4i a = acc[t][e] >> 4;
            int *dst = p.partial + ((long)ks * p.M + m) * p.N + n;
            if (quads) {
                *reinterpret_cast<v4i *>(dst) = a;
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (n + r < p.N) dst[r] = a[r];
            }
        }
    }
}

// Short reductions (K <= 4096: q|k|v, o_proj, gate|up): the K slices are the EIGHT WAVES of one workgroup, all on the same channel
// pair; their partial sums meet in LDS and the waves share the epilogue -- no workspace, no second launch (8.2 -> ~5 us for q|k|v
// at M = 1, profiles/r5_decode_gemm_bench.txt).  Plan id 61.
template <int EPI, int TMX>
__global__ __launch_bounds__(512) void gemm_skinny_wg_kernel(GemmArgs p)
{
    kernarg_warm<sizeof(GemmArgs), true>();
    __shared__ v4i red[8][TMX * 2][64];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const long kts = p.K_pad >> 6;
    const long ntp = blockIdx.x;                            // one channel pair per workgroup (grid.x = n_pairs)
    // grid.y > 1 (long reductions): workgroup slice gs owns p.kq (+1 if gs < p.kr) k-tiles, its eight waves share them; the
    // workgroups' sums then go through the workspace and splitk_reduce_kernel like the other kernel's
    const int gs = (int)blockIdx.y;
    const int g0 = gs * p.kq + (gs < p.kr ? gs : p.kr), gn = p.kq + (gs < p.kr ? 1 : 0);
    const int kq = gn >> 3, kr = gn & 7;                    // wave w owns kq (+1 if w < kr) of them
    const int kt0 = g0 + wave * kq + (wave < kr ? wave : kr);
    const int nkt = kq + (wave < kr ? 1 : 0);
    const long MT = (p.M + 15) >> 4;
    const char *wp = reinterpret_cast<const char *>(p.w) + (ntp * kts + kt0) * 1024 + lane * 16;
    const char *xp[TMX];
#pragma unroll
    for (int t = 0; t < TMX; ++t) {
        const long mt = t < MT ? t : MT - 1;
        xp[t] = reinterpret_cast<const char *>(p.a) + (mt * kts + kt0) * 1024 + lane * 16;
    }
    v4i acc[TMX][2];
#pragma unroll
    for (int t = 0; t < TMX; ++t) acc[t][0] = acc[t][1] = v4i{0, 0, 0, 0};
    auto mac = [&](const v4i wq, const v4i (&x)[TMX]) {
        const v4i w0 = v4i{(wq[0] << 4) & (int)0xF0F0F0F0, wq[0] & (int)0xF0F0F0F0, (wq[1] << 4) & (int)0xF0F0F0F0, wq[1] & (int)0xF0F0F0F0};
        const v4i w1 = v4i{(wq[2] << 4) & (int)0xF0F0F0F0, wq[2] & (int)0xF0F0F0F0, (wq[3] << 4) & (int)0xF0F0F0F0, wq[3] & (int)0xF0F0F0F0};
#pragma unroll
        for (int t = 0; t < TMX; ++t) {
            acc[t][0] = __builtin_amdgcn_mfma_i32_16x16x64_i8(w0, x[t], acc[t][0], 0, 0, 0);
            acc[t][1] = __builtin_amdgcn_mfma_i32_16x16x64_i8(w1, x[t], acc[t][1], 0, 0, 0);
        }
    };
    constexpr int U = 4;
    int kt = 0;
    for (; kt + U <= nkt; kt += U) {
        v4i wq[U], x[U][TMX];
#pragma unroll
        for (int u = 0; u < U; ++u) wq[u] = __builtin_nontemporal_load(reinterpret_cast<const v4i *>(wp + (long)(kt + u) * 1024));
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int t = 0; t < TMX; ++t) x[u][t] = *reinterpret_cast<const v4i *>(xp[t] + (long)(kt + u) * 1024);
#pragma unroll
        for (int u = 0; u < U; ++u) mac(wq[u], x[u]);
    }
    for (; kt < nkt; ++kt) {
        v4i x[TMX];
        const v4i wq = __builtin_nontemporal_load(reinterpret_cast<const v4i *>(wp + (long)kt * 1024));
#pragma unroll
        for (int t = 0; t < TMX; ++t) x[t] = *reinterpret_cast<const v4i *>(xp[t] + (long)kt * 1024);
        mac(wq, x);
    }
#pragma unroll
    for (int t = 0; t < TMX; ++t) {
        red[wave][t * 2][lane] = acc[t][0];
        red[wave][t * 2 + 1][lane] = acc[t][1];
    }
    __syncthreads();
    if (wave >= TMX * 2) return;                            // wave (2 t + e) finishes tile e of row tile t
    const int t = wave >> 1, e = wave & 1;
    v4i a = red[0][wave][lane];
#pragma unroll
    for (int w = 1; w < 8; ++w) a += red[w][wave][lane];   // integers: exact in any order
    const long m = t * 16 + (lane & 15), n = (ntp * 2 + e) * 16 + (lane >> 4) * 4;
    if (m >= p.M || n >= p.N) return;
    if (gridDim.y > 1) {                                    // one of several workgroup slices: park the exact partial sums
        const v4i q = a >> 4;
        int *dst = p.partial + ((long)gs * p.M + m) * p.N + n;
        if ((p.N % 4) == 0) {
            *reinterpret_cast<v4i *>(dst) = q;
        } else {
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (n + r < p.N) dst[r] = q[r];
        }
        return;
    }
    float sx = p.sx0, xz = 0.0f, x1v = 0.0f;
    if (EPI != EPI_I32) {
        if (p.sx_vec) sx = p.sx_vec[m];
        else if (p.row_sel && p.row_sel[m]) sx = p.sx1;
        if (p.x0) xz = p.x0[m];
        if (p.x1) x1v = p.x1[m];
    }
    store_quad<EPI>(p, m, n, a >> 4, sx, xz, x1v);          // (levels x 16: the shift is exact)
}

template <int EPI>
int launch_gemm_skinny_wg(const GemmArgs &p, hipStream_t st)
{
    if (!p.a_tiled || p.M > 32 || p.M < 1 || p.n_pairs > 0x7fffffffL || p.K_pad < 512)
        return fail(MQ_EINVAL, "gemm_skinny_wg: needs tiled activations, 1 <= M <= 32, K_pad >= 512");
    if (p.splits < 1 || p.splits > 65535) return fail(MQ_EINVAL, "gemm_skinny_wg: bad slice count %d", p.splits);
    GemmArgs g = p;
    const long kts = p.K_pad / 64;
    g.kq = (int)(kts / p.splits);
    g.kr = (int)(kts % p.splits);
    const dim3 grid((unsigned)p.n_pairs, (unsigned)p.splits);
    if (p.M <= 16) hipLaunchKernelGGL((gemm_skinny_wg_kernel<EPI, 1>), grid, dim3(512), 0, st, g);
    else hipLaunchKernelGGL((gemm_skinny_wg_kernel<EPI, 2>), grid, dim3(512), 0, st, g);
    return check_launch("gemm_skinny_wg");
}

template int launch_gemm_skinny_wg<EPI_F16>(const GemmArgs &, hipStream_t);
template int launch_gemm_skinny_wg<EPI_BF16>(const GemmArgs &, hipStream_t);
template int launch_gemm_skinny_wg<EPI_F32>(const GemmArgs &, hipStream_t);
template int launch_gemm_skinny_wg<EPI_I32>(const GemmArgs &, hipStream_t);

// workgroup slices of the eight-wave kernel on a long reduction: ~2 workgroups per CU, at least 32 k-tiles (four per wave) each
int skinny_wg_slices(long M, long N, long K_pad, size_t ws_bytes)
{
    const long pairs = ceil_div(N, 32), kts = K_pad / 64;
    long s = ceil_div(2L * device_cu_count(), pairs);
    if (s > kts / 32) s = kts / 32;
    if (s > 64) s = 64;
    while (s > 1 && (size_t)(s * M * N * 4) > ws_bytes) --s;
    return s > 1 ? (int)s : 1;
}

// K slices for a skinny launch: enough workgroups to stream the image from every CU (~3 per CU), at least four 64-wide k-tiles each
int skinny_slices(long M, long N, long K_pad, size_t ws_bytes)
{
    const long blocks = ceil_div(ceil_div(N, 32), 4), kts = K_pad / 64;
    long s = ceil_div(3L * device_cu_count(), blocks);
    if (s > kts / 4) s = kts / 4;
    if (s > 64) s = 64;
    while (s > 1 && (size_t)(s * M * N * 4) > ws_bytes) --s;
    return s > 1 ? (int)s : 1;
}

template <int EPI>
int launch_gemm_skinny(const GemmArgs &p, hipStream_t st)
{
    const long blocks = ceil_div(p.n_pairs, 4);
    if (!p.a_tiled || p.M > 64 || p.M < 1 || blocks > SK_MAX_BLOCKS || p.splits < 1 || p.splits > 65535)
        return fail(MQ_EINVAL, "gemm_skinny: needs tiled activations, 1 <= M <= 64 and at most %d channel blocks", SK_MAX_BLOCKS);
    GemmArgs g = p;
    const long kts = p.K_pad / 64;
    g.kq = (int)(kts / p.splits);
    g.kr = (int)(kts % p.splits);
    const dim3 grid((unsigned)blocks, (unsigned)p.splits);
    if (p.M <= 16) hipLaunchKernelGGL((gemm_skinny_kernel<EPI, 1>), grid, dim3(256), 0, st, g);
    else if (p.M <= 32) hipLaunchKernelGGL((gemm_skinny_kernel<EPI, 2>), grid, dim3(256), 0, st, g);
    else hipLaunchKernelGGL((gemm_skinny_kernel<EPI, 4>), grid, dim3(256), 0, st, g);
    return check_launch("gemm_skinny");
}

template int launch_gemm_skinny<EPI_F16>(const GemmArgs &, hipStream_t);
template int launch_gemm_skinny<EPI_BF16>(const GemmArgs &, hipStream_t);
template int launch_gemm_skinny<EPI_F32>(const GemmArgs &, hipStream_t);
template int launch_gemm_skinny<EPI_I32>(const GemmArgs &, hipStream_t);

}  // namespace mq
