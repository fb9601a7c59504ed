// Probe: bytes per clock one CU can pull from L2 when the two GEMM operands take DIFFERENT paths.
//   MODE 0: activations and weights both by LDS-DMA (global_load_lds_dwordx4), the round-1 GEMM;
//   MODE 1: both by global_load_dwordx4 into VGPRs (no LDS at all);
//   MODE 2: activations by LDS-DMA, weights by global_load_dwordx4 into VGPRs (each wave owns its
//           weight columns, so the weight operand needs no sharing through LDS).
// Geometry = the small W4A8 GEMMs (o_proj: M 768, K 3584, N 3584, 96x128 tile, 4 waves): per
// 128-byte k-step a workgroup needs XP activation pieces and WP weight pieces of 1 KiB.  Eight
// workgroups share a weight panel, every workgroup of a row block shares the activation rows
// (L2-resident after the first touch), as in the real launch.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void gbl_void;

struct Args { const char *a; long lda; const char *w; int nk; int m_blocks; int *sink; };

#define VLOAD(dst, ptr) asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(dst) : "v"(ptr) : "memory")

template <int MODE, int NWAVES, int XP, int WP, int DEPTH, int FULL_LINES>
__global__ __launch_bounds__(NWAVES * 64) void k(Args p)
{
    constexpr int XPW = XP / NWAVES, WPW = WP / NWAVES;
    constexpr int DMA_PW = (MODE == 0) ? XPW + WPW : (MODE == 2) ? XPW : 0;      // LDS-DMA ops per wave per k-step
    constexpr int REG_PW = (MODE == 1) ? XPW + WPW : (MODE == 2) ? WPW : 0;      // VGPR loads per wave per k-step
    constexpr int OPS = DMA_PW + REG_PW;
    constexpr int STAGE = (XP + WP) * 1024;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int bm = blockIdx.x % p.m_blocks, bn = blockIdx.x / p.m_blocks;
    const char *xs[XPW]; const char *ws[WPW];
#pragma unroll
    for (int i = 0; i < XPW; ++i) {
        const int f = wave + i * NWAVES, mt = f >> 1, kt = f & 1;
        const long m0 = (long)bm * (XP / 2) * 16;
        if (FULL_LINES) xs[i] = p.a + (m0 + mt * 16 + kt * 8 + (lane & 7)) * p.lda + (lane >> 3) * 16;
        else xs[i] = p.a + (m0 + mt * 16 + (lane & 15)) * p.lda + kt * 64 + (lane >> 4) * 16;
    }
#pragma unroll
    for (int i = 0; i < WPW; ++i) {
        const int g = wave * WPW + i;     // a wave's pieces are adjacent (its own columns)
        ws[i] = p.w + ((((long)bn * (WP / 2) + (g >> 1)) * p.nk) * 2 + (g & 1)) * 1024 + lane * 16;
    }
    v4i r[DEPTH][REG_PW > 0 ? REG_PW : 1];
#pragma unroll
    for (int d = 0; d < DEPTH; ++d)
#pragma unroll
        for (int i = 0; i < (REG_PW > 0 ? REG_PW : 1); ++i) r[d][i] = v4i{0, 0, 0, 0};
    int acc = 0;
    auto issue = [&](int slot, int it, v4i (&rr)[REG_PW > 0 ? REG_PW : 1]) {
        char *base = smem + slot * STAGE;
#pragma unroll
        for (int i = 0; i < XPW; ++i) {
            const char *s = xs[i] + (long)it * 128;
            if (MODE == 1) VLOAD(rr[i], s);
            else __builtin_amdgcn_global_load_lds((gbl_void *)s, (lds_void *)(base + (wave + i * NWAVES) * 1024), 16, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < WPW; ++i) {
            const char *s = ws[i] + (long)it * 2048;
            if (MODE == 0) __builtin_amdgcn_global_load_lds((gbl_void *)s, (lds_void *)(base + (XP + wave + i * NWAVES) * 1024), 16, 0, 0);
            else VLOAD(rr[(MODE == 1 ? XPW : 0) + i], s);
        }
    };
#pragma unroll
    for (int d = 0; d < DEPTH - 1; ++d) issue(d, d, r[d]);
    for (int it = 0; it < p.nk; it += DEPTH) {
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
            // k-step it+d has landed when at most DEPTH-2 younger k-steps are outstanding
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"((DEPTH - 2) * OPS) : "memory");
            if (MODE != 1) __builtin_amdgcn_s_barrier();
            int nx = it + d + DEPTH - 1; if (nx >= p.nk) nx = p.nk - 1;
            issue((d + DEPTH - 1) % DEPTH, nx, r[(d + DEPTH - 1) % DEPTH]);
            if (MODE != 1) acc += *reinterpret_cast<const int *>(smem + d * STAGE + threadIdx.x * 4);
#pragma unroll
            for (int i = 0; i < REG_PW; ++i) {
                asm volatile("" : "+v"(r[d][i]));
                acc ^= r[d][i][0] ^ r[d][i][1] ^ r[d][i][2] ^ r[d][i][3];
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int d = 0; d < DEPTH; ++d)      // keep every destination register of the asm loads live up to here
#pragma unroll
        for (int i = 0; i < REG_PW; ++i) {
            asm volatile("" : "+v"(r[d][i]));
            acc ^= r[d][i][0];
        }
    if (acc == 0x7fffffff) p.sink[0] = acc;
}

template <int MODE, int NWAVES, int XP, int WP, int DEPTH, int FULL_LINES>
static void run(const char *name, int m_blocks, int n_blocks, int nk)
{
    const long lda = 128L * nk;
    const long a_rows = (long)m_blocks * (XP / 2) * 16;
    const long w_bytes = (long)n_blocks * (WP / 2) * nk * 2 * 1024;
    char *a, *w; int *sink;
    hipMalloc(&a, a_rows * lda + 4096); hipMalloc(&w, w_bytes + 4096); hipMalloc(&sink, 64);
    hipMemset(a, 1, a_rows * lda); hipMemset(w, 1, w_bytes);
    Args p{a, lda, w, nk, m_blocks, sink};
    auto kern = k<MODE, NWAVES, XP, WP, DEPTH, FULL_LINES>;
    const int smem = DEPTH * (XP + WP) * 1024;
    hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    const int blocks = m_blocks * n_blocks;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(kern, dim3(blocks), dim3(NWAVES * 64), smem, 0, p);
    hipDeviceSynchronize();
    const int reps = 20;
    hipEventRecord(e0);
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(kern, dim3(blocks), dim3(NWAVES * 64), smem, 0, p);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double us = ms * 1e3 / reps;
    const double bytes = (double)blocks * nk * (XP + WP) * 1024.0;
    const int cus = blocks < 256 ? blocks : 256;
    printf("%-58s blocks %4d nk %3d  %7.1f us  %6.2f TB/s  %5.1f B/clk/CU\n", name, blocks, nk, us, bytes / us / 1e6,
           bytes / (us * 2400.0) / cus);
    fflush(stdout);
    hipFree(a); hipFree(w); hipFree(sink);
}

int main(int argc, char **argv)
{
    const int only = argc > 1 ? atoi(argv[1]) : -1;      // run ONE configuration (a faulting one must not hide the rest)
    int idx = 0;
#define RUN(...) do { if (only < 0 || only == idx) { __VA_ARGS__; } ++idx; } while (0)
    for (int nk : {24, 120}) {
        // 96x128 tile, 4 waves: 12 activation pieces + 8 weight pieces per k-step; 8 x 28 = 224 workgroups
        RUN(run<0, 4, 12, 8, 3, 0>("96x128 4w  all LDS-DMA            depth3 frag-rows", 8, 28, nk));
        RUN(run<0, 4, 12, 8, 3, 1>("96x128 4w  all LDS-DMA            depth3 full-lines", 8, 28, nk));
        RUN(run<0, 4, 12, 8, 4, 1>("96x128 4w  all LDS-DMA            depth4 full-lines", 8, 28, nk));
        RUN(run<1, 4, 12, 8, 2, 0>("96x128 4w  all VGPR               depth2 frag-rows", 8, 28, nk));
        RUN(run<1, 4, 12, 8, 3, 0>("96x128 4w  all VGPR               depth3 frag-rows", 8, 28, nk));
        RUN(run<1, 4, 12, 8, 3, 1>("96x128 4w  all VGPR               depth3 full-lines", 8, 28, nk));
        RUN(run<2, 4, 12, 8, 3, 0>("96x128 4w  x LDS-DMA + w VGPR     depth3 frag-rows", 8, 28, nk));
        RUN(run<2, 4, 12, 8, 3, 1>("96x128 4w  x LDS-DMA + w VGPR     depth3 full-lines", 8, 28, nk));
        RUN(run<2, 4, 12, 8, 4, 1>("96x128 4w  x LDS-DMA + w VGPR     depth4 full-lines", 8, 28, nk));
        // 128x128 tile, 8 waves (2 waves per SIMD): 16 + 8 pieces
        RUN(run<0, 8, 16, 8, 3, 1>("128x128 8w all LDS-DMA            depth3 full-lines", 6, 28, nk));
        RUN(run<1, 8, 16, 8, 3, 1>("128x128 8w all VGPR               depth3 full-lines", 6, 28, nk));
        RUN(run<2, 8, 16, 8, 3, 1>("128x128 8w x LDS-DMA + w VGPR     depth3 full-lines", 6, 28, nk));
        // weight-heavy split: 4 activation + 16 weight pieces
        RUN(run<0, 4, 4, 16, 3, 1>("32x256 4w  all LDS-DMA            depth3 full-lines", 24, 14, nk));
        RUN(run<1, 4, 4, 16, 3, 1>("32x256 4w  all VGPR               depth3 full-lines", 24, 14, nk));
        RUN(run<2, 4, 4, 16, 3, 1>("32x256 4w  x LDS-DMA + w VGPR     depth3 full-lines", 24, 14, nk));
    }
    return 0;
}
