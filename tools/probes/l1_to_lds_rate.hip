// Probe: how many bytes per clock one CU can move global -> LDS, (a) with LDS-DMA
// (global_load_lds_dwordx4, the GEMM's current path, ring of 3 stages, counted vmcnt) and
// (b) through registers (global_load_dwordx4 -> ds_write_b128, software-pipelined DEPTH stages deep).
// Access pattern = the W4A8 GEMM's: per 128-byte k-step a workgroup fetches X_FRAGS activation
// pieces (16 rows x 64 B, row stride lda) and W pieces (1 KiB contiguous).  All workgroups of an
// XCD re-read a small footprint (L2-resident) or stream distinct data (footprint = 0).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void gbl_void;

struct Args { const char *a; long lda; const char *w; int nk; long a_rows; long w_pieces_per_kt; int *sink; };

#ifndef FULL_LINES
#define FULL_LINES 0
#endif
#ifndef SAME_STEP      // 1: every k-step re-reads the SAME 16-48 KB (L1-resident after the first pass)
#define SAME_STEP 0
#endif
template <int MODE, int NWAVES, int XF, int WP, int DEPTH>
__global__ __launch_bounds__(NWAVES * 64) void k(Args p)
{
    constexpr int PIECES = XF + WP, LPW = PIECES / NWAVES, STAGE = PIECES * 1024, RING = 3;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const char *src[LPW]; int step[LPW];
    const long m0 = ((long)blockIdx.x * (XF / 2) * 16) % p.a_rows;
    const long wp0 = ((long)blockIdx.x * (WP / 2)) % (p.w_pieces_per_kt);
#pragma unroll
    for (int i = 0; i < LPW; ++i) {
        const int f = wave + i * NWAVES;
        if (f < XF) {
            const int mt = f >> 1, kt = f & 1;
            if (FULL_LINES)   // one instruction = 8 rows x one whole 128-byte line
                src[i] = p.a + (m0 + mt * 16 + kt * 8 + (lane & 7)) * p.lda + (lane >> 3) * 16;
            else              // one instruction = 16 rows x half a line (the GEMM's fragment order)
                src[i] = p.a + (m0 + mt * 16 + (lane & 15)) * p.lda + kt * 64 + (lane >> 4) * 16;
            step[i] = SAME_STEP ? 0 : 128;
        } else {
            const int g = f - XF;
            src[i] = p.w + (((wp0 + (g >> 1)) * (long)(2 * p.nk)) * 2 + (g & 1)) * 1024 + lane * 16; step[i] = SAME_STEP ? 0 : 2048;
        }
    }
    int acc = 0;
    if (MODE == 0) {
        auto issue = [&](int stage, int it) {
#pragma unroll
            for (int i = 0; i < LPW; ++i)
                __builtin_amdgcn_global_load_lds((gbl_void *)(src[i] + (long)it * step[i]),
                                                 (lds_void *)(smem + stage * STAGE + (wave + i * NWAVES) * 1024), 16, 0, 0);
        };
        issue(0, 0); issue(1, 1);
        int cur = 0;
        for (int it = 0; it < p.nk; ++it) {
            if (it + 1 < p.nk) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LPW) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            int nxt = cur + 2; if (nxt >= RING) nxt -= RING;
            if (it + 2 < p.nk) issue(nxt, it + 2);
            acc += *reinterpret_cast<const int *>(smem + cur * STAGE + threadIdx.x * 4);
            if (++cur == RING) cur = 0;
        }
    } else {
        v4i r[DEPTH][LPW];
#pragma unroll
        for (int d = 0; d < DEPTH; ++d)
#pragma unroll
            for (int i = 0; i < LPW; ++i) r[d][i] = *reinterpret_cast<const v4i *>(src[i] + (long)d * step[i]);
        for (int it = 0; it < p.nk; it += DEPTH) {
#pragma unroll
            for (int d = 0; d < DEPTH; ++d) {
                // oldest stage in registers -> LDS; refill the slot with stage it+d+DEPTH
                asm volatile("s_waitcnt vmcnt(%0)" ::"n"((DEPTH - 1) * LPW) : "memory");
                char *dst = smem + ((it + d) & 1) * STAGE;
#pragma unroll
                for (int i = 0; i < LPW; ++i) *reinterpret_cast<v4i *>(dst + (wave + i * NWAVES) * 1024 + lane * 16) = r[d][i];
                int nx = it + d + DEPTH; if (nx >= p.nk) nx = p.nk - 1;
#pragma unroll
                for (int i = 0; i < LPW; ++i) r[d][i] = *reinterpret_cast<const v4i *>(src[i] + (long)nx * step[i]);
                __syncthreads();
                acc += *reinterpret_cast<const int *>(dst + threadIdx.x * 4);
            }
        }
    }
    if (acc == 0x7fffffff) p.sink[0] = acc;
}

template <int MODE, int NWAVES, int XF, int WP, int DEPTH>
static void run(const char *name, int blocks, int shared_footprint)
{
    const int nk = 64;
    const long lda = 128L * nk;
    // 2: ~1.5 MB re-read by everybody (fits every XCD's 4 MB L2); 1: 24 MB (Infinity Cache); 0: distinct data
    const long a_rows = shared_footprint == 2 ? (XF / 2) * 16 : shared_footprint ? 1024 : (long)blocks * (XF / 2) * 16;
    const long w_pairs = shared_footprint == 2 ? (WP / 2) : shared_footprint ? 64 : (long)blocks * (WP / 2);
    char *a, *w; int *sink;
    hipMalloc(&a, a_rows * lda + 4096); hipMalloc(&w, w_pairs * 2L * nk * 2 * 1024 + 4096); hipMalloc(&sink, 64);
    hipMemset(a, 1, a_rows * lda); hipMemset(w, 1, w_pairs * 2L * nk * 2 * 1024);
    Args p{a, lda, w, nk, a_rows, w_pairs, sink};
    auto kern = k<MODE, NWAVES, XF, WP, DEPTH>;
    const int smem = 3 * (XF + WP) * 1024;
    hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(kern, dim3(blocks), dim3(NWAVES * 64), smem, 0, p);
    hipDeviceSynchronize();
    const int reps = 20;
    hipEventRecord(e0);
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(kern, dim3(blocks), dim3(NWAVES * 64), smem, 0, p);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double us = ms * 1e3 / reps;
    const double bytes = (double)blocks * nk * (XF + WP) * 1024.0;
    const int cus = blocks < 256 ? blocks : 256;
    printf("%-44s blocks %4d %s  %7.1f us  %6.2f TB/s  %5.1f B/clk/CU (%.2f us per k-step)\n", name, blocks,
           shared_footprint == 2 ? "L2-resident " : shared_footprint ? "MALL-resident" : "streaming    ", us, bytes / us / 1e6, bytes / (us * 2400.0) / cus, us / nk);
    hipFree(a); hipFree(w); hipFree(sink);
}

int main()
{
    for (int sf = 2; sf >= 0; --sf) {
        for (int blocks : {256, 512}) {
            run<0, 4, 8, 8, 1>("64x128 tile, 4 waves, LDS-DMA ring3", blocks, sf);
            run<1, 4, 8, 8, 1>("64x128 tile, 4 waves, regs depth1", blocks, sf);
            run<1, 4, 8, 8, 2>("64x128 tile, 4 waves, regs depth2", blocks, sf);
            run<1, 4, 8, 8, 4>("64x128 tile, 4 waves, regs depth4", blocks, sf);
        }
        run<0, 8, 16, 8, 1>("128x128 tile, 8 waves, LDS-DMA ring3", 256, sf);
        run<1, 8, 16, 8, 2>("128x128 tile, 8 waves, regs depth2", 256, sf);
        run<1, 8, 16, 8, 4>("128x128 tile, 8 waves, regs depth4", 256, sf);
        run<0, 16, 32, 16, 1>("256x256 tile, 16 waves, LDS-DMA ring3", 256, sf);
        run<1, 16, 32, 16, 2>("256x256 tile, 16 waves, regs depth2", 256, sf);
        run<1, 16, 32, 16, 4>("256x256 tile, 16 waves, regs depth4", 256, sf);
    }
    return 0;
}
