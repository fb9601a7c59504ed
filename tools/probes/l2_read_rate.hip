// Probe: what one CU can read from its XCD's L2 (bytes per clock), as a function of WHO reads WHAT.
//   private : every workgroup loops over its own 64 KiB (2 MiB per XCD: L2 hits, L1 misses)
//   xcd     : the workgroups of an XCD walk one shared 2 MiB region from staggered offsets
//   same    : every workgroup reads the same 64 KiB at the same time
//   mall    : 128 MiB footprint (misses the 32 MiB of L2, sits in the 256 MiB Infinity Cache)
// by plain global_load_dwordx4 into VGPRs (LDS = 0) or by LDS-DMA (LDS = 1), 1 KiB per wave
// instruction, U instructions in flight per wave, W waves per workgroup, one workgroup per CU
// (or two: blocks = 512).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void gbl_void;

struct Args { const char *buf; long region; long stride_block; long stride_xcd; long block_base; int iters; int *sink; };

template <int LDS, int NWAVES, int U>
__global__ __launch_bounds__(NWAVES * 64) void k(Args p)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
    const char *base = p.buf + (long)xcd * p.stride_xcd + (long)blockIdx.x * p.block_base;
    long off = ((long)idx * p.stride_block) % p.region;           // workgroup's start inside the region
    const long step = (long)NWAVES * U * 1024;                     // bytes per workgroup per iteration
    int acc = 0;
    for (int it = 0; it < p.iters; ++it) {
        v4i r[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            long o = off + ((long)(u * NWAVES + wave)) * 1024;
            if (o >= p.region) o -= p.region;
            const char *s = base + o + lane * 16;
            if (LDS) __builtin_amdgcn_global_load_lds((gbl_void *)s, (lds_void *)(smem + (u * NWAVES + wave) * 1024), 16, 0, 0);
            else r[u] = *reinterpret_cast<const v4i *>(s);
        }
        if (LDS) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            acc += *reinterpret_cast<const int *>(smem + threadIdx.x * 4);
        } else {
#pragma unroll
            for (int u = 0; u < U; ++u) acc ^= r[u][0] ^ r[u][1] ^ r[u][2] ^ r[u][3];
        }
        off += step;
        if (off >= p.region) off -= p.region;
    }
    if (acc == 0x7fffffff) p.sink[0] = acc;
}

template <int LDS, int NWAVES, int U>
static void run(const char *name, int blocks, long region, long stride_block, long stride_xcd, long block_base, long total)
{
    char *buf; int *sink;
    hipMalloc(&buf, total + 4096); hipMalloc(&sink, 64);
    hipMemset(buf, 1, total);
    const int iters = 256;
    Args p{buf, region, stride_block, stride_xcd, block_base, iters, sink};
    auto kern = k<LDS, NWAVES, U>;
    const int smem = LDS ? NWAVES * U * 1024 : 0;
    hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 2; ++i) hipLaunchKernelGGL(kern, dim3(blocks), dim3(NWAVES * 64), smem, 0, p);
    hipDeviceSynchronize();
    const int reps = 10;
    hipEventRecord(e0);
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(kern, dim3(blocks), dim3(NWAVES * 64), smem, 0, p);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double us = ms * 1e3 / reps;
    const double bytes = (double)blocks * iters * NWAVES * U * 1024.0;
    printf("%-8s %s waves %2d x %d in flight, blocks %3d: %8.1f us  %6.2f TB/s  %5.1f B/clk/CU\n", name, LDS ? "LDS-DMA" : "VGPR   ",
           NWAVES, U, blocks, us, bytes / us / 1e6, bytes / (us * 2400.0) / 256);
    fflush(stdout);
    hipFree(buf); hipFree(sink);
}

template <int LDS, int NWAVES, int U>
static void all_patterns(int blocks)
{
    const long K64 = 64 << 10, M2 = 2 << 20;
    run<LDS, NWAVES, U>("private", blocks, K64, 0, 0, K64, (long)blocks * K64);
    run<LDS, NWAVES, U>("xcd", blocks, M2, K64, M2, 0, 8 * M2);
    run<LDS, NWAVES, U>("same", blocks, K64, 0, 0, 0, K64);
    run<LDS, NWAVES, U>("mall", blocks, 16L << 20, 512 << 10, 16L << 20, 0, 128L << 20);
}

int main(int argc, char **argv)
{
    const int only = argc > 1 ? atoi(argv[1]) : -1;
    int idx = 0;
#define RUN(...) do { if (only < 0 || only == idx) { __VA_ARGS__; } ++idx; } while (0)
    RUN(all_patterns<0, 4, 4>(256));
    RUN(all_patterns<0, 4, 8>(256));
    RUN(all_patterns<0, 8, 4>(256));
    RUN(all_patterns<0, 16, 4>(256));
    RUN(all_patterns<0, 16, 2>(512));
    RUN(all_patterns<0, 16, 1>(512));
    RUN(all_patterns<1, 4, 8>(256));
    RUN(all_patterns<1, 8, 4>(256));
    RUN(all_patterns<1, 16, 4>(256));
    RUN(all_patterns<1, 16, 2>(512));
    return 0;
}
