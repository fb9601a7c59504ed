// Probe: L2 -> CU read rate of the GEMM's ACTIVATION access pattern as a function of the row
// stride.  One wave instruction (global_load_dwordx4 or LDS-DMA, 16 B per lane) fetches either
//   SHAPE 0: 8 rows x one whole 128-byte line          (lane l: row l & 7,  bytes 16 (l >> 3))
//   SHAPE 1: 16 rows x half a line (the MFMA fragment) (lane l: row l & 15, bytes 16 (l >> 4))
//   SHAPE 2: 4 rows x 256 bytes                        (lane l: row l & 3,  bytes 16 (l >> 2))
//   SHAPE 3: 1 KiB contiguous (reference: the weight stream)
// of a row-major [768][stride] int8 matrix; a workgroup owns 128 rows and walks along the row in
// 128-byte (SHAPE 2: 256-byte) k-steps, as the GEMM does.  All workgroups with the same row block
// read the same bytes (L2 hits after the first touch).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void gbl_void;

struct Args { const char *buf; long stride; int nk; int m_blocks; int *sink; };

template <int LDS, int SHAPE, int NWAVES>
__global__ __launch_bounds__(NWAVES * 64) void k(Args p)
{
    constexpr int ROWS = 128;                       // rows per workgroup
    constexpr int RPI = SHAPE == 0 ? 8 : SHAPE == 1 ? 16 : SHAPE == 2 ? 4 : 8;   // rows per instruction
    constexpr int KB = SHAPE == 2 ? 256 : 128;      // bytes of k per step
    constexpr int PIECES = ROWS * KB / 1024;        // instructions per k-step per workgroup
    constexpr int U = PIECES / NWAVES;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const long m0 = (long)(blockIdx.x % p.m_blocks) * ROWS;
    const char *src[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const int f = u * NWAVES + wave;
        if (SHAPE == 0) src[u] = p.buf + (m0 + f * 8 + (lane & 7)) * p.stride + (lane >> 3) * 16;
        else if (SHAPE == 1) src[u] = p.buf + (m0 + (f >> 1) * 16 + (lane & 15)) * p.stride + (f & 1) * 64 + (lane >> 4) * 16;
        else if (SHAPE == 2) src[u] = p.buf + (m0 + f * 4 + (lane & 3)) * p.stride + (lane >> 2) * 16;
        else src[u] = p.buf + (m0 * p.stride) + ((long)f * p.nk) * 1024 + lane * 16;   // piece-major image
    }
    int acc = 0;
    constexpr int KU = 4;                           // k-steps in flight per wave
    for (int it = 0; it + KU <= p.nk; it += KU) {
        v4i r[KU][U];
#pragma unroll
        for (int kk = 0; kk < KU; ++kk)
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const char *s = src[u] + (SHAPE == 3 ? (long)(it + kk) * 1024 : (long)(it + kk) * KB);
                if (LDS) __builtin_amdgcn_global_load_lds((gbl_void *)s, (lds_void *)(smem + ((kk * U + u) * NWAVES + wave) * 1024), 16, 0, 0);
                else r[kk][u] = *reinterpret_cast<const v4i *>(s);
            }
        if (LDS) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            acc += *reinterpret_cast<const int *>(smem + threadIdx.x * 4);
        } else {
#pragma unroll
            for (int kk = 0; kk < KU; ++kk)
#pragma unroll
                for (int u = 0; u < U; ++u) acc ^= r[kk][u][0] ^ r[kk][u][1] ^ r[kk][u][2] ^ r[kk][u][3];
        }
    }
    if (acc == 0x7fffffff) p.sink[0] = acc;
}

template <int LDS, int SHAPE, int NWAVES>
static void run(long stride, int blocks)
{
    const int m_blocks = 6;
    const int nk = (int)(stride / (SHAPE == 2 ? 256 : 128));
    char *buf; int *sink;
    const long total = 768 * stride;
    hipMalloc(&buf, total + 4096); hipMalloc(&sink, 64);
    hipMemset(buf, 1, total);
    Args p{buf, stride, nk, m_blocks, sink};
    auto kern = k<LDS, SHAPE, NWAVES>;
    const int smem = LDS ? 4 * 128 * (SHAPE == 2 ? 256 : 128) : 0;
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
    const double bytes = (double)blocks * 128.0 * (nk / 4 * 4) * (SHAPE == 2 ? 256 : 128);
    static const char *names[] = {"8 rows x 128 B", "16 rows x 64 B", "4 rows x 256 B", "1 KiB contiguous"};
    printf("stride %6ld  %-16s %s waves %2d blocks %3d: %7.1f us  %6.2f TB/s  %5.1f B/clk/CU\n", stride, names[SHAPE],
           LDS ? "LDS-DMA" : "VGPR   ", NWAVES, blocks, us, bytes / us / 1e6, bytes / (us * 2400.0) / 256);
    fflush(stdout);
    hipFree(buf); hipFree(sink);
}

int main(int argc, char **argv)
{
    const int only = argc > 1 ? atoi(argv[1]) : -1;
    int idx = 0;
#define RUN(...) do { if (only < 0 || only == idx) { __VA_ARGS__; } ++idx; } while (0)
    for (long stride : {1280L, 3584L, 4096L, 5120L, 8192L, 19968L}) {
        RUN(run<0, 3, 16>(stride, 256));
        RUN(run<0, 0, 16>(stride, 256));
        RUN(run<0, 1, 16>(stride, 256));
        RUN(run<0, 2, 16>(stride, 256));
        RUN(run<1, 0, 16>(stride, 256));
        RUN(run<1, 1, 16>(stride, 256));
        RUN(run<0, 0, 8>(stride, 256));
        RUN(run<0, 1, 8>(stride, 256));
        RUN(run<0, 0, 8>(stride, 512));
        RUN(run<0, 1, 8>(stride, 512));
    }
    return 0;
}
