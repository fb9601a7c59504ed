// Probe: what a short kernel costs on the stream (kernel time + boundary), replayed from a hipGraph
// of 100 launches: the floor under every GEMM launch of the prefill (243 per step).
//   empty      : grid x 512 threads, nothing
//   store_rows : every workgroup stores a 96 x 128 fp16 tile of a [768][3584] matrix (16 B per lane,
//                256 B per row: the GEMM epilogue's pattern)                       5.5 MB per launch
//   store_flat : every workgroup stores 24 KiB contiguous                          5.5 MB per launch
//   load_stage : every workgroup LDS-DMAs 20 KiB (one pipeline stage), waits, exits
//   load_store : load_stage + store_rows
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void gbl_void;

struct Args { char *out; const char *in; int *sink; };

template <int MODE>
__global__ __launch_bounds__(512) void k(Args p)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int bm = blockIdx.x % 8, bn = blockIdx.x / 8;
    int v = tid;
    if (MODE == 3 || MODE == 4) {
#pragma unroll
        for (int i = 0; i < 5; ++i) {
            if (wave < 4) {
                const int f = wave + i * 4;
                __builtin_amdgcn_global_load_lds((gbl_void *)(p.in + ((long)blockIdx.x * 20 + f) * 1024 + lane * 16),
                                                 (lds_void *)(smem + f * 1024), 16, 0, 0);
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        v = *reinterpret_cast<const int *>(smem + tid * 4);
    }
    if (MODE == 1 || MODE == 4) {
        // 512 threads: 16 lanes per row (8 fp16 each = 16 B), 32 rows per pass, 3 passes
#pragma unroll
        for (int r0 = 0; r0 < 96; r0 += 32) {
            const int row = r0 + tid / 16, c8 = (tid % 16) * 8;
            char *o = p.out + (((long)bm * 96 + row) * 3584 + bn * 128 + c8) * 2;
            *reinterpret_cast<v4i *>(o) = v4i{v, v, v, v};
        }
    }
    if (MODE == 2) {
#pragma unroll
        for (int i = 0; i < 3; ++i)
            *reinterpret_cast<v4i *>(p.out + (long)blockIdx.x * 24576 + (i * 512 + tid) * 16) = v4i{v, v, v, v};
    }
    if (MODE == 3 && v == 0x7fffffff) p.sink[0] = v;
}

template <int MODE>
static void run(const char *name)
{
    char *out, *in; int *sink;
    hipMalloc(&out, 768L * 3584 * 2 + 4096); hipMalloc(&in, 224L * 20 * 1024 + 4096); hipMalloc(&sink, 64);
    hipMemset(in, 1, 224L * 20 * 1024);
    Args p{out, in, sink};
    hipStream_t st; hipStreamCreate(&st);
    auto kern = k<MODE>;
    hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
    hipLaunchKernelGGL(kern, dim3(224), dim3(512), 100 * 1024, st, p);
    hipStreamSynchronize(st);
    hipGraph_t g; hipGraphExec_t ge;
    hipStreamBeginCapture(st, hipStreamCaptureModeGlobal);
    for (int i = 0; i < 100; ++i) hipLaunchKernelGGL(kern, dim3(224), dim3(512), 100 * 1024, st, p);
    hipStreamEndCapture(st, &g);
    hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
    hipGraphLaunch(ge, st); hipStreamSynchronize(st);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0, st);
    for (int r = 0; r < 5; ++r) hipGraphLaunch(ge, st);
    hipEventRecord(e1, st); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-12s %6.2f us per launch (graph of 100, 224 workgroups x 512 threads, 100 KiB LDS)\n", name, ms * 1e3 / 500);
    fflush(stdout);
}

int main()
{
    run<0>("empty");
    run<1>("store_rows");
    run<2>("store_flat");
    run<3>("load_stage");
    run<4>("load_store");
    return 0;
}
