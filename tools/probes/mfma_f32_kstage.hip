// Probe: candidate inner loops for the Hadamard K x K stage (one wave per SIMD), cycles per MFMA.
//   NJ j-tiles x NC column tiles per unit; per k-step: NC staged values (fp16 in LDS) are converted,
//   NJ sign operands are derived from sign words, NJ*NC MFMAs issue.
//   PIPE = 0: operands prepared right before their MFMAs; PIPE = 1: prepared one k-step ahead.
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <stdio.h>
typedef float v4f __attribute__((ext_vector_type(4)));

template <int NJ, int NC, int PIPE>
__global__ __launch_bounds__(256) void k(const unsigned *gwords, float *out, int groups, long long *cyc)
{
    __shared__ __attribute__((aligned(16))) unsigned short y[160 * 128];
    __shared__ unsigned hw[176 * 5];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 160 * 128; i += 256) y[i] = (unsigned short)(0x3c00 + (i & 31));
    for (int i = threadIdx.x; i < 176 * 5; i += 256) hw[i] = gwords[i & 7] * (i + 1);
    __syncthreads();
    const int lc = lane & 15, lk = lane >> 4;
    const unsigned sh0 = 31u - (unsigned)lk;
    v4f acc[NJ][NC];
    for (int j = 0; j < NJ; ++j)
        for (int c = 0; c < NC; ++c) acc[j][c] = v4f{0.f, 0.f, 0.f, 0.f};
    struct Ops { float a[NJ]; float b[NC]; };
    unsigned word[NJ];
    auto prepare = [&](Ops &o, int g, int q) {
        const unsigned short *src = &y[((g * 8 + q) * 4 + lk) % 160 * 128 + wave * 32 + lc * NC];
#pragma unroll
        for (int c = 0; c < NC; ++c) o.b[c] = __half2float(__ushort_as_half(src[c]));
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            unsigned t = word[j] << (sh0 - 4u * (unsigned)q);
            t = (t & 0x80000000u) ^ 0xBF800000u;
            o.a[j] = __uint_as_float(t);
        }
    };
    auto fire = [&](const Ops &o) {
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int c = 0; c < NC; ++c)
                acc[j][c] = __builtin_amdgcn_mfma_f32_16x16x4f32(o.a[j], o.b[c], acc[j][c], 0, 0, 0);
    };
    const long long t0 = __builtin_readcyclecounter();
    Ops cur;
#pragma unroll
    for (int j = 0; j < NJ; ++j) word[j] = hw[(j * 16 + lc) * 5];
    if (PIPE) prepare(cur, 0, 0);
    for (int g = 0; g < groups; ++g) {
        unsigned wn[NJ];
#pragma unroll
        for (int j = 0; j < NJ; ++j) wn[j] = hw[(j * 16 + lc) * 5 + ((g + 1) % 5)];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            if (PIPE) {
                Ops nxt;
                if (q < 7) prepare(nxt, g, q + 1);
                else {
#pragma unroll
                    for (int j = 0; j < NJ; ++j) word[j] = wn[j];
                    prepare(nxt, g + 1, 0);
                }
                fire(cur);
                cur = nxt;
            } else {
                Ops o;
                prepare(o, g, q);
                fire(o);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if (!PIPE) {
#pragma unroll
            for (int j = 0; j < NJ; ++j) word[j] = wn[j];
        }
    }
    const long long t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
    float sum = 0.f;
    for (int j = 0; j < NJ; ++j)
        for (int c = 0; c < NC; ++c) sum += acc[j][c][0] + acc[j][c][3];
    out[blockIdx.x * 256 + threadIdx.x] = sum;
}

template <int NJ, int NC, int PIPE> static void run(int blocks)
{
    unsigned hwv[8] = {0x12345678u, 0x9abcdef0u, 0x0f0f0f0fu, 0x33333333u, 0x55555555u, 0xdeadbeefu, 0x1u, 0x80000000u};
    unsigned *dw; float *o; long long *c, hc;
    hipMalloc(&dw, sizeof(hwv)); hipMalloc(&o, blocks * 256 * 4); hipMalloc(&c, 8);
    hipMemcpy(dw, hwv, sizeof(hwv), hipMemcpyHostToDevice);
    const int groups = 500;
    hipLaunchKernelGGL((k<NJ, NC, PIPE>), dim3(blocks), dim3(256), 0, 0, dw, o, 5, c);
    hipDeviceSynchronize();
    hipLaunchKernelGGL((k<NJ, NC, PIPE>), dim3(blocks), dim3(256), 0, 0, dw, o, groups, c);
    hipDeviceSynchronize();
    hipMemcpy(&hc, c, 8, hipMemcpyDeviceToHost);
    printf("NJ=%2d NC=%d PIPE=%d blocks=%d: %.1f cycles per MFMA (wave 0)\n", NJ, NC, PIPE, blocks,
           (double)hc / ((double)groups * 8 * NJ * NC));
    hipFree(dw); hipFree(o); hipFree(c);
}

int main()
{
    run<1, 4, 0>(256); run<1, 4, 1>(256);     // the 16 rows x 64 columns unit
    run<5, 2, 0>(256); run<5, 2, 1>(256);
    run<10, 1, 0>(256); run<10, 1, 1>(256);
    run<10, 2, 0>(256); run<10, 2, 1>(256);
    run<5, 4, 0>(256); run<5, 4, 1>(256);
    run<3, 2, 1>(256); run<3, 4, 1>(256);
    run<10, 1, 1>(512); run<5, 2, 1>(512);
    return 0;
}
