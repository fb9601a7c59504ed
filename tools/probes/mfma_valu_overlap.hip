// Probe: do MFMAs of one wave overlap with VALU / LDS-read instructions of ANOTHER wave on the same
// SIMD?  Workgroups of 8 waves (one per CU): waves 0-3 (one per SIMD) run MFMAs only, waves 4-7 (again
// one per SIMD) run VALU only (or ds_read_b128 only).  If the pipes overlap, the combined run takes
// max(t_mfma, t_other); if issue is serialised per SIMD it takes the sum.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

struct Args { int n; int *sink; };

// MODE bit 0: waves 0-3 run 32x32x32 MFMAs; bit 1: waves 4-7 run VALU; bit 2: waves 4-7 run ds_read_b128;
// bit 3: the MFMA waves use the 16x16x64 form; bit 4: they use the fp32 form V_MFMA_F32_16X16X4_F32 (the Hadamard K x K stage)
template <int MODE>
__global__ __launch_bounds__(512) void k(Args p)
{
    __shared__ __attribute__((aligned(16))) char smem[32768];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    for (int i = threadIdx.x; i < 8192; i += 512) reinterpret_cast<int *>(smem)[i] = i;
    __syncthreads();
    int t = 0;
    if (wave < 4) {
        if (MODE & 1) {
            if (MODE & 16) {
                typedef float v4f __attribute__((ext_vector_type(4)));
                v4f acc[6];
#pragma unroll
                for (int j = 0; j < 6; ++j) acc[j] = v4f{0.f, 0.f, 0.f, 0.f};
                const float a = (float)lane, b = (float)(lane + 1);
                for (int it = 0; it < p.n; ++it) {
#pragma unroll
                    for (int r = 0; r < 4; ++r)
#pragma unroll
                        for (int j = 0; j < 6; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[j], 0, 0, 0);
                }
#pragma unroll
                for (int j = 0; j < 6; ++j) t ^= (int)acc[j][0];
            } else if (MODE & 8) {
                v4i acc[6];
#pragma unroll
                for (int j = 0; j < 6; ++j) acc[j] = v4i{0, 0, 0, 0};
                v4i a = {lane, 1, 2, 3}, b = {lane, 5, 6, 7};
                for (int it = 0; it < p.n; ++it) {
#pragma unroll
                    for (int j = 0; j < 6; ++j) acc[j] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, acc[j], 0, 0, 0);
#pragma unroll
                    for (int j = 0; j < 6; ++j) acc[j] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, acc[j], 0, 0, 0);
                }
#pragma unroll
                for (int j = 0; j < 6; ++j) t ^= acc[j][0];
            } else {
                v16i acc[6];
#pragma unroll
                for (int j = 0; j < 6; ++j)
#pragma unroll
                    for (int e = 0; e < 16; ++e) acc[j][e] = 0;
                v4i a = {lane, 1, 2, 3}, b = {lane, 5, 6, 7};
                for (int it = 0; it < p.n; ++it) {
#pragma unroll
                    for (int j = 0; j < 6; ++j) acc[j] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, acc[j], 0, 0, 0);
                }
#pragma unroll
                for (int j = 0; j < 6; ++j) t ^= acc[j][0];
            }
        }
    } else {
        if (MODE & 2) {
            int x0 = lane, x1 = lane * 3, x2 = lane * 5, x3 = lane * 7;
            for (int it = 0; it < p.n; ++it) {
#pragma unroll
                for (int u = 0; u < 12; ++u) {       // 48 independent-ish VALU per iteration
                    x0 = (x0 << 4) & 0xF0F0F0F0; x1 = x1 & 0xF0F0F0F1; x2 = (x2 << 3) ^ 0x55; x3 = x3 + 0x1234567;
                    asm volatile("" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3));
                }
            }
            t ^= x0 ^ x1 ^ x2 ^ x3;
        }
        if (MODE & 4) {
            v4i s = {0, 0, 0, 0};
            for (int it = 0; it < p.n; ++it) {
#pragma unroll
                for (int u = 0; u < 16; ++u) {
                    const v4i v = *reinterpret_cast<const v4i *>(smem + ((it + u) & 31) * 1024 + lane * 16);
                    s ^= v;
                }
            }
            t ^= s[0] ^ s[1] ^ s[2] ^ s[3];
        }
    }
    if (t == 0x12345678) p.sink[0] = t;
}

template <int MODE>
static double run(const char *name)
{
    int *sink; hipMalloc(&sink, 64);
    Args p{512, sink};
    for (int i = 0; i < 2; ++i) hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(512), 0, 0, p);
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    for (int i = 0; i < 10; ++i) hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(512), 0, 0, p);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double us = ms * 1e3 / 10;
    printf("%-58s %8.1f us per launch\n", name, us);
    fflush(stdout);
    return us;
}

int main()
{
    run<1>("32x32x32 MFMA waves alone (6 per iteration)");
    run<2>("VALU waves alone (48 per iteration)");
    run<3>("32x32x32 MFMA waves + VALU waves on the same SIMDs");
    run<4>("ds_read_b128 waves alone (16 per iteration)");
    run<5>("32x32x32 MFMA waves + ds_read waves on the same SIMDs");
    run<9>("16x16x64 MFMA waves alone (12 per iteration)");
    run<11>("16x16x64 MFMA waves + VALU waves");
    run<13>("16x16x64 MFMA waves + ds_read waves");
    run<17>("fp32 16x16x4 MFMA waves alone (24 per iteration)");
    run<19>("fp32 16x16x4 MFMA waves + VALU waves on the same SIMDs");
    run<21>("fp32 16x16x4 MFMA waves + ds_read waves on the same SIMDs");
    return 0;
}
