// Probe: sustained issue rate of V_MFMA_F32_16X16X4_F32 from ONE wave per SIMD with 4 independent
// accumulators, bare and with the operand preparation of the Hadamard K x K stage mixed in.
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <stdio.h>
typedef float v4f __attribute__((ext_vector_type(4)));
typedef unsigned short v4us __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ __launch_bounds__(256) void k(const unsigned *words, float *out, int iters, long long *cyc)
{
    __shared__ __attribute__((aligned(16))) unsigned short lds[64 * 64 * 4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 64 * 64 * 4; i += 256) lds[i] = (unsigned short)(0x3c00 + (i & 15));
    __syncthreads();
    v4f acc[4];
    for (int g = 0; g < 4; ++g) acc[g] = v4f{0.f, 0.f, 0.f, 0.f};
    const unsigned sh0 = 31u - (unsigned)(lane >> 4);
    unsigned word = words[lane & 7];
    const long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
        if (MODE >= 2) word = ((const unsigned *)lds)[(it * 5 + (lane & 15) * 8) & 1023];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            float a = 1.0f;
            float b[4] = {1.f, 1.f, 1.f, 1.f};
            if (MODE >= 1) {
                unsigned t = word << (sh0 - 4u * (unsigned)q);
                t = (t & 0x80000000u) ^ 0xBF800000u;
                a = __uint_as_float(t);
            }
            if (MODE == 2) {   // fp32 staging: one 16-byte read, no conversion
                const v4f fv = *reinterpret_cast<const v4f *>(&lds[(((it * 8 + q) & 31) * 256 + lane * 4) * 2]);
#pragma unroll
                for (int g = 0; g < 4; ++g) b[g] = fv[g];
            }
            if (MODE == 3) {   // fp16 staging: one 8-byte read + 4 conversions
                const v4us hv = *reinterpret_cast<const v4us *>(&lds[((it * 8 + q) & 63) * 256 + lane * 4]);
#pragma unroll
                for (int g = 0; g < 4; ++g) b[g] = __half2float(__ushort_as_half(hv[g]));
            }
#pragma unroll
            for (int g = 0; g < 4; ++g) acc[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b[g], acc[g], 0, 0, 0);
        }
    }
    const long long t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
    out[blockIdx.x * 256 + threadIdx.x] = acc[0][0] + acc[1][1] + acc[2][2] + acc[3][3] + (float)wave;
}

template <int MODE> static void run(const char *name, int blocks)
{
    unsigned hw[8] = {0x12345678u, 0x9abcdef0u, 0x0f0f0f0fu, 0x33333333u, 0x55555555u, 0xdeadbeefu, 0x1u, 0x80000000u};
    unsigned *dw; float *o; long long *c, hc;
    hipMalloc(&dw, sizeof(hw)); hipMalloc(&o, blocks * 256 * 4); hipMalloc(&c, 8);
    hipMemcpy(dw, hw, sizeof(hw), hipMemcpyHostToDevice);
    const int iters = 2000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, dw, o, 10, c);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, dw, o, iters, c);
    hipEventRecord(e1); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    hipMemcpy(&hc, c, 8, hipMemcpyDeviceToHost);
    const double mf = (double)iters * 32.0;
    printf("%-28s blocks=%4d: %.1f ns per MFMA per wave, %.1f shader-clock ticks per MFMA (wall %.3f ms)\n", name, blocks,
           ms * 1e6 / mf, (double)hc / mf, ms);
}

int main()
{
    for (int blocks : {256, 512, 768}) {   // 1, 2, 3 waves per SIMD
        run<0>("bare mfma", blocks);
        run<1>("+ sign operand", blocks);
        run<2>("+ lds b128 (fp32 staging)", blocks);
        run<3>("+ lds b64 + 4 cvt (fp16)", blocks);
    }
    return 0;
}
