// valu_pk_rate.hip -- issue rate of the fp32 vector ALU forms a VALU formulation of the exact K x K Hadamard stage could use
// (round 4): v_fma_f32, v_pk_fma_f32 (VGPR and SGPR multiplier), v_pk_add_f32, v_add_f32; independent accumulators, 1 / 2 / 4
// waves per SIMD on every CU.  Prints cycles per wave-instruction per SIMD (s_memtime) and the clock the chip ran at.
//   hipcc --offload-arch=gfx950 -O3 -o valu_pk_rate.bin valu_pk_rate.hip && ./valu_pk_rate.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef float v2f __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ __launch_bounds__(256) void burn(float *out, unsigned long long *cyc, int iters, float sgn)
{
    v2f acc[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = v2f{(float)threadIdx.x, (float)i};
    v2f x = v2f{1.0f + threadIdx.x * 1e-3f, 2.0f};
    const float s = __builtin_amdgcn_readfirstlane(__float_as_int(sgn)) ? sgn : 1.0f;   // wave-uniform
    const unsigned long long sp = ((unsigned long long)__float_as_uint(s) << 32) | __float_as_uint(s);   // +-1 in an SGPR pair; op_sel_hi 0 reads the low word twice
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            if (MODE == 0) { asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(acc[i][0]) : "v"(x[0]), "v"(x[1])); }
            if (MODE == 1) { asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(x), "v"(x)); }
            if (MODE == 2) { asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(acc[i]) : "v"(x), "s"(sp)); }
            if (MODE == 3) { asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(acc[i]) : "v"(x)); }
            if (MODE == 4) { asm volatile("v_add_f32 %0, %0, %1" : "+v"(acc[i][0]) : "v"(x[0])); }
            if (MODE == 5) { asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(acc[i][0]) : "v"(x[0]), "s"(s)); }
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float r = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) r += acc[i][0] + acc[i][1];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int MODE>
static void run(const char *name, int waves_per_simd)
{
    const int blocks = 256 * waves_per_simd, iters = 20000;
    float *out; unsigned long long *cyc;
    hipMalloc(&out, blocks * 256 * 4); hipMalloc(&cyc, blocks * 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    burn<MODE><<<blocks, 256>>>(out, cyc, 100, -1.0f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    burn<MODE><<<blocks, 256>>>(out, cyc, iters, -1.0f);
    hipEventRecord(e1); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(blocks);
    hipMemcpy(h.data(), cyc, blocks * 8, hipMemcpyDeviceToHost);
    double avg = 0; for (auto c : h) avg += (double)c; avg /= blocks;
    const double per = avg / (iters * 16.0) / waves_per_simd;   // SIMD cycles per wave-instruction
    printf("%-34s %d wave(s)/SIMD: %6.2f cycles per wave-instruction per SIMD, kernel %7.3f ms, clock %.2f GHz\n", name, waves_per_simd,
           per, ms, avg / (ms * 1e6));
    hipFree(out); hipFree(cyc);
}

int main()
{
    for (int w : {1, 2, 4}) {
        run<0>("v_fma_f32 (3 VGPR)", w);
        run<5>("v_fma_f32 (SGPR multiplier)", w);
        run<4>("v_add_f32", w);
        run<1>("v_pk_fma_f32 (3 VGPR pairs)", w);
        run<2>("v_pk_fma_f32 (SGPR multiplier, splat)", w);
        run<3>("v_pk_add_f32", w);
    }
    return 0;
}
