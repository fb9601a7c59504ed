// Probe: what dense int8 rate does the chip SUSTAIN (power / clock management included)?  Every SIMD of every CU runs
// register-only V_MFMA_I32_32X32X32_I8 chains (no memory, no LDS) for `ms` milliseconds; the achieved rate against the
// nominal 5 POP/s (2.4 GHz x 256 CUs x 8192 op/clk) is the clock the part actually holds under matrix load.  A short
// single-wave chain of dependent VALU adds before (idle clock) and right after the burn reads the shader clock itself.
// build: hipcc --offload-arch=gfx950 -O2 -o mfma_i8_sustained.bin mfma_i8_sustained.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

__global__ __launch_bounds__(256) void burn(int iters, int *sink, int waves_per_simd_tag)
{
    v16i acc[4];
    for (int j = 0; j < 4; ++j)
        for (int e = 0; e < 16; ++e) acc[j][e] = 0;
    v4i a = {(int)threadIdx.x, 1, 2, 3}, b = {4, 5, 6, (int)blockIdx.x};
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, acc[j], 0, 0, 0);
    }
    int t = 0;
    for (int j = 0; j < 4; ++j)
        for (int e = 0; e < 16; ++e) t += acc[j][e];
    if (t == 0x7fffffff) sink[0] = t;
}

__global__ void meter(int n, float *out)
{
    float v = (float)threadIdx.x;
    for (int i = 0; i < n; ++i) {
#pragma unroll
        for (int u = 0; u < 16; ++u) v = v + 1.0f;     // dependent chain
    }
    out[threadIdx.x] = v;
}

static float time_kernel(void (*launch)(void *), void *arg)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    launch(arg);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    return ms;
}

// burn with LDS traffic on top: every MFMA operand pair is re-read from LDS (ds_read_b128), as a GEMM's math loop does
__global__ __launch_bounds__(256) void burn_lds(int iters, int *sink)
{
    __shared__ __attribute__((aligned(16))) int lds[16384];
    for (int i = threadIdx.x; i < 16384; i += 256) lds[i] = i * 2654435761u;
    __syncthreads();
    v16i acc[4];
    for (int j = 0; j < 4; ++j)
        for (int e = 0; e < 16; ++e) acc[j][e] = 0;
    const v4i *src = reinterpret_cast<const v4i *>(lds) + threadIdx.x;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const v4i a = src[((i * 4 + j) & 7) * 256], b = src[(((i * 4 + j) & 7) + 8) * 256];
            acc[j] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, acc[j], 0, 0, 0);
        }
    }
    int t = 0;
    for (int j = 0; j < 4; ++j)
        for (int e = 0; e < 16; ++e) t += acc[j][e];
    if (t == 0x7fffffff) sink[0] = t;
}

int main(int argc, char **argv)
{
    int *sink; float *out;
    hipMalloc(&sink, 4); hipMalloc(&out, 256);
    const int chain = 200000;                       // 3.2 M dependent adds
    auto run_meter = [&]() {
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0);
        hipLaunchKernelGGL(meter, dim3(1), dim3(64), 0, 0, chain, out);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        return ms;
    };
    hipLaunchKernelGGL(meter, dim3(1), dim3(64), 0, 0, 1000, out);
    hipDeviceSynchronize();
    float m_idle = run_meter();
    m_idle = run_meter();
    printf("meter, idle chip: %.3f ms for %d dependent adds\n", m_idle, chain * 16);
    for (int wps = 1; wps <= 2; ++wps) {
        for (int rep = 0; rep < 3; ++rep) {
            const int iters = 200000;
            const int blocks = 256 * wps;              // 256 threads = 4 waves = one per SIMD; wps workgroups per CU
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            hipEventRecord(e0);
            hipLaunchKernelGGL(burn, dim3(blocks), dim3(256), 0, 0, iters, sink, wps);
            hipEventRecord(e1);
            hipLaunchKernelGGL(meter, dim3(1), dim3(64), 0, 0, chain, out);
            hipEvent_t e2; hipEventCreate(&e2); hipEventRecord(e2);
            hipEventSynchronize(e2);
            float ms, mm; hipEventElapsedTime(&ms, e0, e1); hipEventElapsedTime(&mm, e1, e2);
            const double ops = (double)blocks * 4 * iters * 4 * 65536.0;
            printf("burn %d wave(s)/SIMD: %.2f ms, %.3f POP/s (%.1f %% of 5.0); meter right after: %.3f ms = %.3f x idle\n",
                   wps, ms, ops / ms / 1e12, ops / ms / 1e12 / 5.0 * 100, mm, mm / m_idle);
        }
    }
    // the meter DURING a burn: second stream, one extra wave on some CU; its dependent-add chain does not compete for
    // the matrix pipe, so its duration reads the shader clock under load
    hipStream_t s2; hipStreamCreate(&s2);
    for (int kind = 0; kind < 2; ++kind) {
        for (int rep = 0; rep < 2; ++rep) {
            const int iters = 400000, blocks = 512;
            hipEvent_t b0, b1, m0, m1; hipEventCreate(&b0); hipEventCreate(&b1); hipEventCreate(&m0); hipEventCreate(&m1);
            hipEventRecord(b0, 0);
            if (kind == 0) hipLaunchKernelGGL(burn, dim3(blocks), dim3(256), 0, 0, iters, sink, 2);
            else hipLaunchKernelGGL(burn_lds, dim3(blocks), dim3(256), 0, 0, iters, sink);
            hipEventRecord(b1, 0);
            hipEventRecord(m0, s2);
            hipLaunchKernelGGL(meter, dim3(1), dim3(64), 0, s2, chain, out);
            hipEventRecord(m1, s2);
            hipDeviceSynchronize();
            float ms, mm; hipEventElapsedTime(&ms, b0, b1); hipEventElapsedTime(&mm, m0, m1);
            const double ops = (double)blocks * 4 * iters * 4 * 65536.0;
            printf("%s burn 2 waves/SIMD: %.2f ms, %.3f POP/s; meter CONCURRENT with it: %.3f ms = %.3f x idle\n",
                   kind ? "MFMA + ds_read_b128" : "register-only MFMA", ms, ops / ms / 1e12, mm, mm / m_idle);
        }
    }
    return 0;
}
