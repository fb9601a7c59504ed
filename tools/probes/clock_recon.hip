// Probe (round 5): which clock does the chip run at under int8 matrix load, and do the readings agree?
//
// Four readings of one launch, side by side:
//   (1) s_memtime ticks per s_memrealtime tick (100 MHz constant clock), both read INSIDE the kernel by every workgroup:
//       the tick rate of the shader-cycle counter while the kernel runs, free of host timing;
//   (2) host-timed duration (HIP events) against the instruction count: matrix ops per second;
//   (3) rocm-smi sclk + package power, sampled from a child process while the kernel runs back to back for ~2.5 s;
//   (4) GRBM_GUI_ACTIVE / SQ_BUSY_CYCLES over the kernel's duration from a separate rocprofv3 --pmc pass of this binary
//       (tools/r5_clock.sh).
// Kernels:
//   burn<KIND>: register-only V_MFMA_I32_32X32X32_I8 (KIND 0) or V_MFMA_I32_16X16X64_I8 (KIND 1) chains, 2 waves per SIMD
//       on every CU, with operand DATA of three kinds: zeros, one small constant in every lane, random bytes (4 x 4 operand
//       register sets rotated so that consecutive MFMAs see different bits on both ports).  A power-managed part runs the
//       same instruction stream slower on busier data; a schedule-bound kernel cannot tell the difference;
//   meter: ONE wave, a chain of dependent VALU adds then a chain of s_sleep, each bracketed by s_memtime + s_memrealtime:
//       what an instruction stream experiences (cycles per add is a constant of the pipeline), alone and CONCURRENT with
//       the burn on a second stream.
// build: hipcc --offload-arch=gfx950 -O2 -o clock_recon.bin clock_recon.hip
// usage: clock_recon.bin [seconds per phase = 2.5] [phase filter]
#include <hip/hip_runtime.h>
#include <algorithm>
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>
#include <unistd.h>

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));
typedef unsigned long long u64;

#define CK(x)                                                                              \
    do {                                                                                   \
        hipError_t e_ = (x);                                                               \
        if (e_ != hipSuccess) {                                                            \
            fprintf(stderr, "%s:%d %s -> %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_)); \
            exit(1);                                                                       \
        }                                                                                  \
    } while (0)

template <int KIND>
__global__ __launch_bounds__(512) void burn(int iters, const v4i *data, u64 *stamps, int *sink)
{
    extern __shared__ char force_one_wg_per_cu[];
    const int lane = threadIdx.x & 63;
    v4i A[4], B[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        A[j] = data[j * 64 + lane];
        B[j] = data[(4 + j) * 64 + lane];
    }
    const u64 t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    if (KIND == 0) {
        v16i acc[4];
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[j][e] = 0;
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_i32_32x32x32_i8(A[j], B[(j + r) & 3], acc[j], 0, 0, 0);
        }
        int t = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) t += acc[j][e];
        if (t == 0x7fffffff) sink[0] = t;
    } else {
        v4i acc[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] = v4i{0, 0, 0, 0};
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int j = 0; j < 8; ++j) acc[j] = __builtin_amdgcn_mfma_i32_16x16x64_i8(A[j & 3], B[(j + r + (j >> 2)) & 3], acc[j], 0, 0, 0);
        }
        int t = 0;
#pragma unroll
        for (int j = 0; j < 8; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e) t += acc[j][e];
        if (t == 0x7fffffff) sink[0] = t;
    }
    const u64 t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0) {
        stamps[blockIdx.x * 2] = t1 - t0;
        stamps[blockIdx.x * 2 + 1] = r1 - r0;
    }
}

__global__ void meter(int n_add, int n_sleep, u64 *out, float *fsink)
{
    float v = (float)threadIdx.x;
    const u64 t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < n_add; ++i) {
#pragma unroll
        for (int u = 0; u < 16; ++u) asm volatile("v_add_f32 %0, 1.0, %0" : "+v"(v));
    }
    const u64 t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < n_sleep; ++i) __builtin_amdgcn_s_sleep(127);
    const u64 t2 = __builtin_amdgcn_s_memtime(), r2 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0) {
        out[0] = t1 - t0;
        out[1] = r1 - r0;
        out[2] = t2 - t1;
        out[3] = r2 - r1;
    }
    fsink[threadIdx.x] = v;
}

static std::string smi_sample()
{
    if (getenv("CLOCK_RECON_NO_SMI")) return "(skipped)";     // under rocprofv3: no child processes
    FILE *f = popen("/opt/rocm/bin/rocm-smi --showpower --showclocks 2>/dev/null | grep -E 'sclk|Power' | tr -s ' \\t' ' ' | tr '\\n' ';'", "r");
    if (!f) return "popen failed";
    char buf[1024];
    std::string s;
    while (fgets(buf, sizeof buf, f)) s += buf;
    pclose(f);
    return s;
}

static double median(std::vector<double> v)
{
    std::sort(v.begin(), v.end());
    return v.empty() ? 0.0 : v[v.size() / 2];
}

int main(int argc, char **argv)
{
    const double seconds = argc > 1 ? atof(argv[1]) : 2.5;
    const char *filter = argc > 2 ? argv[2] : "";
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    int wall_khz = 0, clk_khz = 0;
    CK(hipDeviceGetAttribute(&wall_khz, hipDeviceAttributeWallClockRate, 0));
    CK(hipDeviceGetAttribute(&clk_khz, hipDeviceAttributeClockRate, 0));
    const int cus = prop.multiProcessorCount;
    printf("device %s, %d CUs, hipDeviceAttributeClockRate %d kHz, WallClockRate (s_memrealtime) %d kHz\n", prop.gcnArchName, cus, clk_khz, wall_khz);
    const double wall_hz = wall_khz * 1e3;

    hipStream_t s1, s2;
    CK(hipStreamCreate(&s1));
    CK(hipStreamCreate(&s2));
    v4i *data;
    u64 *stamps, *mout;
    int *sink;
    float *fsink;
    CK(hipMalloc(&data, 8 * 64 * sizeof(v4i)));
    CK(hipMalloc(&stamps, cus * 2 * sizeof(u64)));
    CK(hipMalloc(&mout, 4 * sizeof(u64)));
    CK(hipMalloc(&sink, 64));
    CK(hipMalloc(&fsink, 64 * sizeof(float)));
    CK(hipFuncSetAttribute((const void *)burn<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024));
    CK(hipFuncSetAttribute((const void *)burn<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024));

    auto run_meter = [&](hipStream_t st, const char *label) {
        const int n_add = 20000, n_sleep = 40;
        hipLaunchKernelGGL(meter, dim3(1), dim3(64), 0, st, n_add, n_sleep, mout, fsink);
        CK(hipStreamSynchronize(st));
        u64 h[4];
        CK(hipMemcpy(h, mout, sizeof h, hipMemcpyDeviceToHost));
        printf("  meter %-28s: %d dependent v_add_f32 = %llu ticks (%.3f per add) in %.2f us -> s_memtime %.3f GHz | %d x s_sleep 127 = %llu ticks (%.0f per sleep) in %.2f us -> %.3f GHz\n",
               label, n_add * 16, h[0], (double)h[0] / (n_add * 16), h[1] / wall_hz * 1e6, h[0] / (h[1] / wall_hz) * 1e-9,
               n_sleep, h[2], (double)h[2] / n_sleep, h[3] / wall_hz * 1e6, h[2] / (h[3] / wall_hz) * 1e-9);
        fflush(stdout);
    };

    printf("\n== idle chip\n");
    for (int i = 0; i < 3; ++i) run_meter(s2, "alone (idle chip)");
    printf("  rocm-smi: %s\n", smi_sample().c_str());

    struct Phase { const char *name; int kind; int mode; };
    const Phase phases[] = {
        {"mfma32x32x32 zeros", 0, 0}, {"mfma32x32x32 constant", 0, 1}, {"mfma32x32x32 random", 0, 2},
        {"mfma16x16x64 zeros", 1, 0}, {"mfma16x16x64 constant", 1, 1}, {"mfma16x16x64 random", 1, 2},
    };
    for (const Phase &ph : phases) {
        if (filter[0] && !strstr(ph.name, filter)) continue;
        std::vector<int> h(8 * 64 * 4);
        srand(1234);
        for (auto &w : h) w = ph.mode == 0 ? 0 : ph.mode == 1 ? 0x01020304 : (int)((unsigned)rand() * 2654435761u ^ (unsigned)rand() << 7);
        CK(hipMemcpy(data, h.data(), h.size() * 4, hipMemcpyHostToDevice));
        const int iters = 6000;                           // x 16 MFMAs (32x32x32) or x 32 (16x16x64) per wave
        const double ops_per_launch = (double)cus * 8 * iters * (ph.kind == 0 ? 16 * 2.0 * 32 * 32 * 32 : 32 * 2.0 * 16 * 16 * 64);
        auto launch = [&]() {
            if (ph.kind == 0) hipLaunchKernelGGL(burn<0>, dim3(cus), dim3(512), 100 * 1024, s1, iters, data, stamps, sink);
            else hipLaunchKernelGGL(burn<1>, dim3(cus), dim3(512), 100 * 1024, s1, iters, data, stamps, sink);
        };
        for (int i = 0; i < 3; ++i) launch();
        CK(hipStreamSynchronize(s1));
        std::atomic<bool> stop{false};
        std::vector<std::string> smi;
        std::thread watcher([&]() {
            usleep(400000);
            while (!stop.load()) {
                smi.push_back(smi_sample());
                usleep(300000);
            }
        });
        hipEvent_t e0, e1;
        CK(hipEventCreate(&e0));
        CK(hipEventCreate(&e1));
        double total_ms = 0.0;
        long launches = 0;
        std::vector<double> ghz, cyc_per_mfma;
        bool metered = false;
        while (total_ms < seconds * 1e3) {
            CK(hipEventRecord(e0, s1));
            for (int i = 0; i < 20; ++i) launch();
            CK(hipEventRecord(e1, s1));
            if (!metered && total_ms > seconds * 300) {   // once per phase: the meter concurrent with the burn
                run_meter(s2, "CONCURRENT with the burn");
                metered = true;
            }
            CK(hipEventSynchronize(e1));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            total_ms += ms;
            launches += 20;
            std::vector<u64> st(cus * 2);
            CK(hipMemcpy(st.data(), stamps, st.size() * 8, hipMemcpyDeviceToHost));
            std::vector<double> g, c;
            for (int b = 0; b < cus; ++b) {
                g.push_back(st[2 * b] / (st[2 * b + 1] / wall_hz) * 1e-9);
                c.push_back((double)st[2 * b] / (iters * (ph.kind == 0 ? 16 : 32)));
            }
            ghz.push_back(median(g));
            cyc_per_mfma.push_back(median(c));
        }
        stop = true;
        watcher.join();
        const double us = total_ms / launches * 1e3;
        printf("== burn %-24s: %.1f us per launch, %.3f POP/s (%.1f %% of 5.0) | in-kernel: s_memtime ticks at %.3f GHz (median WG, median launch), %.2f ticks per MFMA per wave (2 waves / SIMD)\n",
               ph.name, us, ops_per_launch / (us * 1e-6) * 1e-15, ops_per_launch / (us * 1e-6) / 5e15 * 100, median(ghz), median(cyc_per_mfma));
        for (size_t i = 0; i < smi.size() && i < 4; ++i) printf("  rocm-smi: %s\n", smi[i].c_str());
        fflush(stdout);
    }
    printf("\n== idle again\n");
    run_meter(s2, "alone (after the burns)");
    return 0;
}
