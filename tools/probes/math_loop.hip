// Probe: cycles per k-step of the math-wave loop of gemm_ws.hip (96 x 128 tile: 6 activation
// fragments + 1 packed weight pair per k-tile, 24 x V_MFMA_I32_16X16X64_I8 per 128-byte k-step),
// with the pieces switched on one by one:
//   bit 0: the MFMAs            bit 1: the ds_read_b128 fragment reads      bit 2: the int4 unpack
//   bit 3: the s_barrier per k-step (with 4 extra waves that only wait at it, like the loaders)
// One workgroup per CU, NW math waves (one per SIMD at NW = 4, two at NW = 8).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef int v4i __attribute__((ext_vector_type(4)));

struct Args { int nk; long long *out; int *sink; };

template <int FLAGS, int NW, int TM>
__global__ __launch_bounds__((NW + ((FLAGS & 8) ? 4 : 0)) * 64) void k(Args p)
{
    constexpr bool MF = FLAGS & 1, RD = FLAGS & 2, UNP = FLAGS & 4, BAR = FLAGS & 8;
    constexpr int STAGE = (TM * NW / 4 * 2 + 8) * 1024, S = 4;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    for (int i = threadIdx.x; i < S * STAGE / 4; i += blockDim.x) reinterpret_cast<int *>(smem)[i] = i * 2654435761u;
    __syncthreads();
    v4i acc[2][TM];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < TM; ++j) acc[i][j] = v4i{0, 0, 0, 0};
    if (wave >= NW) {          // stand-ins for the loader waves: only the barriers
        for (int it = 0; it < p.nk; ++it) __builtin_amdgcn_s_barrier();
        return;
    }
    const int wn = wave % 4;
    struct Frag { v4i x[TM]; v4i w; };
    auto load = [&](Frag &f, int slot, int kt) {
        const char *xs = smem + slot * STAGE;
#pragma unroll
        for (int j = 0; j < TM; ++j) {
            if (RD) f.x[j] = *reinterpret_cast<const v4i *>(xs + (j * 2 + kt) * 1024 + lane * 16);
            else f.x[j] = v4i{lane + j, kt, slot, 7};
        }
        if (RD) f.w = *reinterpret_cast<const v4i *>(xs + TM * 2048 + (wn * 2 + kt) * 1024 + lane * 16);
        else f.w = v4i{lane, kt, slot, 3};
    };
    auto mfmas = [&](const Frag &f) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            v4i wf;
            if (UNP) {
                const int lo = f.w[i * 2], hi = f.w[i * 2 + 1];
                wf[0] = (lo << 4) & 0xF0F0F0F0; wf[1] = lo & 0xF0F0F0F0;
                wf[2] = (hi << 4) & 0xF0F0F0F0; wf[3] = hi & 0xF0F0F0F0;
            } else {
                wf = f.w;
            }
#pragma unroll
            for (int j = 0; j < TM; ++j) {
                if (MF) acc[i][j] = __builtin_amdgcn_mfma_i32_16x16x64_i8(wf, f.x[j], acc[i][j], 0, 0, 0);
                else acc[i][j] += wf ^ f.x[j];
            }
        }
    };
    Frag f0, f1;
    load(f0, 0, 0);
    int cur = 0;
    const long long t0 = clock64();
    for (int it = 0; it < p.nk; ++it) {
        if (BAR) __builtin_amdgcn_s_barrier();
        int nxt = cur + 1;
        if (nxt == S) nxt = 0;
        __builtin_amdgcn_sched_barrier(0);
        load(f1, cur, 1);
        __builtin_amdgcn_sched_barrier(0);
        mfmas(f0);
        __builtin_amdgcn_sched_barrier(0);
        load(f0, nxt, 0);
        __builtin_amdgcn_sched_barrier(0);
        mfmas(f1);
        __builtin_amdgcn_sched_barrier(0);
        cur = nxt;
    }
    const long long t1 = clock64();
    int t = 0;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < TM; ++j) t ^= acc[i][j][0] ^ acc[i][j][1] ^ acc[i][j][2] ^ acc[i][j][3];
    if (t == 0x12345678) p.sink[0] = t;
    if (blockIdx.x == 7 && threadIdx.x == 0) p.out[0] = t1 - t0;
}

typedef int v16i __attribute__((ext_vector_type(16)));
typedef int v2i __attribute__((ext_vector_type(2)));
// 32x32x32 variant of the same loop: a wave owns 96 rows x 32 columns = 3 tiles of 32x32, four K=32
// sub-steps per 128-byte k-step: 12 MFMAs (each twice the work of a 16x16x64).
//   bit 0: MFMAs   bit 1: fragment reads (3 x ds_read_b128 + 1 x ds_read_b64 per sub-step)
//   bit 2: int4 unpack (6 VALU per sub-step)   bit 3: s_barrier per k-step (+ 4 waiting waves)
//   bit 4: the unpacked weight registers alternate between two sets (else the compiler reuses one)
template <int FLAGS, int NW>
__global__ __launch_bounds__((NW + ((FLAGS & 8) ? 4 : 0)) * 64) void k32(Args p)
{
    constexpr bool MF = FLAGS & 1, RD = FLAGS & 2, UNP = FLAGS & 4, BAR = FLAGS & 8, ALT = FLAGS & 16;
    constexpr int STAGE = 20 * 1024, S = 4;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    for (int i = threadIdx.x; i < S * STAGE / 4; i += blockDim.x) reinterpret_cast<int *>(smem)[i] = i * 2654435761u;
    __syncthreads();
    if (wave >= NW) {
        for (int it = 0; it < p.nk; ++it) __builtin_amdgcn_s_barrier();
        return;
    }
    v16i acc[3];
#pragma unroll
    for (int j = 0; j < 3; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[j][e] = 0;
    const int wn = wave % 4;
    const int lane_x = ((lane >> 4) & 1) * 2048 + ((lane >> 5) * 16 + (lane & 15)) * 16;
    const int lane_w = ((lane >> 5) * 16 + (lane & 15)) * 16 + ((lane >> 4) & 1) * 8;
    struct Frag { v4i x[3]; v2i w; };
    auto load = [&](Frag &f, int slot, int sub) {
        const char *xs = smem + slot * STAGE + (sub >> 1) * 1024 + (sub & 1) * 512;
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            if (RD) f.x[j] = *reinterpret_cast<const v4i *>(xs + j * 4096 + lane_x);
            else f.x[j] = v4i{lane + j, sub, slot, 7};
        }
        if (RD) f.w = *reinterpret_cast<const v2i *>(xs + 12288 + wn * 2048 + lane_w);
        else f.w = v2i{lane + sub, slot};
    };
    v4i wfa = {1, 2, 3, 4}, wfb = {5, 6, 7, 8};
    auto mfmas = [&](const Frag &f, v4i &wf) {
        if (UNP) {
            wf[0] = (f.w[0] << 4) & 0xF0F0F0F0; wf[1] = f.w[0] & 0xF0F0F0F0;
            wf[2] = (f.w[1] << 4) & 0xF0F0F0F0; wf[3] = f.w[1] & 0xF0F0F0F0;
        } else {
            wf[0] ^= f.w[0];
        }
        if (ALT) asm volatile("" : "+v"(wf));       // keep the two sets in their own registers
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            if (MF) acc[j] = __builtin_amdgcn_mfma_i32_32x32x32_i8(wf, f.x[j], acc[j], 0, 0, 0);
            else acc[j][0] += wf[0] ^ f.x[j][0];
        }
    };
    Frag f0, f1;
    load(f0, 0, 0);
    int cur = 0;
    const long long t0 = clock64();
    for (int it = 0; it < p.nk; ++it) {
        if (BAR) __builtin_amdgcn_s_barrier();
        int nxt = cur + 1;
        if (nxt == S) nxt = 0;
        __builtin_amdgcn_sched_barrier(0);
        load(f1, cur, 1);
        __builtin_amdgcn_sched_barrier(0);
        mfmas(f0, wfa);
        __builtin_amdgcn_sched_barrier(0);
        load(f0, cur, 2);
        __builtin_amdgcn_sched_barrier(0);
        mfmas(f1, ALT ? wfb : wfa);
        __builtin_amdgcn_sched_barrier(0);
        load(f1, cur, 3);
        __builtin_amdgcn_sched_barrier(0);
        mfmas(f0, wfa);
        __builtin_amdgcn_sched_barrier(0);
        load(f0, nxt, 0);
        __builtin_amdgcn_sched_barrier(0);
        mfmas(f1, ALT ? wfb : wfa);
        __builtin_amdgcn_sched_barrier(0);
        cur = nxt;
    }
    const long long t1 = clock64();
    int t = wfa[0] ^ wfb[0];
#pragma unroll
    for (int j = 0; j < 3; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) t ^= acc[j][e];
    if (t == 0x12345678) p.sink[0] = t;
    if (blockIdx.x == 7 && threadIdx.x == 0) p.out[0] = t1 - t0;
}

template <int FLAGS, int NW>
static void run32(const char *name)
{
    long long *out; int *sink;
    hipMalloc(&out, 64); hipMalloc(&sink, 64);
    const int nk = 256;
    Args p{nk, out, sink};
    auto kern = k32<FLAGS, NW>;
    const int smem = 4 * 20 * 1024;
    hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    const int threads = (NW + ((FLAGS & 8) ? 4 : 0)) * 64;
    for (int i = 0; i < 2; ++i) hipLaunchKernelGGL(kern, dim3(256), dim3(threads), smem, 0, p);
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    for (int i = 0; i < 10; ++i) hipLaunchKernelGGL(kern, dim3(256), dim3(threads), smem, 0, p);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    long long cyc; hipMemcpy(&cyc, out, 8, hipMemcpyDeviceToHost);
    const double us_step = ms * 1e3 / 10 / nk;
    const double tops = 2.0 * 96 * 128 * 128 * (NW / 4) * 256 / (us_step * 1e-6) / 1e12;
    printf("32x32x32 %-35s waves %d      : %6.1f cycles/k-step (wave 0 of WG 7), %.3f us/k-step wall -> %6.0f TOP/s chip\n", name, NW,
           (double)cyc / nk, us_step, tops);
    fflush(stdout);
}

template <int FLAGS, int NW, int TM>
static void run(const char *name)
{
    long long *out; int *sink;
    hipMalloc(&out, 64); hipMalloc(&sink, 64);
    const int nk = 256;
    Args p{nk, out, sink};
    auto kern = k<FLAGS, NW, TM>;
    const int smem = 4 * (TM * NW / 4 * 2 + 8) * 1024;
    hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    const int threads = (NW + ((FLAGS & 8) ? 4 : 0)) * 64;
    for (int i = 0; i < 2; ++i) hipLaunchKernelGGL(kern, dim3(256), dim3(threads), smem, 0, p);
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    for (int i = 0; i < 10; ++i) hipLaunchKernelGGL(kern, dim3(256), dim3(threads), smem, 0, p);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    long long cyc; hipMemcpy(&cyc, out, 8, hipMemcpyDeviceToHost);
    const double us_step = ms * 1e3 / 10 / nk;
    const double tops = 2.0 * (TM * 16) * 128 * 128 * (NW / 4) / 1.0 * 256 / (us_step * 1e-6) / 1e12;
    printf("%-44s waves %d TM %2d: %6.1f cycles/k-step (wave 0 of WG 7), %.3f us/k-step wall -> %6.0f TOP/s chip\n", name, NW, TM,
           (double)cyc / nk, us_step, tops);
    fflush(stdout);
}

int main()
{
    run<1, 4, 6>("MFMA only");
    run<15, 4, 6>("MFMA + reads + unpack + barrier");
    run<1, 8, 6>("MFMA only");
    run<15, 8, 6>("MFMA + reads + unpack + barrier");
    run32<1, 4>("MFMA only");
    run32<3, 4>("MFMA + reads");
    run32<5, 4>("MFMA + unpack (one register set)");
    run32<21, 4>("MFMA + unpack (two register sets)");
    run32<7, 4>("MFMA + reads + unpack");
    run32<23, 4>("MFMA + reads + unpack, two sets");
    run32<15, 4>("MFMA + reads + unpack + barrier");
    run32<31, 4>("all, two sets");
    run32<9, 4>("MFMA + barrier");
    run32<1, 8>("MFMA only");
    run32<31, 8>("all, two sets");
    return 0;
}
