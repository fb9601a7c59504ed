// bench_probe.hip -- BENCH-ONLY (libmquant_bench.so, include/mquant_bench.h; not part of libmquant_hip.so): the dense int8 matrix rate the box SUSTAINS under its package power limit.
//
// The W4A8 GEMM family runs at the power limit: the same instruction stream is 10-30 % slower on real operand bytes than on
// zeros (profiles/r5_clock_reconciliation.txt).  The nominal 5 POP/s (2.4 GHz x 256 CUs x 8192 op/clk) therefore is not what
// the matrix cores of a given box can deliver on the benchmark's data.  This kernel measures that ceiling: register-only
// V_MFMA_I32_16X16X64_I8 (kind 1, what the GEMM kernels issue) or V_MFMA_I32_32X32X32_I8 (kind 0) chains, two waves per SIMD on every
// CU, no memory and no LDS traffic, operands = 2 x 4 fragment register sets loaded from caller-supplied bytes (bench.py passes
// int8 activation levels and int4 weight levels in the high nibble, as the GEMMs see them) and rotated so that consecutive
// MFMAs see different bits on both ports.  bench.py reports it as roofline.peak_sustained_measured next to the nominal peak.
#include "mq_common.h"
#include "../../include/mquant_bench.h"

namespace mq {

typedef int v16i_p __attribute__((ext_vector_type(16)));

template <int KIND>
__global__ __launch_bounds__(512) void mfma_burn_kernel(int iters, const v4i *data, int *sink)
{
    extern __shared__ char one_workgroup_per_cu[];
    const int lane = threadIdx.x & 63;
    v4i A[4], B[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        A[j] = data[j * 64 + lane];
        B[j] = data[(4 + j) * 64 + lane];
    }
    int t = 0;
    if (KIND == 0) {
        v16i_p acc[4];
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
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) t += acc[j][e];
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
#pragma unroll
        for (int j = 0; j < 8; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e) t += acc[j][e];
    }
    if (t == 0x7fffffff) sink[0] = t;        // keeps the chains alive; practically never taken
}

}  // namespace mq

extern "C" const char *mq_bench_last_error(void) { return mq::last_error_buf(); }

extern "C" int mq_bench_mfma_burn(int kind, const void *operands, int iters, int launches, int *sink, double *ops_per_s, void *stream)
{
    using namespace mq;
    MQ_REQUIRE(kind == 0 || kind == 1, "mq_bench_mfma_burn: kind 0 (32x32x32) or 1 (16x16x64)");
    MQ_REQUIRE(operands && sink && ops_per_s && iters > 0 && launches > 0, "mq_bench_mfma_burn: bad arguments");
    MQ_REQUIRE(((uintptr_t)operands) % 16 == 0, "mq_bench_mfma_burn: operands must be 16-byte aligned (8 x 64 x 16 bytes)");
    hipStream_t st = (hipStream_t)stream;
    const void *kern = kind == 0 ? (const void *)mfma_burn_kernel<0> : (const void *)mfma_burn_kernel<1>;
    constexpr int SMEM = 100 * 1024;             // one 8-wave workgroup per CU: two waves per SIMD
    int rc = ensure_dynamic_lds(kern, SMEM);
    if (rc != MQ_OK) return rc;
    const int cus = device_cu_count();
    hipEvent_t e0, e1;
    if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) return fail(MQ_EUNSUPPORTED, "mq_bench_mfma_burn: hipEventCreate failed");
    auto launch = [&]() {
        if (kind == 0) hipLaunchKernelGGL(mfma_burn_kernel<0>, dim3(cus), dim3(512), SMEM, st, iters, (const v4i *)operands, sink);
        else hipLaunchKernelGGL(mfma_burn_kernel<1>, dim3(cus), dim3(512), SMEM, st, iters, (const v4i *)operands, sink);
    };
    for (int i = 0; i < 3; ++i) launch();        // reach the steady clock before the timed region
    (void)hipEventRecord(e0, st);
    for (int i = 0; i < launches; ++i) launch();
    (void)hipEventRecord(e1, st);
    rc = check_launch("mfma_burn");
    float ms = 0.0f;
    if (rc == MQ_OK && (hipEventSynchronize(e1) != hipSuccess || hipEventElapsedTime(&ms, e0, e1) != hipSuccess)) rc = fail(MQ_EUNSUPPORTED, "mq_bench_mfma_burn: timing failed");
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    if (rc != MQ_OK) return rc;
    const double per_wave = (kind == 0) ? 16.0 * 2.0 * 32 * 32 * 32 : 32.0 * 2.0 * 16 * 16 * 64;
    *ops_per_s = (double)cus * 8.0 * iters * per_wave * launches / (ms * 1e-3);
    return MQ_OK;
}
