// Probe: does V_MFMA_F32_32X32X2_F32 accumulate its 2 k-terms in ascending k (a sequential fp32
// chain across consecutive instructions), like the 16x16x4 form?  And how fast does it issue?
// A: 32 rows x 2 k: lane l holds A[row = l % 32][k = l / 32];  B: 2 k x 32 cols: lane l holds
// B[k = l / 32][col = l % 32];  D (16 VGPRs): lane l, reg r -> row = 8*(r/4) + 4*(l/32)... + r%4, col = l % 32.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float v16f __attribute__((ext_vector_type(16)));
__global__ void k(const float *bk, float *out, int nsteps)
{
    const int lane = threadIdx.x;
    v16f acc;
    for (int i = 0; i < 16; ++i) acc[i] = 0.0f;
    for (int s = 0; s < nsteps; ++s) {
        const float a = 1.0f;
        const float b = bk[s * 2 + (lane >> 5)];
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
    }
    if (lane == 0) out[0] = acc[0];
}
static float run(const float *h, int n)
{
    float *d, *o, r;
    (void)hipMalloc(&d, n * 4); (void)hipMalloc(&o, 4);
    (void)hipMemcpy(d, h, n * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, o, n / 2);
    (void)hipMemcpy(&r, o, 4, hipMemcpyDeviceToHost);
    (void)hipFree(d); (void)hipFree(o);
    return r;
}
static float seq(const float *h, int n) { volatile float a = 0; for (int i = 0; i < n; ++i) a = a + h[i]; return a; }
static float rev2(const float *h, int n) { volatile float a = 0; for (int s = 0; s < n; s += 2) { a = a + h[s + 1]; a = a + h[s]; } return a; }
static float pair2(const float *h, int n) { volatile float a = 0; for (int s = 0; s < n; s += 2) { volatile float t = h[s] + h[s + 1]; a = a + t; } return a; }
int main()
{
    int ok[3] = {0, 0, 0}, trials = 2000;
    srand(1);
    for (int t = 0; t < trials; ++t) {
        float h[40];
        for (int i = 0; i < 40; ++i) h[i] = ((rand() % 20001) - 10000) * 1e-3f * ((rand() & 1) ? 1.f : 37.f);
        const float g = run(h, 40);
        ok[0] += (g == seq(h, 40)); ok[1] += (g == rev2(h, 40)); ok[2] += (g == pair2(h, 40));
    }
    printf("32x32x2 f32, random K=40: sequential %d, reversed-in-pair %d, pair-then-add %d of %d\n", ok[0], ok[1], ok[2], trials);
    return 0;
}
