// Probe: in which order does V_MFMA_F32_16X16X4_F32 accumulate its 4 k-terms?
// Prints D[0][0] for value patterns whose result depends on the summation order.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float v4f __attribute__((ext_vector_type(4)));
__global__ void k(const float *bk, const float *c0, float *out, int nsteps)
{
    const int lane = threadIdx.x;
    v4f acc = {c0[0], c0[0], c0[0], c0[0]};
    for (int s = 0; s < nsteps; ++s) {
        const float a = 1.0f;
        const float b = bk[s * 4 + (lane >> 4)];
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc, 0, 0, 0);
    }
    if (lane == 0) out[0] = acc[0];
}
static float run(const float *h, int n, float c)
{
    float *d, *dc, *o, r;
    hipMalloc(&d, n * 4); hipMalloc(&dc, 4); hipMalloc(&o, 4);
    hipMemcpy(d, h, n * 4, hipMemcpyHostToDevice);
    hipMemcpy(dc, &c, 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, dc, o, n / 4);
    hipMemcpy(&r, o, 4, hipMemcpyDeviceToHost);
    hipFree(d); hipFree(dc); hipFree(o);
    return r;
}
static float seq(const float *h, int n, float c) { volatile float a = c; for (int i = 0; i < n; ++i) a = a + h[i]; return a; }
static float rev4(const float *h, int n, float c) { volatile float a = c; for (int s = 0; s < n; s += 4) for (int i = 3; i >= 0; --i) a = a + h[s + i]; return a; }
static float tree4(const float *h, int n, float c) { volatile float a = c; for (int s = 0; s < n; s += 4) { volatile float p = h[s] + h[s+1]; volatile float q = h[s+2] + h[s+3]; volatile float t = p + q; a = a + t; } return a; }
static float first4(const float *h, int n, float c) { volatile float a = c; for (int s = 0; s < n; s += 4) { volatile float t = h[s]; t = t + h[s+1]; t = t + h[s+2]; t = t + h[s+3]; a = a + t; } return a; }
int main()
{
    const float pats[][8] = {
        {1e8f, 1.f, -1e8f, 1.f, 0, 0, 0, 0},
        {1.f, 1e8f, 1.f, -1e8f, 0, 0, 0, 0},
        {1e8f, -1e8f, 1.f, 1.f, 0, 0, 0, 0},
        {1.f, 1.f, 1e8f, -1e8f, 0, 0, 0, 0},
        {16777216.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f},
        {1.f, 1.f, 1.f, 16777216.f, 1.f, 1.f, 1.f, 1.f},
        {0.1f, 0.2f, 0.3f, 0.4f, 0.5f, 0.6f, 0.7f, 0.8f},
    };
    for (unsigned i = 0; i < sizeof(pats) / sizeof(pats[0]); ++i) {
        for (float c : {0.0f, 3.0f}) {
            float g = run(pats[i], 8, c);
            printf("pat %u c=%g: gpu=%.9g seq=%.9g rev4=%.9g tree4=%.9g first4=%.9g\n", i, c, g,
                   seq(pats[i], 8, c), rev4(pats[i], 8, c), tree4(pats[i], 8, c), first4(pats[i], 8, c));
        }
    }
    // random stress: which model matches bitwise over many trials
    int ok[4] = {0, 0, 0, 0}, trials = 2000;
    srand(1);
    for (int t = 0; t < trials; ++t) {
        float h[40];
        for (int i = 0; i < 40; ++i) h[i] = ((rand() % 20001) - 10000) * 1e-3f * ((rand() & 1) ? 1.f : 37.f);
        float g = run(h, 40, 0.0f);
        ok[0] += (g == seq(h, 40, 0.0f)); ok[1] += (g == rev4(h, 40, 0.0f));
        ok[2] += (g == tree4(h, 40, 0.0f)); ok[3] += (g == first4(h, 40, 0.0f));
    }
    printf("random K=40: seq %d rev4 %d tree4 %d first4 %d of %d\n", ok[0], ok[1], ok[2], ok[3], trials);
    return 0;
}
