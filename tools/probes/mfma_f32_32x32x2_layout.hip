// Probe: operand and result layout of V_MFMA_F32_32X32X2_F32 (A: lane l = row l%32, k l/32;
// B: lane l = k l/32, col l%32; D: reg r of lane l = row 8*(r/4) + 4*(l/32) + r%4, col l%32).
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float v16f __attribute__((ext_vector_type(16)));
__global__ void k(float *rows, float *cols)
{
    const int l = threadIdx.x;
    v16f z;
    for (int i = 0; i < 16; ++i) z[i] = 0.0f;
    // D = row index: A[row][0] = row, A[row][1] = 0; B[0][col] = 1, B[1][col] = 1
    v16f d1 = __builtin_amdgcn_mfma_f32_32x32x2f32((l < 32) ? (float)(l & 31) : 0.0f, 1.0f, z, 0, 0, 0);
    // D = col index: A = 1 (k=0), 0 (k=1); B[0][col] = col
    v16f d2 = __builtin_amdgcn_mfma_f32_32x32x2f32((l < 32) ? 1.0f : 0.0f, (float)(l & 31), z, 0, 0, 0);
    for (int r = 0; r < 16; ++r) { rows[l * 16 + r] = d1[r]; cols[l * 16 + r] = d2[r]; }
}
int main()
{
    float *dr, *dc, hr[1024], hc[1024];
    (void)hipMalloc(&dr, 4096); (void)hipMalloc(&dc, 4096);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dr, dc);
    (void)hipMemcpy(hr, dr, 4096, hipMemcpyDeviceToHost); (void)hipMemcpy(hc, dc, 4096, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int l = 0; l < 64; ++l)
        for (int r = 0; r < 16; ++r) {
            const int row = 8 * (r / 4) + 4 * (l / 32) + (r % 4), col = l % 32;
            if (hr[l * 16 + r] != (float)row || hc[l * 16 + r] != (float)col) ++bad;
        }
    printf("layout mismatches: %d of 1024 (lane 33: rows", bad);
    for (int r = 0; r < 16; ++r) printf(" %g", hr[33 * 16 + r]);
    printf(")\n");
    return 0;
}
