// Probe: semantics and bank behaviour of ds_read_b64_tr_b16 (gfx950) for a row-major [k][i] fp16 image.
//   part 1: every lane reads 8 bytes at lds + lane * 8; prints which source halfs each lane receives
//   part 2: the 16-lane group reads a [4 k][16 i] block of a [k][i] image with row stride S halfs; checks that
//           lane t receives y[k0..k0+3][i0 + t]
//   part 3: cycles per read for row strides / swizzles (a wave issuing 64 reads back to back, 4 waves per CU)
// build: hipcc --offload-arch=gfx950 -O2 -o ds_read_tr.bin ds_read_tr.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef short v4s __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) v4s lds_v4s;

__global__ void part1(unsigned short *out)
{
    __shared__ __attribute__((aligned(16))) unsigned short lds[1024];
    for (int i = threadIdx.x; i < 1024; i += 64) lds[i] = (unsigned short)i;
    __syncthreads();
    const int l = threadIdx.x;
    v4s r = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4s *)(lds + l * 4));
    for (int e = 0; e < 4; ++e) out[l * 4 + e] = (unsigned short)r[e];
}

__global__ void part2(unsigned short *out, int S)
{
    extern __shared__ __attribute__((aligned(16))) unsigned short lds2[];
    for (int i = threadIdx.x; i < 8 * S; i += 64) lds2[i] = (unsigned short)((i / S) * 1000 + (i % S));   // value = k * 1000 + i
    __syncthreads();
    const int l = threadIdx.x, t = l & 15, g = l >> 4;
    // group g: i0 = 16 * (g & 1), k0 = 4 * (g >> 1)
    const unsigned short *p = lds2 + (4 * (g >> 1) + t / 4) * S + 16 * (g & 1) + 4 * (t % 4);
    v4s r = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4s *)p);
    for (int e = 0; e < 4; ++e) out[l * 4 + e] = (unsigned short)r[e];
}

// mode: row stride in bytes and swizzle kind; 16 k-rows x 128 i image; each wave reads 32x16 fragments repeatedly
__global__ void part3(long long *cycles, int *sink, int S_bytes, int swz, int iters)
{
    extern __shared__ __attribute__((aligned(16))) char lds3[];
    const int l = threadIdx.x & 63;
    const int t = l & 15, g = (l >> 4) & 1, ko = l >> 5;     // fragment: i = 16 g + t, k-octet ko
    int acc = 0;
    __syncthreads();
    const long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int tile = 0; tile < 4; ++tile) {
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int k = 8 * ko + 4 * q + t / 4;
                int col = (32 * tile + 16 * g + 4 * (t % 4)) * 2;
                if (swz == 1) col ^= (k & 1) << 7;
                if (swz == 2) col ^= (k & 3) << 5;
                if (swz == 3) col ^= ((k & 3) << 5) ^ (((k >> 2) & 1) << 7);
                v4s r = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4s *)(lds3 + k * S_bytes + col));
                acc += r[0] + r[1] + r[2] + r[3];
            }
        }
    }
    const long long t1 = clock64();
    if (l == 0) cycles[blockIdx.x * (blockDim.x / 64) + (threadIdx.x >> 6)] = t1 - t0;
    sink[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

int main()
{
    unsigned short *out; hipMalloc(&out, 64 * 4 * 2);
    std::vector<unsigned short> h(256);
    part1<<<1, 64>>>(out);
    hipMemcpy(h.data(), out, 512, hipMemcpyDeviceToHost);
    printf("part 1: lane -> the four source half indices it received (each lane read lds + lane*4 halfs)\n");
    for (int l = 0; l < 64; ++l) printf("  lane %2d: %4d %4d %4d %4d\n", l, h[l * 4], h[l * 4 + 1], h[l * 4 + 2], h[l * 4 + 3]);
    for (int S : {128, 136}) {
        part2<<<1, 64, 8 * S * 2>>>(out, S);
        hipMemcpy(h.data(), out, 512, hipMemcpyDeviceToHost);
        int bad = 0;
        for (int l = 0; l < 64; ++l) {
            const int t = l & 15, g = l >> 4;
            for (int e = 0; e < 4; ++e) {
                const int want = (4 * (g >> 1) + e) * 1000 + 16 * (g & 1) + t;
                if (h[l * 4 + e] != want) { if (bad < 8) printf("  S=%d lane %d e %d got %d want %d\n", S, l, e, h[l * 4 + e], want); ++bad; }
            }
        }
        printf("part 2 (row stride %d halfs): %s (%d mismatches) -- lane t of a group gets y[k0..k0+3][i0+t]\n", S, bad ? "MISMATCH" : "as expected", bad);
    }
    long long *cyc; int *sink; hipMalloc(&cyc, 4096 * 8); hipMalloc(&sink, 256 * 256 * 4);
    std::vector<long long> hc(4096);
    const int iters = 200;
    struct { int S, swz; const char *name; } modes[] = {{256, 0, "stride 256 B, no swizzle"}, {256, 1, "stride 256 B, ^(k&1)<<7"},
        {256, 2, "stride 256 B, ^(k&3)<<5"}, {256, 3, "stride 256 B, ^(k&3)<<5 ^ (k>>2&1)<<7"}, {272, 0, "stride 272 B, no swizzle"}, {288, 0, "stride 288 B"}, {320, 0, "stride 320 B"}};
    for (auto &m : modes) {
        for (int waves : {1, 4, 8}) {
            part3<<<256, waves * 64, 16 * 320>>>(cyc, sink, m.S, m.swz, iters);
            hipDeviceSynchronize();
            hipMemcpy(hc.data(), cyc, 256 * waves * 8, hipMemcpyDeviceToHost);
            double s = 0; for (int i = 0; i < 256 * waves; ++i) s += hc[i];
            printf("part 3: %-44s %d waves/CU: %.1f cycles per ds_read_b64_tr_b16 per wave\n", m.name, waves, s / (256 * waves) / (iters * 8.0));
        }
    }
    return 0;
}
