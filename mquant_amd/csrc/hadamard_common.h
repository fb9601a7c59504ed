// hadamard_common.h -- argument block and the row-chunk helpers shared by the online-Hadamard kernels
// (hadamard.hip: the exact kernel; hadamard_fast.hip: the non-default fp16-matrix-core K x K stage).
#pragma once
#include "mq_common.h"

#ifndef MQ_HAD_X4
#define MQ_HAD_X4 1   // lane ^ 4 exchange of the butterflies: 0 = two DPP rotations + select, 1 = ds_swizzle (measured a little faster)
#endif

namespace mq {

struct HadArgs {
    const void *x;
    const void *x2 = nullptr;   // second operand of a fused activation (up of silu(gate)*up), same ldx
    int act = MQ_ACT_NONE;      // activation applied to the loaded row before the transform
    long M, n_in, ldx, n;
    int K, m;
    const uint8_t *had_bits;
    const unsigned long long *masks = nullptr;   // prepared descriptor (mq_hadamard_prepare): lane masks of the sign operand
    const v4i *hfrag = nullptr;                  // prepared descriptor: +-1 half operand images of the fast mode, [ceil(K/32)][ceil(K/16)][64]
    int unit_j = 1, unit_g = 4;                  // K x K stage: a wave owns unit_j 16-row tiles x unit_g 16-column tiles
    int parts_log2 = 0;                          // short batches: 2^parts_log2 workgroups share a row's K x K units (set by the launcher)
    int fp32_had;
    void *out;
    long ldo;
    float s0, s1;
    const uint8_t *row_sel;
    int skip_col0;
    float *x0_out;
    int8_t *qout;
    long K_pad, ldq;
    int vec_ok;
    int vec_ok2;       // same for x2
    int vec_ok_q;      // 16-byte stores into the int8 output are aligned
    float inv_sqrt_n;  // 1.0f / sqrtf((float)n), computed on the host in IEEE fp32
    int row_bytes;     // LDS bytes per k-row of the staged copy
    int swz;           // XOR-swizzle odd k-rows by 128 B (row_bytes % 256 == 0)
    int y_bytes;       // LDS bytes of the staged row
};

// One 512-element chunk of a row, 8 consecutive elements per lane starting at ``idx``: the global loads
// (zero beyond n_in: the pad of fake_quant/utils.py:465-471) and, with ACT, the fused activation
// silu(x) * x2 / quick_gelu(x) evaluated like the torch ops (mq_common.h).
template <int DT, bool ACT>
__device__ __forceinline__ void had_load_chunk(const HadArgs &p, long row, long idx, float (&vb)[8])
{
    typedef typename Elem<DT>::T T;
    const T *xr = reinterpret_cast<const T *>(p.x) + row * p.ldx;
    if (idx + 8 <= p.n_in && p.vec_ok) {
        if (sizeof(T) == 2) {
            const v8us a = *reinterpret_cast<const v8us *>(xr + idx);
#pragma unroll
            for (int i = 0; i < 8; ++i) vb[i] = Elem<DT>::ld((T)a[i]);
        } else {
            const v4f a = *reinterpret_cast<const v4f *>((const float *)xr + idx);
            const v4f b = *reinterpret_cast<const v4f *>((const float *)xr + idx + 4);
#pragma unroll
            for (int i = 0; i < 4; ++i) { vb[i] = a[i]; vb[4 + i] = b[i]; }
        }
    } else {
#pragma unroll
        for (int i = 0; i < 8; ++i) vb[i] = (idx + i < p.n_in) ? Elem<DT>::ld(xr[idx + i]) : 0.0f;
    }
    if (ACT && idx < p.n_in) {                            // fused activation prologue (own instantiation)
        float ub[8];
        if (p.act == MQ_ACT_SILU_MUL) {
            const T *ur = reinterpret_cast<const T *>(p.x2) + row * p.ldx;
            if (idx + 8 <= p.n_in && p.vec_ok2) {
                if (sizeof(T) == 2) {
                    const v8us a = *reinterpret_cast<const v8us *>(ur + idx);
#pragma unroll
                    for (int i = 0; i < 8; ++i) ub[i] = Elem<DT>::ld((T)a[i]);
                } else {
                    const v4f a = *reinterpret_cast<const v4f *>((const float *)ur + idx);
                    const v4f b = *reinterpret_cast<const v4f *>((const float *)ur + idx + 4);
#pragma unroll
                    for (int i = 0; i < 4; ++i) { ub[i] = a[i]; ub[4 + i] = b[i]; }
                }
            } else {
#pragma unroll
                for (int i = 0; i < 8; ++i) ub[i] = (idx + i < p.n_in) ? Elem<DT>::ld(ur[idx + i]) : 0.0f;
            }
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (idx + i >= p.n_in) continue;
            vb[i] = (p.act == MQ_ACT_SILU_MUL) ? act_silu_mul<DT>(vb[i], ub[i]) : act_quick_gelu<DT>(vb[i]);
        }
    }
}

// One butterfly stage between lanes l and l ^ LM (LM = 1, 2, 4, 8) for the 8 registers of a lane.  Lower lane of a pair:
// a + b; upper lane: a - b seen from the lower one = other + (-own): other + (+-own) with its ONE rounding either way.
// The partner's value comes through the DPP lane network (lane ^ 1, ^ 2: quad permutes; ^ 8: rotation by 8 inside the row
// of 16; ^ 4: the two rotations by 4, selected by the lane's bit) -- plain vector-ALU operands, where ds_bpermute sends
// every exchange through the LDS crossbar: the 32 exchanges per chunk bound phase A of the kernel
// (profiles/r4_hadamard_cu_timeline.txt).
template <int LM>
__device__ __forceinline__ void had_lane_stage(float (&v)[8], int lane)
{
    const unsigned flip = (lane & LM) ? 0x80000000u : 0u;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int own = __float_as_int(v[i]);
        int o;
        if (LM == 1) {
            o = __builtin_amdgcn_update_dpp(0, own, 0xB1, 0xf, 0xf, true);          // quad_perm [1,0,3,2]
        } else if (LM == 2) {
            o = __builtin_amdgcn_update_dpp(0, own, 0x4E, 0xf, 0xf, true);          // quad_perm [2,3,0,1]
        } else if (LM == 4) {
#if MQ_HAD_X4 == 1
            o = __builtin_amdgcn_ds_swizzle(own, 0x101f);                           // lane ^ 4 (bit mode)
#else
            const int up = __builtin_amdgcn_update_dpp(0, own, 0x124, 0xf, 0xf, true);   // row_ror:4
            const int dn = __builtin_amdgcn_update_dpp(0, own, 0x12C, 0xf, 0xf, true);   // row_ror:12
            o = (lane & 4) ? up : dn;
#endif
        } else {
            o = __builtin_amdgcn_update_dpp(0, own, 0x128, 0xf, 0xf, true);         // row_ror:8
        }
        v[i] = __int_as_float(o) + __uint_as_float((unsigned)own ^ flip);
    }
}

// Butterflies of one chunk in the reference's ascending-stride (a + b, a - b) order: strides 1, 2, 4 inside
// the lane's 8 registers, strides 8 .. min(m, 512) / 2 by wavefront shuffles (lane ^ stride / 8); when the
// whole co-factor fits the chunk (m <= 512) also * 1 / sqrt(n) and the cast the FHT extension performs.
template <int DT>
__device__ __forceinline__ void had_butterfly_chunk(float (&v)[8], int lane, int m, float scale, bool mid_round)
{
#pragma unroll
    for (int h = 1; h < 8; h <<= 1) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if ((i & h) == 0) {
                const float a0 = v[i], a1 = v[i + h];
                v[i] = a0 + a1;
                v[i + h] = a0 - a1;
            }
        }
    }
    // strides 8 .. 64 (lane ^ 1 .. ^ 8) on the DPP lane network, one straight-line stage each; 128, 256 (m = 256, 512) by
    // wavefront shuffle
    if (m > 8) had_lane_stage<1>(v, lane);
    if (m > 16) had_lane_stage<2>(v, lane);
    if (m > 32) had_lane_stage<4>(v, lane);
    if (m > 64) had_lane_stage<8>(v, lane);
    for (int h = 128; h < m && h < 512; h <<= 1) {
        const int lm = h >> 3;
        const float sgn = (lane & lm) ? -1.0f : 1.0f;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const float o = __shfl_xor(v[i], lm);
            v[i] = __builtin_fmaf(v[i], sgn, o);
        }
    }
    if (m <= 512) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            float t = v[i] * scale;
            if (mid_round) t = Elem<DT>::rnd(t);
            v[i] = t;
        }
    }
}

// Byte distance between the int8 outputs (row, j * m + c) and (row, (j + 1) * m + c) for a co-factor m >= 64:
// m in a row-major matrix; m / 64 pieces of 1 KiB in the tiled layout (act_offset, mq_common.h).
__device__ __forceinline__ long had_out_stride(const HadArgs &p)
{
    return p.ldq == MQ_LD_TILED ? 16L * p.m : (long)p.m;
}

// hadamard_fast.hip: MQ_EUNSUPPORTED (no message) = shape / dtype outside the fast mode, run the exact kernel
int hadamard_fast_dispatch(const HadArgs &p, int x_dtype, bool quant, hipStream_t st);

}  // namespace mq
