// attn_prefill.hip -- prefill attention, three instantiations of one kernel:
//   * mq_attn_prefill_fp8kv: consumes the fp8 (OCP e4m3fn) KV cache DIRECTLY -- K and V leave HBM as one byte per element
//     and are widened inside the kernel; no dequantise-on-read pass, no fp16 copy of the cache in HBM (SURVEY 8(f4),
//     BASELINE configuration 5);
//   * mq_attn_prefill: the same dataflow over unquantised 16-bit K / V, head_dim 128 (decoder, causal) or 80 (Qwen2-VL's
//     vision tower), q / k / v read in place from the fused q|k|v GEMM output (glue of the whole-prefill report, 8(f3));
//   * mq_attn_prefill_quant_i8: either of them with the NEXT Linear's static int8 activation quantizer fused into the store.
// The reference has neither a KV-cache quantizer nor an attention kernel of its own (fake_quant/utils.py:220-267 are flags
// of an unused parser; attention is HF model code): PARITY UNPINNED -- the checker is float64 softmax attention (over the
// dequantised cache for the e4m3 variant), restated in tests/test_gpu_attn_prefill.py.
//
//     S[q][k] = (sum_d Q[q][d] * K8[k][d]) * s_k[kvh] * softmax_scale        (causal: k <= q)
//     O[q][d] = (sum_k softmax_k(S)[q][k] * V8[k][d]) * s_v[kvh]
// The per-head cache scales are scalars of a (head, kv-head) pair, so they fold into the score scale and the output
// scale: the matrix core multiplies the e4m3 VALUES (exact in fp16 / bf16), never a dequantised tensor.
//
// A prefill of a few hundred tokens is a LATENCY problem (4 GFLOP for the 7B model's 768 tokens): one workgroup =
// 32 query rows of one head, and its 4 waves SPLIT THE KEYS (wave w takes the 32-key blocks w, w + 4, ...), each with
// its own running softmax statistics, merged through LDS at the end -- 672 workgroups (504 with the shallow tiles paired)
// with a critical path of 6 blocks instead of 168 with a critical path of 24.  Waves never synchronise inside the key loop:
//   * S is computed TRANSPOSED (keys x queries, V_MFMA_F32_32X32X16_F16/_BF16 with A = K rows, B = Q^T): in the D
//     layout a lane then holds ONE query column and 16 keys in registers, so the softmax statistics of a query are
//     lane-local (one exchange with lane + 32 for the other half of the keys) instead of 5-step shuffles per row;
//   * the K operand never touches LDS: a lane reads 64 contiguous bytes of its key row straight from the cache and
//     widens them in registers (V_CVT_SCALEF32_PK_F16_FP8, two values per instruction); the contraction over d is
//     order-agnostic, so lane half ko simply owns d = 64 ko .. 64 ko + 63 and Q^T is loaded to match;
//   * P is packed to half precision in registers; two v_permlane32_swap per 16 keys turn the D layout into the B
//     operand of the second GEMM, O^T[d][q] += V^T[d][k] P^T[k][q], whose A operand comes out of a wave-private
//     row-major V tile in LDS through ds_read_b64_tr_b16 (hardware transpose read: a 16-lane group reads a
//     [4 k][16 d] block, lane t receives V[k0..k0+3][d0 + t]; tools/probes/ds_read_tr.hip);
//   * the next block's K and V bytes are in flight (registers) while the current block is multiplied; the running
//     output is rescaled only when some query's maximum actually moved.
// Round 6 (profiles/r6_attention_rework.txt): K / V arrive through buffer descriptors (hardware range check, one 32-bit offset per
// lane); the grid is (heads, rows of query tiles) with the deepest tiles of every head first, the second round of a causal launch
// reversed, and -- where that brings a causal grid into one round of slots -- the shallow half of the tiles paired two to a
// workgroup; launches beyond one round take 2-wave workgroups (attn_launch_t).
#include "mq_common.h"
#include <type_traits>

namespace mq {

typedef short at_v4s __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) at_v4s at_lds_v4s;
typedef _Float16 at_v8h __attribute__((ext_vector_type(8)));
typedef __bf16 at_v8bf __attribute__((ext_vector_type(8)));
typedef float at_v16f __attribute__((ext_vector_type(16)));

struct AttnArgs {
    const void *q;           // [T, heads * 128] (row stride ldq elements), fp16 / bf16
    const uint8_t *k, *v;    // first K / V head of token 0; a token's heads are contiguous, row stride ldkv BYTES.  e4m3 cache:
                             // v = k + kv_heads * 128 ([T, 2 * kv_heads, 128]); 16-bit K / V: two column slices of the q|k|v output
    const float *kv_scale;   // e4m3 only: [2 * kv_heads]
    void *out;               // [T, heads * 128] (row stride ldo elements), q's dtype
    long T, ldq, ldkv, ldo;
    int heads, kv_heads, causal;
    float softmax_scale;
    // optional: emit the int8 levels of the NEXT Linear's static activation quantizer instead of 16-bit values (the
    // o_proj / proj input): q = clamp(rint(cast(o) / s), -128, 127), s = qs1 where row_sel[row] != 0 else qs0, in the
    // activation layout (row-major with leading dimension q_ld, or MQ_LD_TILED) -- what act_quant_kernel writes for `out`
    int8_t *qout;
    long q_kpad, q_ld;
    float qs0, qs1;
    const uint8_t *row_sel;
    int deep_rows;           // grid rows that are ONE query tile (the deepest ones); the rows after them carry two (see the kernel)
    float qi0, qi1;          // 1 / qs0, 1 / qs1 (IEEE, from the host) and whether the reciprocal form may be used (quant_rcp_ok)
    int qr0, qr1;
};

template <int DT> struct AttnMma;
template <> struct AttnMma<MQ_F16> {
    static __device__ __forceinline__ at_v16f mma(v4i a, v4i b, at_v16f c)
    {
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(at_v8h, a), __builtin_bit_cast(at_v8h, b), c, 0, 0, 0);
    }
};
template <> struct AttnMma<MQ_BF16> {
    static __device__ __forceinline__ at_v16f mma(v4i a, v4i b, at_v16f c)
    {
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(at_v8bf, a), __builtin_bit_cast(at_v8bf, b), c, 0, 0, 0);
    }
};

constexpr int AT_KB = 32;            // keys per block
constexpr int AT_STATS = 4 * 32 * 2 * 4;   // [waves <= 4][32 queries][m, l]
// Per head dimension HD (128: Qwen2-VL decoder; 80: its vision tower, 16-bit K / V only): the V tile is [32 keys][HD] 16-bit
// values, rows padded to whole 32-column d-tiles.  HD = 128: 256-byte rows, 32-byte groups XOR-ed by key & 3; HD = 80: 192-byte
// rows (the four rows of a transpose read already fall into four different 64-byte bank slots).  The 16-bit K tile has rows
// of 2 HD + 16 bytes (an odd number of 16-byte slots: conflict-free ds_read_b128 with lane = key).
template <int HD> struct AttnGeo {
    static constexpr int NKS = HD / 16;                       // k-steps of the first GEMM
    static constexpr int NDT = (HD + 31) / 32;                // 32-row d-tiles of the second GEMM
    static constexpr int HALF = HD / 2;                       // d values of one lane half
    static constexpr int VROW = NDT * 64;                     // bytes per V row in LDS
    static constexpr int VBYTES = 32 * VROW;
    static constexpr int KROW = 2 * HD + 16;
    static constexpr int OBYTES = NDT * 16 * 64 * 4;          // a wave's partial O^T in the merge
    static constexpr int LOOP = VBYTES + 32 * KROW;
    static constexpr int WAVE_LDS = LOOP > OBYTES ? LOOP : OBYTES;
    static __device__ __forceinline__ int vswz(int key) { return HD == 128 ? (key & 3) << 5 : 0; }
};

template <int DT> struct AttnCvt;
typedef float at_v2f __attribute__((ext_vector_type(2)));
template <> struct AttnCvt<MQ_F16> {
    typedef _Float16 h2 __attribute__((ext_vector_type(2)));
    static __device__ __forceinline__ unsigned pack2(float a, float b) { return __builtin_bit_cast(unsigned, __builtin_convertvector(at_v2f{a, b}, h2)); }   // V_CVT_PK_F16_F32, RNE
    static __device__ __forceinline__ int lo(int w) { return __builtin_bit_cast(int, __builtin_amdgcn_cvt_scalef32_pk_f16_fp8(w, 1.0f, false)); }
    static __device__ __forceinline__ int hi(int w) { return __builtin_bit_cast(int, __builtin_amdgcn_cvt_scalef32_pk_f16_fp8(w, 1.0f, true)); }
};
template <> struct AttnCvt<MQ_BF16> {
    typedef __bf16 b2 __attribute__((ext_vector_type(2)));
    static __device__ __forceinline__ unsigned pack2(float a, float b) { return __builtin_bit_cast(unsigned, __builtin_convertvector(at_v2f{a, b}, b2)); }   // V_CVT_PK_BF16_F32, RNE
    static __device__ __forceinline__ int lo(int w) { return __builtin_bit_cast(int, __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(w, 1.0f, false)); }
    static __device__ __forceinline__ int hi(int w) { return __builtin_bit_cast(int, __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(w, 1.0f, true)); }
};

// sixteen e4m3 bytes -> two operands of eight 16-bit values (exact: e4m3 has 3 mantissa bits)
template <int DT>
__device__ __forceinline__ void widen16(const v4i w, v4i &a, v4i &b)
{
    a = v4i{AttnCvt<DT>::lo(w[0]), AttnCvt<DT>::hi(w[0]), AttnCvt<DT>::lo(w[1]), AttnCvt<DT>::hi(w[1])};
    b = v4i{AttnCvt<DT>::lo(w[2]), AttnCvt<DT>::hi(w[2]), AttnCvt<DT>::lo(w[3]), AttnCvt<DT>::hi(w[3])};
}

// KV8: K / V are e4m3 bytes (widened here); else they are 16-bit values of q's dtype, used as they are.
// NW = waves per workgroup = ways the keys are split (4, or 2: half the LDS, so four workgroups fit a CU and a prefill whose
// 4-wave grid needs a second round -- 672 workgroups on 512 slots at the 7B shape -- is resident at once).
template <int DT, bool KV8, int HD, int NW>
__global__ __launch_bounds__(NW * 64, 2) void attn_prefill_kernel(AttnArgs p)
{
    kernarg_warm<sizeof(AttnArgs), true>();                           // one scalar-load round trip for the argument block (mq_common.h)
    typedef AttnGeo<HD> G;
    static_assert(HD == 128 || (HD == 80 && !KV8), "head dimensions built: 128, and 80 for 16-bit K / V");
    constexpr int NKS = G::NKS, NDT = G::NDT, AT_D = HD, AT_VROW = G::VROW, AT_KROW = G::KROW, AT_WAVE_LDS = G::WAVE_LDS;
    constexpr int NR = KV8 ? 4 : (HD == 80 ? 5 : 8);                  // 16-byte loads per lane and operand and block
    typedef AttnMma<DT> MM;
    __shared__ __attribute__((aligned(16))) char smem[NW * AT_WAVE_LDS + AT_STATS];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // grid = (heads, rows of query tiles): workgroups are handed out x first, so the deepest (last) query tiles of EVERY head
    // start before any shallow one (causal: a tile's work grows with its index; with the tiles in x the late heads' deep tiles
    // waited for a second round of slots and set the launch time -- 23.2 -> 20.3 us at the 7B shape).
    // Causal launches of 4-wave workgroups PAIR the shallow half of the tiles: rows [0, deep_rows) are the deepest tiles, keys
    // split four ways; each row after them carries two tiles of the shallow half -- the deepest with the shallowest left, keys
    // split two ways by a pair of waves each -- so no wave walks more than a quarter of the deepest tile's blocks and 0.75 n rows
    // do the work of n (7B: 504 workgroups, all resident at once, instead of 672 on 512 slots).
    const int head = blockIdx.x, kvh = head / (p.heads / p.kv_heads);
    const long n_all = (p.T + AT_KB - 1) / AT_KB;
    long r = blockIdx.y;
    if (p.causal) {
        // The first 256 workgroups take the first slot of the 256 CUs, the next 256 the second: run that second round in reverse,
        // so that the CU holding the deepest tile gets the lightest row of the second round (every CU then carries about the same
        // number of key blocks).  Measured 20.4 -> 19.6 us (7B), 19.1 -> 17.5 us (e4m3) before the pairing.
        const long rows_a = 256 / gridDim.x;
        if (rows_a > 0 && 2 * rows_a <= (long)gridDim.y && r >= rows_a && r < 2 * rows_a) r = 3 * rows_a - 1 - r;
    }
    int nwe = NW, wig = wave;                                         // waves that share this wave's tile, and its index among them
    long qt = n_all - 1 - r;
    if (r >= p.deep_rows) {
        const long j = r - p.deep_rows, hi = n_all - p.deep_rows - 1 - j;
        qt = hi;
        if (hi != j) {                                                // (an odd shallow half leaves its middle tile alone)
            nwe = NW / 2;
            wig = wave & (NW / 2 - 1);
            if (wave >= NW / 2) qt = j;
        }
    }
    const long q_row = qt * 32 + (lane & 31);                         // the query this lane owns (D layout: lane = column)
    // the token-type flag of this row (fused int8 store), requested here rather than after the merge where it is used (a cold
    // byte load on the workgroup's tail; launch time unchanged within the harness's resolution -- other workgroups fill the CU)
    const unsigned sel = (p.qout && p.row_sel && q_row < p.T) ? p.row_sel[q_row] : 0;
    const int ko = lane >> 5;                                         // lane half: d 64 ko.. of K / Q, keys + 4 ko of S, octet ko of P
    const float sc = (KV8 ? p.kv_scale[kvh] : 1.0f) * p.softmax_scale * 1.4426950408889634f;   // K scale and log2(e) folded into the score scale
    const float s_v = KV8 ? p.kv_scale[p.kv_heads + kvh] : 1.0f;
    char *vt = smem + wave * AT_WAVE_LDS;                             // this wave's V tile: [32 keys][128 d] 16-bit, row-major, swizzled

    // ---- Q^T operand: lane = query; k-step ds covers d = 64 ko + 8 ds .. + 7 (the K operand is loaded to match) ----
    v4i Qf[NKS];
    {
        const unsigned short *qp = reinterpret_cast<const unsigned short *>(p.q) + q_row * p.ldq + (long)head * AT_D + G::HALF * ko;
#pragma unroll
        for (int ds = 0; ds < NKS; ++ds)
            Qf[ds] = (q_row < p.T) ? *reinterpret_cast<const v4i *>(qp + ds * 8) : v4i{0, 0, 0, 0};
    }

    at_v16f O[NDT];
#pragma unroll
    for (int dt = 0; dt < NDT; ++dt)
#pragma unroll
        for (int e = 0; e < 16; ++e) O[dt][e] = 0.0f;
    float m_run = -1.0e30f, l_run = 0.0f;                              // log2 domain

    const int n_blocks = (int)((p.causal && qt + 1 < n_all) ? qt + 1 : n_all);

    // A lane's loads for one block.  e4m3: K -- the 64 bytes d 64 ko.. of key (lane & 31), straight into its MFMA operands;
    // V -- the 64 bytes d 64 (lane & 1).. of key lane / 2.  16-bit K / V: rows are 256 bytes and a lane-per-row pattern
    // would touch 64 cache lines per instruction (measured: 55 us against 25 us for the e4m3 cache at the 7B shape), so
    // both tiles are loaded COALESCED -- instruction j = rows 4 j .. 4 j + 3, lane = (row, 16-byte piece) -- and K takes
    // the detour through a padded LDS tile as well.
    // Round 6: the loads go through buffer descriptors of this KV head's column slice ([0, (T - 1) ldkv + row bytes)): one
    // 32-bit offset per lane and block, rows past T come back as zeros from the hardware's range check -- no per-load
    // compare / branch and no sixteen 64-bit row pointers (the first form held 32 registers of addresses and the compiler,
    // out of registers, waited for every LDS operand right before its MFMA).
    const int v_key = KV8 ? lane >> 1 : lane >> 4, v_d = (lane & 1) * 64, pc16 = (lane & 15) * 16;
    char *kt = vt + G::VBYTES;
    constexpr unsigned ESZ = KV8 ? 1 : 2;
    const unsigned ldkv = (unsigned)p.ldkv, kv_bytes = (unsigned)(p.T - 1) * ldkv + HD * ESZ;   // host: (T + 32) ldkv < 2^32
    const __amdgpu_buffer_rsrc_t k_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t *>(p.k) + (long)kvh * (AT_D * ESZ), 0, kv_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t v_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t *>(p.v) + (long)kvh * (AT_D * ESZ), 0, kv_bytes, 0x00020000);
    // HD = 80: a row is ten 16-byte pieces, a block 320 of them -- five instructions with every lane busy (instruction j, lane l:
    // piece 64 j + l = row (64 j + l) / 10, piece (64 j + l) % 10) instead of eight with ten lanes in sixteen
    const unsigned off_k = KV8 ? (unsigned)(lane & 31) * ldkv + 64 * ko : (unsigned)v_key * ldkv + pc16;
    const unsigned off_v = KV8 ? (unsigned)v_key * ldkv + v_d : off_k;
    unsigned off80[HD == 80 ? 5 : 1];
    int lds80[HD == 80 ? 5 : 1];                                      // row * 16 + piece of the same five pieces
    if (HD == 80) {
#pragma unroll
        for (int j = 0; j < 5; ++j) {
            const int idx = 64 * j + lane, row = idx / 10, piece = idx - 10 * row;
            off80[j] = (unsigned)row * ldkv + 16 * piece;
            lds80[j] = row * 16 + piece;
        }
    }
    auto load_block = [&](int kb, v4i (&kraw)[NR], v4i (&vraw)[NR]) {
        const unsigned blk = (unsigned)kb * (AT_KB * ldkv);
#pragma unroll
        for (int j = 0; j < NR; ++j) {
            const unsigned step = KV8 ? 16 * j : 4 * j * ldkv;
            const unsigned ok = HD == 80 ? blk + off80[HD == 80 ? j : 0] : blk + off_k + step, ov = HD == 80 ? ok : blk + off_v + step;
            kraw[j] = __builtin_bit_cast(v4i, __builtin_amdgcn_raw_buffer_load_b128(k_rs, ok, 0, 0));
            vraw[j] = __builtin_bit_cast(v4i, __builtin_amdgcn_raw_buffer_load_b128(v_rs, ov, 0, 0));
        }
    };

    v4i kraw[NR], vraw[NR];
    if (wig < n_blocks) load_block(wig, kraw, vraw);
    const int t16 = lane & 15, g16 = (lane >> 4) & 1;
    for (int kb = wig; kb < n_blocks; kb += nwe) {
        // ---- K into MFMA operands, V into this wave's LDS tile (e4m3: widened on the way) ------------------------
        v4i Kf[8];
        if (KV8) {
#pragma unroll
            for (int j = 0; j < 4; ++j) widen16<DT>(kraw[j], Kf[2 * j], Kf[2 * j + 1]);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                v4i a, b;
                widen16<DT>(vraw[j], a, b);
                const int col = ((v_d + 16 * j) * 2) ^ ((v_key & 3) << 5);    // 32 B = sixteen values = one swizzle group
                *reinterpret_cast<v4i *>(vt + v_key * AT_VROW + col) = a;
                *reinterpret_cast<v4i *>(vt + v_key * AT_VROW + col + 16) = b;
            }
        } else if (HD == 80) {
#pragma unroll
            for (int j = 0; j < 5; ++j) {
                const int r = lds80[HD == 80 ? j : 0] >> 4, pc = (lds80[HD == 80 ? j : 0] & 15) * 16;
                *reinterpret_cast<v4i *>(kt + r * AT_KROW + pc) = kraw[j % NR];
                *reinterpret_cast<v4i *>(vt + r * AT_VROW + pc) = vraw[j % NR];
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int r = 4 * j + v_key;
                *reinterpret_cast<v4i *>(kt + r * AT_KROW + pc16) = kraw[j % NR];
                *reinterpret_cast<v4i *>(vt + r * AT_VROW + (pc16 ^ G::vswz(r))) = vraw[j % NR];
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
        if (kb + nwe < n_blocks) load_block(kb + nwe, kraw, vraw);      // in flight during this block's arithmetic
        const long key0 = (long)kb * AT_KB;

        // ---- S^T = K Q^T : [32 keys][32 queries] ----------------------------------------------------------------
        at_v16f S;
#pragma unroll
        for (int e = 0; e < 16; ++e) S[e] = 0.0f;
        if (!KV8) {                                                   // all operand reads first, then the MFMAs wait for them one by one
#pragma unroll
            for (int ds = 0; ds < NKS; ++ds)
                Kf[ds] = *reinterpret_cast<const v4i *>(kt + (lane & 31) * AT_KROW + (G::HALF * ko + 8 * ds) * 2);
        }
#pragma unroll
        for (int ds = 0; ds < NKS; ++ds) S = MM::mma(Kf[ds], Qf[ds], S);

        // ---- online softmax (log2 domain); register r <-> key key0 + (r & 3) + 8 (r >> 2) + 4 ko.  The maximum is taken
        // over the raw scores (sc > 0) and the scale rides in the exponent's fma; a masked score is -1e30 BEFORE scaling,
        // which exp2 turns into an exact 0 -- every row of a block this wave visits has at least one unmasked key (causal:
        // key0 <= q), so the running maximum is never the mask value and no 0 / 1 ambiguity arises.
        const bool edge = key0 + AT_KB > p.T || (p.causal && key0 + AT_KB - 1 > qt * 32);   // wave-uniform
        if (edge) {
            const int kmax = (int)((p.causal && q_row < p.T ? q_row : p.T - 1) - key0);     // last admissible key of this row, relative
#pragma unroll
            for (int r = 0; r < 16; ++r)
                if ((r & 3) + 8 * (r >> 2) + 4 * ko > kmax) S[r] = -1.0e30f;
        }
        float m_raw = fmaxf(S[0], S[1]);
#pragma unroll
        for (int r = 2; r < 16; ++r) m_raw = fmaxf(m_raw, S[r]);
        {                                                             // lane + 32 holds the other 16 keys of this query: (lo, lo) and (hi, hi)
            const auto mx = __builtin_amdgcn_permlane32_swap(__float_as_uint(m_raw), __float_as_uint(m_raw), false, false);
            m_raw = fmaxf(__uint_as_float(mx[0]), __uint_as_float(mx[1]));
        }
        const float m_new = fmaxf(m_run, m_raw * sc);
        float psum = 0.0f;
        unsigned pk[8];                                               // P as 16-bit pairs: pk[2 g + e2] = keys 8 g + 4 ko + 2 e2, + 1
#pragma unroll
        for (int r = 0; r < 16; r += 2) {
            const float p0 = __builtin_amdgcn_exp2f(fmaf(S[r], sc, -m_new));
            const float p1 = __builtin_amdgcn_exp2f(fmaf(S[r + 1], sc, -m_new));
            psum += p0 + p1;
            pk[r >> 1] = AttnCvt<DT>::pack2(p0, p1);
        }
        if (__any(m_new > m_run)) {                                   // some query's maximum moved: rescale the running output
            const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
            l_run *= alpha;
#pragma unroll
            for (int dt = 0; dt < NDT; ++dt)
#pragma unroll
                for (int e = 0; e < 16; ++e) O[dt][e] *= alpha;
            m_run = m_new;
        }
        l_run += psum;

        // ---- O^T += V^T P^T : 2 k-steps of 16 keys x 4 tiles of 32 d ---------------------------------------------
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");        // this wave's V tile stores before its transpose reads
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            // B operand (lane = query, octet ko of the k-step): keys 16 ks + 8 ko .. + 7 = group g = 2 ks + ko from BOTH
            // lane halves; the lower half trades its group 2 ks + 1 for the upper half's group 2 ks
            const auto x0 = __builtin_amdgcn_permlane32_swap(pk[4 * ks + 0], pk[4 * ks + 2], false, false);
            const auto x1 = __builtin_amdgcn_permlane32_swap(pk[4 * ks + 1], pk[4 * ks + 3], false, false);
            const v4i pf = v4i{(int)x0[0], (int)x1[0], (int)x0[1], (int)x1[1]};
            v4i Af[NDT];
#pragma unroll
            for (int dt = 0; dt < NDT; ++dt) {
                // A operand: row d = 32 dt + (lane & 31), keys 16 ks + 8 ko + 0..7, two transpose reads of [4 k][16 d]
                // (HD = 80: rows 80..95 of the last tile hold whatever the padding holds; an output row depends on its own
                //  operand row only, and those rows are never stored)
                const int d_lane = dt * 32 + 16 * g16 + 4 * (t16 & 3);
                const int kA = ks * 16 + 8 * ko + (t16 >> 2);
                const at_v4s r0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                    (at_lds_v4s *)(vt + kA * AT_VROW + ((d_lane * 2) ^ G::vswz(kA))));
                const at_v4s r1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                    (at_lds_v4s *)(vt + (kA + 4) * AT_VROW + ((d_lane * 2) ^ G::vswz(kA + 4))));
                const v2i lo = __builtin_bit_cast(v2i, r0), hi = __builtin_bit_cast(v2i, r1);
                Af[dt] = v4i{lo[0], lo[1], hi[0], hi[1]};
            }
#pragma unroll
            for (int dt = 0; dt < NDT; ++dt) O[dt] = MM::mma(Af[dt], pf, O[dt]);
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");        // ... and the reads before the next block's stores
        __builtin_amdgcn_wave_barrier();
    }

    // ---- merge the waves' partial results: wave w finishes the 32-row d-tiles w, w + NW, ... ----------------------------
    l_run += __shfl_xor(l_run, 32, 64);
    float *stats = reinterpret_cast<float *>(smem + NW * AT_WAVE_LDS);
    float *mine = reinterpret_cast<float *>(vt);                      // [dt][e][lane]
#pragma unroll
    for (int dt = 0; dt < NDT; ++dt)
#pragma unroll
        for (int e = 0; e < 16; ++e) mine[(dt * 16 + e) * 64 + lane] = O[dt][e];
    if (lane < 32) {
        stats[(wave * 32 + lane) * 2] = m_run;
        stats[(wave * 32 + lane) * 2 + 1] = l_run;
    }
    __syncthreads();
    auto finish = [&](auto nwe_c) {
    constexpr int NWE = decltype(nwe_c)::value;
    const int g0 = wave - wig;                                        // first wave of this tile
    float M = -1.0e30f;
#pragma unroll
    for (int w = 0; w < NWE; ++w) M = fmaxf(M, stats[((g0 + w) * 32 + (lane & 31)) * 2]);
    float L = 0.0f, fw[NWE];
#pragma unroll
    for (int w = 0; w < NWE; ++w) {
        fw[w] = __builtin_amdgcn_exp2f(stats[((g0 + w) * 32 + (lane & 31)) * 2] - M);
        L += stats[((g0 + w) * 32 + (lane & 31)) * 2 + 1] * fw[w];
    }
    const float f = L > 0.0f ? s_v / L : 0.0f;
    const float qs = sel ? p.qs1 : p.qs0, qinv = sel ? p.qi1 : p.qi0;
    const bool qrcp = (sel ? p.qr1 : p.qr0) != 0;
#pragma unroll
    for (int dt0 = 0; dt0 < NDT; dt0 += NWE) {
        const int dt = dt0 + wig;                                    // wave-uniform
        if (dt >= NDT) break;                                         // HD = 80: three d-tiles
        float acc[16];
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[e] = 0.0f;
#pragma unroll
        for (int w = 0; w < NWE; ++w) {
            const float *src = reinterpret_cast<const float *>(smem + (g0 + w) * AT_WAVE_LDS) + dt * 16 * 64 + lane;
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[e] += src[e * 64] * fw[w];
        }
        if (q_row >= p.T) continue;
        if (p.qout) {
            // the sixteen levels of this lane in one go (quant_levels_i8_packed: the levels of quant_levels by construction, one
            // rarely-taken branch, packed by v_perm; 1 / s and the reciprocal-form flag come from the host)
            float v16[16];
            unsigned w4[4];
#pragma unroll
            for (int e = 0; e < 16; ++e) v16[e] = Elem<DT>::rnd(acc[e] * f);               // the 16-bit value the unfused path stores
            quant_levels_i8_packed<16>(v16, qs, qinv, qrcp, w4);
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                if (dt * 32 + 8 * g >= HD) break;                     // HD = 80: the last tile holds d 64 .. 79 only
                const long col = (long)head * AT_D + dt * 32 + 8 * g + 4 * ko;
                *reinterpret_cast<unsigned *>(p.qout + act_offset(q_row, col, p.q_kpad, p.q_ld)) = w4[g];
            }
        } else {
            unsigned short *o = reinterpret_cast<unsigned short *>(p.out) + q_row * p.ldo + (long)head * AT_D + dt * 32;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                if (dt * 32 + 8 * g >= HD) break;
                v4us h;
#pragma unroll
                for (int e = 0; e < 4; ++e) h[e] = Elem<DT>::st(acc[4 * g + e] * f);
                *reinterpret_cast<v4us *>(o + 8 * g + 4 * ko) = h;
            }
        }
    }
    };
    if (nwe == NW) finish(std::integral_constant<int, NW>());
    else finish(std::integral_constant<int, NW / 2>());
}

}  // namespace mq

static thread_local int g_attn_waves = 0;      // TEST-ONLY (mq_attn_debug_waves): 0 = by shape, 2 / 4 = forced, 5 = 4 waves without the tile pairing

template <int HD, bool KV8>
static void attn_launch_t(mq::AttnArgs a, int dtype, hipStream_t st)
{
    using namespace mq;
    const long n = (a.T + AT_KB - 1) / AT_KB;
    // 4-wave workgroups: two per CU (LDS, registers).  When their grid needs more than one round on those 512 slots, 2-wave
    // workgroups (four per CU) keep more of a prefill resident.  Round 6 (after the grid order and the buffer loads), 4 -> 2 waves:
    // 7B shape (672 workgroups) 18.4 -> 17.6 us on 16-bit K / V and 18.5 -> 16.0 on the e4m3 cache, 72B (1536) 31.9 -> 29.8 and
    // 31.3 -> 26.8, 4096 tokens 230 -> 217 and 200 -> 192; worse when the 4-wave grid fits one round (512 tokens: 11.4 -> 12.7;
    // vision tower, 512 workgroups: 18.7 -> 21.7 us) -- tools/debug/attn_waves.py.
    // Causal launches whose 4-wave grid misses one round but fits it with the shallow half of the tiles PAIRED (see the kernel:
    // 0.75 n rows) take that form: 7B shape 672 -> 504 workgroups, 18.2 (unpaired) / 17.3 (2 waves) -> 16.2 us, e4m3 18.6 / 16.1 ->
    // 15.6; a grid that fits anyway is better left alone (512 tokens: 11.6 -> 13.5 paired), a larger one goes to 2 waves (72B: 28.8
    // against 32.3 paired).
    const long rows_paired = n / 2 + (n - n / 2 + 1) / 2;
    const bool fits = a.heads * n <= 512, fits_paired = a.causal && n >= 2 && a.heads * rows_paired <= 512;
    int nw = (HD == 128 && !fits && !fits_paired) ? 2 : 4;
    bool pair = !fits && fits_paired;
    if (g_attn_waves == 2 || g_attn_waves == 4 || g_attn_waves == 5) {
        nw = g_attn_waves == 2 ? 2 : 4;
        pair = g_attn_waves == 4 && a.causal && n >= 2;
    }
    a.deep_rows = (int)(pair ? n / 2 : n);
    const dim3 grid((unsigned)a.heads, (unsigned)(pair ? rows_paired : n));
    if (nw == 2) {
        if (dtype == MQ_F16) hipLaunchKernelGGL((attn_prefill_kernel<MQ_F16, KV8, HD, 2>), grid, dim3(128), 0, st, a);
        else hipLaunchKernelGGL((attn_prefill_kernel<MQ_BF16, KV8, HD, 2>), grid, dim3(128), 0, st, a);
    } else {
        if (dtype == MQ_F16) hipLaunchKernelGGL((attn_prefill_kernel<MQ_F16, KV8, HD, 4>), grid, dim3(256), 0, st, a);
        else hipLaunchKernelGGL((attn_prefill_kernel<MQ_BF16, KV8, HD, 4>), grid, dim3(256), 0, st, a);
    }
}

extern "C" int mq_attn_debug_waves(int waves)
{
    g_attn_waves = waves;
    return MQ_OK;
}

static int attn_launch(const mq::AttnArgs &a, int dtype, bool kv8, int head_dim, void *stream)
{
    // K / V are addressed through 32-bit buffer offsets (one block past the last row included)
    MQ_REQUIRE((a.T + mq::AT_KB) * a.ldkv < (1L << 32), "attention: (T + 32) x the K / V row stride in bytes must stay below 4 GiB (T %ld, stride %ld bytes)", a.T, a.ldkv);
    MQ_REQUIRE(a.T <= 65535L * mq::AT_KB, "attention: T %ld exceeds the grid (65535 query tiles of 32 rows)", a.T);
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (kv8) attn_launch_t<128, true>(a, dtype, st);
    else if (head_dim == 128) attn_launch_t<128, false>(a, dtype, st);
    else attn_launch_t<80, false>(a, dtype, st);
    return mq::check_launch("attn_prefill");
}

extern "C" int mq_attn_prefill_fp8kv(const void *q, int dtype, long T, int heads, int kv_heads, int head_dim, long ldq,
                                     const uint8_t *kv_cache, long ldkv, const float *kv_scale, float softmax_scale,
                                     int causal, void *out, long ldo, void *stream)
{
    using namespace mq;
    constexpr int D = 128;
    MQ_REQUIRE(dtype == MQ_F16 || dtype == MQ_BF16, "mq_attn_prefill_fp8kv: q / out dtype must be fp16 or bf16 (got %d)", dtype);
    MQ_REQUIRE(T >= 0 && heads >= 1 && kv_heads >= 1 && heads % kv_heads == 0, "mq_attn_prefill_fp8kv: bad head counts %d / %d", heads, kv_heads);
    MQ_REQUIRE(head_dim == D, "mq_attn_prefill_fp8kv: head_dim %d (the e4m3 variant is built for 128)", head_dim);
    // the running maximum is taken over RAW scores and the (positive) scale applied afterwards: a zero / negative / non-finite scale would
    // silently produce wrong probabilities (kv_scale entries must be positive and finite as well: kv_heads K scales, then kv_heads V scales)
    MQ_REQUIRE(softmax_scale > 0.0f && softmax_scale < 3.0e38f, "mq_attn_prefill_fp8kv: softmax_scale must be positive and finite (got %g)", (double)softmax_scale);
    if (T == 0) return MQ_OK;
    MQ_REQUIRE(q && kv_cache && kv_scale && out, "mq_attn_prefill_fp8kv: null pointer");
    MQ_REQUIRE(ldq >= (long)heads * D && ldo >= (long)heads * D && ldkv >= 2L * kv_heads * D, "mq_attn_prefill_fp8kv: row strides too short");
    MQ_REQUIRE(((uintptr_t)q) % 16 == 0 && (ldq * 2) % 16 == 0 && ((uintptr_t)kv_cache) % 16 == 0 && ldkv % 16 == 0 &&
                   ((uintptr_t)out) % 8 == 0 && (ldo * 2) % 8 == 0,
               "mq_attn_prefill_fp8kv: q / cache rows must be 16-byte aligned, out rows 8-byte aligned");
    AttnArgs a{q, kv_cache, kv_cache + (long)kv_heads * D, kv_scale, out, T, ldq, ldkv, ldo, heads, kv_heads, causal ? 1 : 0, softmax_scale,
               nullptr, 0, 0, 1.0f, 1.0f, nullptr};
    return attn_launch(a, dtype, true, head_dim, stream);
}

extern "C" int mq_attn_prefill(const void *q, int dtype, long T, int heads, int kv_heads, int head_dim, long ldq,
                               const void *k, const void *v, long ldkv, float softmax_scale, int causal, void *out, long ldo,
                               void *stream)
{
    using namespace mq;
    MQ_REQUIRE(dtype == MQ_F16 || dtype == MQ_BF16, "mq_attn_prefill: dtype must be fp16 or bf16 (got %d)", dtype);
    MQ_REQUIRE(T >= 0 && heads >= 1 && kv_heads >= 1 && heads % kv_heads == 0, "mq_attn_prefill: bad head counts %d / %d", heads, kv_heads);
    MQ_REQUIRE(head_dim == 128 || head_dim == 80, "mq_attn_prefill: head_dim %d (built: 128 and 80)", head_dim);
    MQ_REQUIRE(softmax_scale > 0.0f && softmax_scale < 3.0e38f, "mq_attn_prefill: softmax_scale must be positive and finite (got %g)", (double)softmax_scale);
    if (T == 0) return MQ_OK;
    MQ_REQUIRE(q && k && v && out, "mq_attn_prefill: null pointer");
    const long D = head_dim;
    MQ_REQUIRE(ldq >= (long)heads * D && ldo >= (long)heads * D && ldkv >= (long)kv_heads * D, "mq_attn_prefill: row strides too short");
    MQ_REQUIRE(((uintptr_t)q) % 16 == 0 && (ldq * 2) % 16 == 0 && ((uintptr_t)k) % 16 == 0 && ((uintptr_t)v) % 16 == 0 && (ldkv * 2) % 16 == 0 &&
                   ((uintptr_t)out) % 8 == 0 && (ldo * 2) % 8 == 0,
               "mq_attn_prefill: q / k / v rows must be 16-byte aligned, out rows 8-byte aligned");
    AttnArgs a{q, reinterpret_cast<const uint8_t *>(k), reinterpret_cast<const uint8_t *>(v), nullptr, out, T, ldq, ldkv * 2, ldo,
               heads, kv_heads, causal ? 1 : 0, softmax_scale, nullptr, 0, 0, 1.0f, 1.0f, nullptr};
    return attn_launch(a, dtype, false, head_dim, stream);
}

extern "C" int mq_attn_prefill_quant_i8(const void *q, int dtype, long T, int heads, int kv_heads, int head_dim, long ldq,
                                        const void *k, const void *v, long ldkv, const uint8_t *kv_cache, long ld_cache,
                                        const float *kv_scale, float softmax_scale, int causal, float scale0, float scale1,
                                        const uint8_t *row_sel, int8_t *out, long K_pad, long ldo, void *stream)
{
    using namespace mq;
    MQ_REQUIRE(dtype == MQ_F16 || dtype == MQ_BF16, "mq_attn_prefill_quant_i8: dtype must be fp16 or bf16 (got %d)", dtype);
    MQ_REQUIRE(T >= 0 && heads >= 1 && kv_heads >= 1 && heads % kv_heads == 0, "mq_attn_prefill_quant_i8: bad head counts %d / %d", heads, kv_heads);
    const bool kv8 = kv_cache != nullptr;
    MQ_REQUIRE(head_dim == 128 || (head_dim == 80 && !kv8), "mq_attn_prefill_quant_i8: head_dim %d (built: 128, and 80 for 16-bit K / V)", head_dim);
    MQ_REQUIRE(softmax_scale > 0.0f && softmax_scale < 3.0e38f, "mq_attn_prefill_quant_i8: softmax_scale must be positive and finite (got %g)", (double)softmax_scale);
    if (T == 0) return MQ_OK;
    const long D = head_dim;
    MQ_REQUIRE(q && out && (kv8 ? kv_scale != nullptr : (k && v)), "mq_attn_prefill_quant_i8: null pointer");
    MQ_REQUIRE(ldq >= (long)heads * D && ((uintptr_t)q) % 16 == 0 && (ldq * 2) % 16 == 0, "mq_attn_prefill_quant_i8: bad q geometry");
    MQ_REQUIRE(K_pad == (long)heads * D && K_pad % 64 == 0 && ((uintptr_t)out) % 16 == 0 && (ldo == MQ_LD_TILED || (ldo >= K_pad && ldo % 4 == 0)),
               "mq_attn_prefill_quant_i8: out must hold exactly heads * head_dim = %ld columns (K_pad %ld, a multiple of 64), ldo = MQ_LD_TILED or a row stride", (long)heads * D, K_pad);
    MQ_REQUIRE(scale0 > 0.0f && scale1 > 0.0f, "mq_attn_prefill_quant_i8: scales must be positive");
    AttnArgs a{};
    a.q = q; a.out = nullptr; a.T = T; a.ldq = ldq; a.ldo = 0; a.heads = heads; a.kv_heads = kv_heads; a.causal = causal ? 1 : 0;
    a.softmax_scale = softmax_scale; a.qout = out; a.q_kpad = K_pad; a.q_ld = ldo; a.qs0 = scale0; a.qs1 = scale1; a.row_sel = row_sel;
    a.qi0 = 1.0f / scale0; a.qi1 = 1.0f / scale1; a.qr0 = quant_rcp_ok(scale0) ? 1 : 0; a.qr1 = quant_rcp_ok(scale1) ? 1 : 0;
    if (kv8) {
        MQ_REQUIRE(ld_cache >= 2L * kv_heads * D && ((uintptr_t)kv_cache) % 16 == 0 && ld_cache % 16 == 0, "mq_attn_prefill_quant_i8: bad cache geometry");
        a.k = kv_cache; a.v = kv_cache + (long)kv_heads * D; a.kv_scale = kv_scale; a.ldkv = ld_cache;
    } else {
        MQ_REQUIRE(ldkv >= (long)kv_heads * D && ((uintptr_t)k) % 16 == 0 && ((uintptr_t)v) % 16 == 0 && (ldkv * 2) % 16 == 0, "mq_attn_prefill_quant_i8: bad k / v geometry");
        a.k = reinterpret_cast<const uint8_t *>(k); a.v = reinterpret_cast<const uint8_t *>(v); a.kv_scale = nullptr; a.ldkv = ldkv * 2;
    }
    return attn_launch(a, dtype, kv8, head_dim, stream);
}
