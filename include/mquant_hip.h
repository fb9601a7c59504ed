/*
 * mquant_hip.h -- C ABI of the MI355X (gfx950) W4A8 static-quant hot path.
 *
 * This is the drop-in boundary.  The reference (StiphyJay/MQuant) has no native
 * code of its own: its hot path is Python calling torch ops plus ONE third-party
 * CUDA extension (fast_hadamard_transform).  Each entry point below replaces the
 * chain of reference calls cited next to it (paths relative to the reference
 * tree); INTEGRATION.md shows the ctypes binding a maintainer adds on the
 * reference side.
 *
 * Conventions
 *   - extern "C", plain pointers and sizes, no torch types.
 *   - every pointer is a DEVICE pointer unless its name ends in _host.
 *   - stream is a hipStream_t passed as void* (NULL = default stream); all work is
 *     stream-ordered, nothing synchronises, nothing allocates.  Re-entrant: the entry points
 *     keep no mutable process state except (i) a per-device cache of kernel attributes behind
 *     a mutex and (ii) the TEST-ONLY overrides mq_gemm_debug_force / mq_hadamard_debug_threads, which are
 *     thread-local.  Nothing process-wide changes numerics: the non-default fast Hadamard stage is a flag of
 *     the call (MQ_HAD_FAST).
 *   - return value: 0 on success; >0 a hipError_t; <0 an argument error
 *     (MQ_EINVAL...).  mq_last_error() returns a thread-local message.
 *   - dtype codes: MQ_F16 / MQ_BF16 / MQ_F32 for activations and outputs.
 *   - bit-exactness contract: integer outputs (int8 levels, int32 accumulators,
 *     packed nibbles) equal oracle/mq_oracle.c bit for bit; fp outputs are
 *     computed with the same single-rounded fp32 operations as the oracle.
 */
#ifndef MQUANT_HIP_H
#define MQUANT_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MQ_F16 0
#define MQ_BF16 1
#define MQ_F32 2
#define MQ_F64 3   /* mq_rotate_f64 only */

#define MQ_OK 0
#define MQ_EINVAL (-1)      /* bad argument (shape / alignment / dtype)          */
#define MQ_EUNSUPPORTED (-2) /* valid request this build does not implement       */

/* Layout of the int8 activation matrix handed from the quantizers to the GEMM.  Every entry point
 * that takes (int8 matrix, K_pad, leading dimension) accepts either
 *   ld >= K_pad      row-major [M, K_pad], ld bytes per row; or
 *   ld == MQ_LD_TILED  the TILED layout: [ceil(M/16)][K_pad/64] pieces of 1 KiB; inside a piece the
 *                    16-byte chunk holding k = 64 kt + 16 c .. +15 of row 16 mt + r sits at byte
 *                    (16 c + r) * 16, i.e. byte offset of element (m, k) =
 *                      ((m/16) * (K_pad/64) + k/64) * 1024 + (((k/16)%4) * 16 + m%16) * 16 + k%16.
 *                    K_pad % 64 == 0; the buffer holds ceil16(M) * K_pad bytes (rows >= M are never
 *                    read into a result).  One piece is one MFMA operand fragment in lane order, so
 *                    the GEMM fetches it with a single contiguous 1 KiB LDS-DMA: a CU gathers rows
 *                    of a row-major matrix at ~14.5 B/clk but streams contiguous KiB at 40-50 B/clk
 *                    (DESIGN.md 4.1). */
#define MQ_LD_TILED 0L

/* Geometry of the pre-tiled weight image consumed by mq_gemm_w4a8. */
#define MQ_W_TILE_N 16   /* output channels per MFMA fragment                     */
#define MQ_W_TILE_K 128  /* reduction elements per fragment pair (2 x K=64 MFMAs) */

int mq_version(void);
const char *mq_last_error(void);

/* Name and compute-unit count of the current device (host strings/ints). */
int mq_device_info(char *name_host, size_t name_len, int *cu_count_host);

/* ---------------------------------------------------------------------------
 * Static activation quantizer: fp -> int8 levels.
 * Replaces UniformQuantizer.quant on the static path
 *   fake_quant/quantizer/uniform.py:20-33, fake_quant/quantizer/base.py:44-50
 *   (called from ActQuantWrapper.forward, fake_quant/quant_utils.py:378-383).
 *   q = clamp(rint(float(x) / s), -128, 127), IEEE division, round-half-even.
 * x: [M, K] row-major with leading dimension ldx (elements).
 * out: [M, K_pad] int8 row-major, leading dimension ldo (bytes) >= K_pad; columns
 *      K..K_pad-1 are written as 0 so the GEMM can consume whole 128-wide tiles.
 * Scale selection:
 *   scale_vec0 == NULL: per-tensor scales passed by value (scale0 / scale1);
 *   scale_vec0 != NULL: per-channel scales [K] (calibration_mode="channel_wise").
 *   row_sel == NULL: every row uses set 0; else row_sel[m] in {0,1} picks the set
 *   (Modality-Specific Static Quantization: vision vs. text rows).
 * skip_col0 != 0 implements ActQuantWrapper.split (quant_utils.py:367-376): channel 0
 *   bypasses the quantizer: out[:,0] = 0 and x0_out[m] = float(x[m,0]).
 * ------------------------------------------------------------------------- */
int mq_quantize_act_i8(const void *x, int x_dtype, long M, long K, long ldx,
                       float scale0, float scale1,
                       const float *scale_vec0, const float *scale_vec1,
                       const uint8_t *row_sel, int skip_col0, float *x0_out,
                       int8_t *out, long K_pad, long ldo, void *stream);

/* Fused quantize->dequantize in x's dtype (the reference's simulated form,
 * uniform.py:20-43 + base.py:44-50), used when weights stay in floating point. */
int mq_fakequant_act(const void *x, int x_dtype, long M, long K, long ldx,
                     float scale0, float scale1,
                     const float *scale_vec0, const float *scale_vec1,
                     const uint8_t *row_sel, int skip_col0,
                     void *out, long ldo, void *stream);

/* ---------------------------------------------------------------------------
 * Online Hadamard rotation, optionally fused with the static quantizer.
 * Replaces revise_down_input + matmul_hadU_cuda (+ UniformQuantizer.quant):
 *   fake_quant/utils.py:465-471, fake_quant/hadamard_utils.py:115-128,
 *   fake_quant/quant_utils.py:334-341 (and :378-383 when quantizing).
 *   y = (H_K (x) H_{n/K}) [x ; 0] / sqrt(n),  flat index = k*(n/K) + j.
 * x: [M, n_in] (n_in <= n: zero padded up to n, the forward-pre-hook's job).
 * had_words: the signs of hadK as word-aligned rows: uint32 [K][ceil(K/32)], bit b of
 *           word w of row j is 1 when hadK[j][32*w + b] == +1 (unused high bits 0);
 *           NULL when K == 1.  (hadK itself is data: fake_quant/hadamard_utils.py:201-.)
 * fp32_had: 0 = the extension's behaviour for half inputs (butterflies in fp32,
 *           result rounded to x's dtype before the K x K stage and again after),
 *           1 = --fp32_had (fp32 throughout, one final cast to x's dtype).
 * The K x K stage accumulates in ascending k as a single fp32 chain (oracle step 4).
 *
 * mq_hadamard:        writes the rotated activations in x's dtype, [M, n].
 * mq_hadamard_quant_i8: writes int8 levels like mq_quantize_act_i8 (same scale /
 *           row_sel / skip_col0 / x0_out semantics); the rotated activations never
 *           reach HBM.
 * ------------------------------------------------------------------------- */
/* fp32_had is a flag word: MQ_HAD_FP32 = --fp32_had; MQ_HAD_PREPARED = had_words points at a
 * descriptor written by mq_hadamard_prepare (16-byte aligned, mq_hadamard_prepared_bytes(K) bytes): the sign
 * words, followed by the 64-lane masks of the fp32 MFMA sign operand (one VALU instruction per operand instead
 * of three -- the fp32 MFMA shares the vector ALU's datapath, every VALU instruction beside it is lost matrix
 * time) and by the +-1 half-precision operand images of the fast stage (MQ_HAD_FAST).  Results are
 * identical with and without the descriptor. */
#define MQ_HAD_FP32 1
#define MQ_HAD_PREPARED 2
/* NON-DEFAULT accuracy / speed flag of ONE call (there is no process-wide mode): without it the K x K stage is the sequential
 * fp32 add chain of the reference's CPU run, bit-identical to the goldens.  With MQ_HAD_FAST, half-precision activations
 * (x_dtype MQ_F16 / MQ_BF16, MQ_HAD_FP32 off, 64 <= n/K <= 512) run that stage on the fp16 / bf16 matrix core
 * (V_MFMA_F32_32X32X16): the products +-1 * y are exact, only the ORDER of the fp32 accumulation differs -- what the
 * reference's own GPU run does with a half-precision cuBLAS GEMM (hadamard_utils.py:127).  Not bit-identical to the exact
 * kernel: DESIGN.md 4.2 gives the measured int8 level flip rate and the verdict against the reference goldens
 * (tests/test_gpu_hadamard_fast.py).  Shapes / dtypes outside the fast kernel take the exact one. */
#define MQ_HAD_FAST 4
size_t mq_hadamard_prepared_bytes(int K);
int mq_hadamard_prepare(const uint32_t *had_words, int K, void *descriptor, void *stream);

int mq_hadamard(const void *x, int x_dtype, long M, long n_in, long ldx,
                long n, int K, const uint32_t *had_words, int fp32_had,
                void *out, long ldo, void *stream);

int mq_hadamard_quant_i8(const void *x, int x_dtype, long M, long n_in, long ldx,
                         long n, int K, const uint32_t *had_words, int fp32_had,
                         float scale0, float scale1, const uint8_t *row_sel,
                         int skip_col0, float *x0_out,
                         int8_t *out, long K_pad, long ldo, void *stream);

/* Same with the activation that produces the rotated Linear's input fused in front (SURVEY 8(f3):
 * "SiLU.up fused into the down_proj Hadamard-quant prologue"): the row fed to the transform is
 *   act == 1 (MQ_ACT_SILU_MUL):   cast(cast(x / (1 + exp(-x))) * x2)   x = gate, x2 = up (same ldx)
 *   act == 2 (MQ_ACT_QUICK_GELU): cast(x * cast(1 / (1 + exp(-cast(1.702 x)))))
 * i.e. torch's F.silu(gate) * up / QuickGELUActivation on tensors of x_dtype (fp32 inside an op,
 * one rounding per op, the device library's expf): bit-identical to running those torch ops on
 * this GPU and then mq_hadamard_quant_i8 (checked at full size in fp16, bf16 and fp32).  These
 * activations are HF model code, not MQuant's; the C oracle restates them with a correctly rounded
 * exp and agrees up to that last bit.  Requires n/K >= 8. */
int mq_act_hadamard_quant_i8(const void *x, const void *x2, int act, int x_dtype, long M,
                             long n_in, long ldx, long n, int K, const uint32_t *had_words,
                             int fp32_had, float scale0, float scale1, const uint8_t *row_sel,
                             int skip_col0, float *x0_out, int8_t *out, long K_pad, long ldo,
                             void *stream);

/* TEST-ONLY hook (thread-local, not part of the drop-in surface): 256 or 512 threads per row. */
int mq_hadamard_debug_threads(int threads);

/* ---------------------------------------------------------------------------
 * Weight formats.
 * mq_pack_i4 / mq_unpack_i4: the reference wire format, quant_utils.py:61-94
 *   (two's-complement nibbles, even index -> low nibble, along the last dim).
 * mq_weight_levels: integer levels of a fake-quantized weight,
 *   q[n][k] = clamp(rint(float(w[n][k]) / scale[n]), lo, hi)
 *   (the inverse of WeightQuantizer.quantize, quant_utils.py:42-45,512-518).
 * mq_prepack_w4 / mq_prepack_w8: re-tile int levels [N, K] into the image the GEMM
 *   streams straight into LDS (layout in DESIGN.md).  N_pad = ceil16(N),
 *   K_pad = ceil128(K); padding is zero.  Output bytes: N_pad*K_pad/2 (w4),
 *   N_pad*K_pad (w8).  zero_col0 clears column 0 (ActQuantWrapper.split: L2 has no
 *   column for the bypassed channel, quant_utils.py:320-326).
 * ------------------------------------------------------------------------- */
int mq_pack_i4(const int8_t *q, long rows, long cols, uint8_t *out, void *stream);
int mq_unpack_i4(const uint8_t *packed, long rows, long cols, int8_t *out, void *stream);
int mq_weight_levels(const void *w, int w_dtype, long N, long K, long ldw,
                     const float *scale, int lo, int hi, int8_t *q, void *stream);
int mq_prepack_w4(const int8_t *q, long N, long K, int zero_col0, uint8_t *out, void *stream);
int mq_prepack_w8(const int8_t *q, long N, long K, int zero_col0, int8_t *out, void *stream);
size_t mq_prepacked_bytes(long N, long K, int w_bits);

/* ---------------------------------------------------------------------------
 * Dynamic per-token activation quantizer: the reference's DEFAULT mode when --*_static is not
 * given (ActQuantizer.find_params + forward, quant_utils.py:116-133,205-268; symmetric,
 * act_per_tensor = False, groupsize = -1).  Per row m, fp32 arithmetic on the converted values:
 *   xmin = min(min_k x, 0)*clip_ratio; xmax = max(max_k x, 0)*clip_ratio; maxq = 2^(bits-1)-1
 *   scale_out[m] = max(|xmin|, xmax) / maxq   (1 where that is 0)
 *   out[m][k]    = clamp(rint(x / scale_out[m]), -(maxq+1), maxq), zero for K <= k < K_pad
 * skip_col0: column 0 is left out of the range, returned in x0_out and given level 0
 * (ActQuantWrapper.split, quant_utils.py:367-372).  Pair with mq_gemm_w4a8_rowscale_ws.
 * ------------------------------------------------------------------------- */
int mq_quantize_act_dyn_i8(const void *x, int x_dtype, long M, long K, long ldx, int bits,
                           float clip_ratio, int skip_col0, float *x0_out, float *scale_out,
                           int8_t *out, long K_pad, long ldo, void *stream);

/* The ASYMMETRIC variant (--a_asym without --*_static; quant_utils.py:239-268 else-branch + asym_quant
 * :27-31), maxq = 2^bits - 1, per row m:
 *   xmin / xmax as above (both 0 -> -1 / +1); scale = (xmax - xmin) / maxq; zero = rint(-xmin / scale)
 *   q = clamp(rint(x / scale) + zero, 0, maxq);  out[m][k] = q - 2^(bits-1)  (int8), zero for k >= K
 *   shift_out[m] = scale * (2^(bits-1) - zero):  x_hat[m][k] = scale * out[m][k] + shift_out[m]
 * Pair with mq_gemm_w4a8_rowscale_ws(s_x_rows = scale_out, x0 = shift_out, w0[n] = s_w[n] * sum_k q_w[n][k]):
 * the rank-1 epilogue term restores offset and zero point.  zero_out may be NULL. */
int mq_quantize_act_dyn_asym_i8(const void *x, int x_dtype, long M, long K, long ldx, int bits,
                                float clip_ratio, float *scale_out, float *zero_out, float *shift_out,
                                int8_t *out, long K_pad, long ldo, void *stream);

/* The PER-TENSOR variant (act_per_tensor = True; quant_utils.py:214-237): one range for the whole
 * tensor, taken from `minmax` = the two device floats written by mq_minmax_tensor (col_begin = 1 under
 * skip_col0) -- no host round trip.  asym = 0: scale = max(|xmin|, xmax) / (2^(bits-1) - 1) (1 if 0);
 * asym = 1: a zero bound becomes -1 resp. +1 on its own, scale / zero / stored levels / shift as in
 * mq_quantize_act_dyn_asym_i8 (no skip_col0 then).  scale_out (and zero_out / shift_out, may be NULL when
 * asym = 0) are written per row for mq_gemm_w4a8_rowscale_ws.  For fp16 / bf16 inputs every step is rounded to
 * x's dtype like the reference's (range * clip, scale, zero point, x / scale, level sum are 16-bit tensors there:
 * `torch.tensor(0).to(x)`, the int64 maxq tensor does not promote); the per-token entry points promote to fp32 (:239). */
int mq_quantize_act_range_i8(const void *x, int x_dtype, long M, long K, long ldx, int bits,
                             float clip_ratio, int asym, int skip_col0, const float *minmax,
                             float *x0_out, float *scale_out, float *zero_out, float *shift_out,
                             int8_t *out, long K_pad, long ldo, void *stream);

/* ---------------------------------------------------------------------------
 * RMSN + static quantizer in one pass.  Replaces module_util.RMSN.forward
 * (module_util.py:55-61) followed by UniformQuantizer.quant (uniform.py:20-33):
 *   h = fp32(x);  y = cast_to_x_dtype(h * (1/sqrt(sum(h*h)/mean_dim + eps)));
 *   out[m][k] = clamp(rint(y / scale_sel(m)), -128, 127), zero for K <= k < K_pad
 * x: [M, ldx], rows 16-byte aligned, K % 16 == 0 (MQ_BF16 rows are normalised in bf16 arithmetic, one
 * rounding per torch op, because upstream promotes fp16 only),
 * K <= 16384.  y_out (optional, [M, ldy], x's dtype) receives the normalised activations.
 * The sum of squares uses the fixed order documented in csrc/rmsn_quant.hip (oracle: orc_rmsn);
 * torch's own reduction order differs in the last fp32 bit, i.e. <= 1 ulp of y.
 * ------------------------------------------------------------------------- */
int mq_rmsn_quantize_i8(const void *x, int x_dtype, long M, long K, long ldx, float mean_dim,
                        float eps, float scale0, float scale1, const uint8_t *row_sel,
                        void *y_out, long ldy, int8_t *out, long K_pad, long ldo, void *stream);

/* ---------------------------------------------------------------------------
 * Weight quantizer on the device (offline weight pipeline).
 * Replaces WeightQuantizer.find_params + WeightQuantizer.quantize for the symmetric
 * per-output-channel case, quant_utils.py:446-518 (sym_quant_dequant :46-58):
 *   xmax[n]  = max(|min(min_k w, 0)|, max(max_k w, 0)) clamped to >= 1e-5
 *   scale[n] = xmax / maxq,  maxq = 2^(bits-1) - 1
 *   mse != 0: clip search over p = 1 - i/grid, i < (int)(maxshrink*grid):
 *             err_i = sum_k |s_i*q - w|^norm (fp32, ascending k), first strict minimum wins
 *   q[n][k]  = clamp(rint(w/scale[n]), -(maxq+1), maxq)
 * All arithmetic is fp32 on the exactly converted weights for EVERY w_dtype (the reference
 * promotes half tensors through the fp32 `tmp` of :458-460).  Optional outputs (NULL = skip):
 *   levels  int8 [N, K]
 *   packed  uint8 [N, K/2], the reference pack_i4 wire format (:61-69); bits == 4, K even
 *   wq      fake-quantized weights scale*q cast to w_dtype, [N, ldq] (what quantize() returns)
 * ------------------------------------------------------------------------- */
int mq_wquant_sym(const void *w, int w_dtype, long N, long K, long ldw, int bits, int mse,
                  float norm, int grid, float maxshrink, float *scale, int8_t *levels,
                  uint8_t *packed, void *wq, long ldq, void *stream);

/* ---------------------------------------------------------------------------
 * Rotary embedding (rotate-half), in place on `heads` consecutive head_dim-wide column blocks of
 * x [T, ldx] -- e.g. the q and k parts of a fused q|k|v GEMM output.  Not an MQuant function (the
 * reference never touches RoPE): glue of the whole-prefill TTFT harness, bit-identical to the HF
 * formula evaluated with torch ops on x_dtype tensors:
 *   out = cast(cast(x*cos) + cast(rotate_half(x)*sin)),  cos/sin: [T, head_dim] in x_dtype.
 * ------------------------------------------------------------------------- */
int mq_rope_inplace(void *x, int x_dtype, long T, int heads, int head_dim, long ldx,
                    const void *cos, const void *sin, void *stream);

/* ---------------------------------------------------------------------------
 * Offline rotation of weight rows in fp64.  Replaces the dense
 *   W <- (W.double() @ Q).to(W.dtype),  Q = random_hadamard_matrix(n) = diag(s) H_n / sqrt(n)
 * of fake_quant/rotation_utils.py (rotate_* helpers) as driven by qwen2vl_rotation.py:232-332 and
 * internvl_rotation.py:223-303; Q is built at hadamard_utils.py:107-112.  In place, per row:
 *   x <- cast( (H_K (x) H_{n/K}) (s . x) / (double)sqrtf(n) ),   flat index = k*(n/K) + j,
 * evaluated in fp64 (sign flip, butterflies, K x K sign stage, one division), cast to x's dtype
 * through fp32 like torch's double -> half conversion.  signs: n doubles (+-1), NULL = no flip
 * (plain H_n, e.g. the per-head Hadamard).  had_words as for mq_hadamard (plain words, NULL when
 * K == 1).  Q^T W is the same call on the rows of W^T.  n * 8 bytes must fit the 160 KiB LDS.
 * ------------------------------------------------------------------------- */
int mq_rotate_f64(void *x, int dtype, long M, long n, long ld, const double *signs,
                  int K, const uint32_t *had_words, void *stream);

/* ---------------------------------------------------------------------------
 * fp8 KV cache (BASELINE configuration 5).  No reference implementation exists
 * (fake_quant/utils.py:220-267 are flags of a parser nobody calls): parity is UNPINNED, the checker
 * is oracle/mq_oracle.c (orc_kv_quant_fp8 / orc_kv_dequant_fp8; its e4m3fn codec is pinned to
 * torch.float8_e4m3fn on the CPU).  OCP e4m3fn (not fnuz), one static scale per KV head:
 *   write:  q[t][h][d] = e4m3fn_rne(clamp(kv[t][h][d] / scale[h], -448, 448))
 *   read:   out[t][h][d] = cast(float(q[t][h][d]) * scale[h])
 * kv / out: [T][ld] elements with the heads * head_dim values of a token contiguous (ld lets the
 * K or V slice of a fused qkv GEMM output be read in place); q: [T][ldo] bytes.  head_dim % 8 == 0,
 * rows 16-byte aligned (8 for the fp8 side).
 * ------------------------------------------------------------------------- */
int mq_kv_quant_fp8(const void *kv, int dtype, long T, int heads, int head_dim, long ld,
                    const float *scale, uint8_t *out, long ldo, void *stream);
int mq_kv_dequant_fp8(const uint8_t *q, long T, int heads, int head_dim, long ld,
                      const float *scale, void *out, int out_dtype, long ldo, void *stream);
/* The write with the READ-BACK fused in: besides the e4m3 bytes it stores readback[t][h][d] = cast(float(q) * s[h])
 * (what mq_kv_dequant_fp8 would return later) in the source dtype, row stride ldr elements -- the K / V the
 * attention of the SAME prefill step consumes, so prefill and decode attend over identical values and the
 * separate dequantise-on-read launch is gone.  readback == NULL: mq_kv_quant_fp8. */
int mq_kv_quant_fp8_readback(const void *kv, int dtype, long T, int heads, int head_dim, long ld,
                             const float *scale, uint8_t *out, long ldo, void *readback, long ldr, void *stream);

/* Prefill attention straight off the e4m3 cache (no reference counterpart -- the reference leaves
 * attention to the HF model code and never quantizes a cache; parity unpinned, checker = softmax attention over the
 * dequantised cache).  K and V are read as ONE byte per element and widened inside the kernel; the per-head scales
 * fold into the score / output scale, so no dequantised copy of the cache ever exists in HBM:
 *     O[t][h][:] = softmax_k((Q[t][h] . K8[k][g]) * s[g] * softmax_scale) @ V8[k][g] * s[kv_heads + g],  g = h / (heads / kv_heads)
 * q / out: [T][ld] elements of `dtype` (MQ_F16 / MQ_BF16) with heads * head_dim values per token (ldq lets the Q
 * columns of a fused q|k|v GEMM output be read in place); kv_cache: [T][ldkv] bytes, per token the K heads then the
 * V heads (the layout mq_kv_quant_fp8 writes for the K|V columns); kv_scale: [2 * kv_heads] floats.  head_dim == 128;
 * q / cache rows 16-byte aligned, out rows 8-byte aligned.  causal != 0: key index <= query index. */
int mq_attn_prefill_fp8kv(const void *q, int dtype, long T, int heads, int kv_heads, int head_dim, long ldq,
                          const uint8_t *kv_cache, long ldkv, const float *kv_scale, float softmax_scale,
                          int causal, void *out, long ldo, void *stream);
/* The same kernel over UNQUANTISED K / V of q's dtype (the K and V column slices of the fused q|k|v GEMM output, read in
 * place): k / v point at the first K / V head of token 0, a token's kv_heads * head_dim values contiguous, row stride ldkv
 * ELEMENTS.  head_dim 128 or 80 (Qwen2-VL's vision tower; rows stay 16-byte aligned: 160 bytes per head).  Replaces
 * repeat_kv + scaled_dot_product_attention in the glue of the whole-prefill report. */
int mq_attn_prefill(const void *q, int dtype, long T, int heads, int kv_heads, int head_dim, long ldq,
                    const void *k, const void *v, long ldkv, float softmax_scale, int causal, void *out, long ldo,
                    void *stream);
/* TEST-ONLY, thread-local: force the waves per workgroup (= ways the keys are split) of the attention kernels: 2 or 4; 0 = by shape. */
int mq_attn_debug_waves(int waves);

/* Either attention with the NEXT Linear's static int8 activation quantizer fused into its store (the o_proj /
 * proj input): out[t][c] = clamp(rint(cast_dtype(o[t][c]) / s_t), -128, 127), s_t = scale1 where row_sel[t] != 0 else
 * scale0 -- the bytes mq_quantize_act_i8 writes for the 16-bit attention output, in the same activation layout (K_pad ==
 * heads * head_dim, ldo = MQ_LD_TILED or a row stride).  kv_cache != NULL: the e4m3 variant (k, v, ldkv ignored); else the
 * 16-bit variant (kv_cache, ld_cache, kv_scale ignored). */
int mq_attn_prefill_quant_i8(const void *q, int dtype, long T, int heads, int kv_heads, int head_dim, long ldq,
                             const void *k, const void *v, long ldkv, const uint8_t *kv_cache, long ld_cache,
                             const float *kv_scale, float softmax_scale, int causal, float scale0, float scale1,
                             const uint8_t *row_sel, int8_t *out, long K_pad, long ldo, void *stream);

/* out[m][n] = sum_k x[m][k] * W[n][k], 16-bit x ([M <= 8, K], ldx elements per row) and W ([N, K], ldw), fp32 products and sums,
 * one rounding to the same 16-bit dtype: the UNQUANTIZED lm_head on the last position(s) of a prefill (the reference leaves
 * lm_head in 16 bits: exam/quant_qwen2vl.py:130-143 wraps the decoder's and the vision tower's Linears only; HF computes
 * logits = lm_head(hidden[:, -1:])).  Glue of the whole-prefill report, replaces torch.matmul (hipBLASLt) there: W is
 * streamed once at the chip's HBM rate.  K a multiple of 8, rows 16-byte aligned, M * K * 2 bytes <= 64 KiB. */
int mq_gemv_f16(const void *x, int dtype, int M, long K, long ldx, const void *w, long N, long ldw, void *out, long ldo,
                void *stream);


/* ---------------------------------------------------------------------------
 * GPTQ: the column loop of one lazy-batch block, gptq/gptq_utils.py:258-279 (symmetric
 * per-channel quantizer, no groups).  For i = 0..cols-1, per output row n:
 *   q = scale[n]*clamp(rint(w_i/scale[n]), -2^(bits-1), 2^(bits-1)-1);  err = (w_i - q)/Hinv1[i][i];
 *   w_j -= err*Hinv1[i][j]  (j > i);   Q1[n][i] = q;  Err1[n][i] = err
 * fp32, the reference's operation order (bit-identical to the torch loop).  W1 is read only
 * ([N, ldw], the block's columns of the working weights), cols <= 128, Hinv1 points at
 * Hinv[i1][i1] of the upper Cholesky factor with leading dimension ldh.  The caller applies the
 * trailing update W[:, i2:] -= Err1 @ Hinv[i1:i2, i2:] (a plain GEMM) as upstream does.
 * ------------------------------------------------------------------------- */
int mq_gptq_block(const float *W1, long N, int cols, long ldw, const float *Hinv1, long ldh,
                  const float *scale, int bits, float *Q1, long ldq, float *Err1, long lde,
                  void *stream);

/* ---------------------------------------------------------------------------
 * Quantized Linear: int8 activations x int4/int8 weights -> int32 -> dequant.
 * Replaces F.linear on fake-quantized tensors, quant_utils.py:384 (and the fp32
 * L1/L2 pair of :374-376 when x0/w0 are given), with
 *   acc[m][n] = sum_k a[m][k] * w[n][k]                      (int32, exact)
 *   y[m][n]   = ((float(acc) * s_x[row_sel[m]]) * s_w[n]) + bias[n] + x0[m]*w0[n]
 * each fp32 operation rounded once, in that order, then cast to out_dtype.
 * a: [M, K_pad] int8, leading dimension lda (bytes, multiple of 16) or lda = MQ_LD_TILED (the
 *    layout the quantizers write on request, fastest), K_pad % 128 == 0.
 * w: image produced by mq_prepack_w4 / _w8 for (N, K) with the same K_pad.
 * bias, x0, w0, row_sel may be NULL.  out: [M, N], leading dimension ldo (elements).
 * mq_gemm_w4a8_i32 stores the raw accumulators (parity interface).
 * ------------------------------------------------------------------------- */
int mq_gemm_w4a8(const int8_t *a, long lda, const void *w, int w_bits,
                 long M, long N, long K_pad,
                 float s_x0, float s_x1, const uint8_t *row_sel,
                 const float *s_w, const float *bias,
                 const float *x0, const float *w0,
                 void *out, int out_dtype, long ldo, void *stream);

/* Same GEMM with the residual add of the surrounding block folded into the epilogue:
 *   out[m][n] = cast(cast(y[m][n]) + residual[m][n])
 * i.e. torch's `hidden + linear(x)` on tensors of out_dtype (the Linear's output is rounded to
 * out_dtype before the add).  residual: [M, ldr] in out_dtype; may alias out (each element is read
 * and written by the same lane). */
int mq_gemm_w4a8_residual_ws(const int8_t *a, long lda, const void *w, int w_bits, long M, long N,
                             long K_pad, float s_x0, float s_x1, const uint8_t *row_sel,
                             const float *s_w, const float *bias, const float *x0, const float *w0,
                             const void *residual, long ldr, void *out, int out_dtype, long ldo,
                             void *workspace, size_t workspace_bytes, void *stream);

/* Same GEMM with one activation scale PER ROW (s_x_rows[m], from mq_quantize_act_dyn_i8) instead of
 * the static scale set(s):  y[m][n] = ((float(acc) * s_x_rows[m]) * s_w[n]) + bias[n] + x0[m]*w0[n]. */
int mq_gemm_w4a8_rowscale_ws(const int8_t *a, long lda, const void *w, int w_bits, long M, long N,
                             long K_pad, const float *s_x_rows, const float *s_w, const float *bias,
                             const float *x0, const float *w0, void *out, int out_dtype, long ldo,
                             void *workspace, size_t workspace_bytes, void *stream);

/* Same GEMM with TWO rank-1 epilogue terms, for the flag combinations whose integer evaluation needs the slot twice
 * (reference quant_utils.py:205-268 asymmetric activations, :367-376 split column, :446-509 asymmetric weights):
 *   y[m][n] = ((float(acc) * s_x[m]) * s_w[n]) + bias[n] + x0[m]*w0[n] + x1[m]*w1[n]      (one rounding per operation, in that order)
 * s_x[m] = s_x_rows[m] when s_x_rows is given (dynamic quantizers), else the static scale set(s) with row_sel.
 *   --w_asym + --visual_split:  (x[m][0], L1 weight)                       and (s_x sum_k a[m][k], s_w (2^(b-1) - z_w))
 *   --a_asym + --visual_split:  (x[m][0], L1 weight)                       and (s_x (2^(b-1) - z_x)[m], s_w sum_k q_w[n][k])
 *   --w_asym + --a_asym:        (s_x (2^(b-1) - z_x)[m], s_w (sum_k q_w[n][k] + K (2^(b-1) - z_w)))  and (s_x sum_k a[m][k], s_w (2^(b-1) - z_w))
 * (stored levels are q - 2^(bits-1) on an asymmetric side; goldens tests/golden/wrapper_rank2_*.npz from the reference's forward). */
int mq_gemm_w4a8_rank2_ws(const int8_t *a, long lda, const void *w, int w_bits, long M, long N, long K_pad,
                          float s_x0, float s_x1, const uint8_t *row_sel, const float *s_x_rows,
                          const float *s_w, const float *bias, const float *x0, const float *w0,
                          const float *x1, const float *w1, void *out, int out_dtype, long ldo,
                          void *workspace, size_t workspace_bytes, void *stream);

int mq_gemm_w4a8_i32(const int8_t *a, long lda, const void *w, int w_bits,
                     long M, long N, long K_pad, int32_t *acc, long ldacc, void *stream);

/* Same two entry points with a caller-owned scratch buffer, which lets the library split
 * the reduction over workgroups when M x N alone cannot fill the 256 CUs (e.g. down_proj:
 * K = 19968, N = 3584).  Partial sums are int32 and are combined in a fixed order, so the
 * result is bit-identical to the unsplit call.  workspace: 16-byte aligned device memory,
 * workspace_bytes >= 8 * M * N * 4 allows every split factor the heuristic may pick;
 * smaller buffers merely restrict it.  workspace == NULL degrades to the calls above. */
int mq_gemm_w4a8_ws(const int8_t *a, long lda, const void *w, int w_bits,
                    long M, long N, long K_pad,
                    float s_x0, float s_x1, const uint8_t *row_sel,
                    const float *s_w, const float *bias,
                    const float *x0, const float *w0,
                    void *out, int out_dtype, long ldo,
                    void *workspace, size_t workspace_bytes, void *stream);

int mq_gemm_w4a8_i32_ws(const int8_t *a, long lda, const void *w, int w_bits,
                        long M, long N, long K_pad, int32_t *acc, long ldacc,
                        void *workspace, size_t workspace_bytes, void *stream);

/* Dynamic GROUP-WISE activation quantizer (--a_groupsize g; ActQuantizer.find_params_per_token_groupwise,
 * quant_utils.py:181-203, + sym_quant :46-50), symmetric: one scale per row and group of g consecutive channels,
 * scale = max(|amin * clip|, amax * clip) / maxq (1 for an all-zero group), no zero inclusion, every intermediate
 * rounded to x_dtype exactly like the reference's tensors of that dtype.  g: a power of two in 16..1024 dividing K;
 * K_pad must hold whole groups.  scales_out: fp32 [M, K / g].  out as for mq_quantize_act_i8 (row-major or tiled). */
int mq_quantize_act_group_i8(const void *x, int x_dtype, long M, long K, long ldx, int groupsize, int bits,
                             float clip_ratio, float *scales_out, int8_t *out, long K_pad, long ldo, void *stream);

/* The same with ASYMMETRIC levels (--a_groupsize g + --a_asym; the sym = False branch of quant_utils.py:181-203 + asym_quant
 * :27-31): per row and group  range = [amin, amax] * clip (0 need not be inside; (-1, +1) for an all-zero group),
 * scale = (xmax - xmin) / (2^bits - 1), zero = round(-xmin / scale), q = clamp(round(x / scale) + zero, 0, 2^bits - 1), every
 * intermediate rounded to x_dtype like the reference's tensors.  out holds q - 2^(bits-1); shift_out[m][g] = scale (2^(bits-1) -
 * zero) in fp32: x_hat = scale * out + shift.  zero_out may be NULL.  Pair with mq_gemm_w4a8_groupscale_asym. */
int mq_quantize_act_group_asym_i8(const void *x, int x_dtype, long M, long K, long ldx, int groupsize, int bits,
                                  float clip_ratio, float *scales_out, float *zero_out, float *shift_out, int8_t *out,
                                  long K_pad, long ldo, void *stream);

/* The GEMM for group-wise activation scales: y[m][n] = (sum_g (float(acc_g[m][n]) * s_x_groups[m][g])) * s_w[n] + bias[n],
 * acc_g = the exact int32 sum over the group's k, groups added in ascending order in fp32 (one rounding per group).
 * Replaces F.linear on the group-wise fake-quantized activations (quant_utils.py:384 after :181-203; the reference
 * accumulates s * q products in floating point in library order: outputs agree to the 1e-3 of the other modes).
 * group_k: 64 or a multiple of 128; n_groups * group_k covers K_pad up to its zero padding. */
int mq_gemm_w4a8_groupscale(const int8_t *a, long lda, const void *w, int w_bits, long M, long N, long K_pad,
                            const float *s_x_groups, long n_groups, int group_k, const float *s_w,
                            const float *bias, void *out, int out_dtype, long ldo, void *stream);

/* ... for asymmetric groups: y[m][n] = (sum_g (float(acc_g) * s_x_groups[m][g] + shift_groups[m][g] * wsum_groups[g][n])) * s_w[n] + bias[n],
 * wsum_groups[g * N + n] = sum_{k in group g} q_w[n][k] as fp32 (the constant part s (2^(b-1) - z) of a group's dequantised levels
 * meets the group's weight sum); the two products of a group are added to the fp32 accumulator in that order. */
int mq_gemm_w4a8_groupscale_asym(const int8_t *a, long lda, const void *w, int w_bits, long M, long N, long K_pad,
                                 const float *s_x_groups, const float *shift_groups, const float *wsum_groups,
                                 long n_groups, int group_k, const float *s_w, const float *bias, void *out,
                                 int out_dtype, long ldo, void *stream);

/* W4A8 Linear with the rotary position embedding of its q | k output columns folded into the store (the fused q|k|v projection of
 * a decoder layer).  NOT part of MQuant -- the reference never touches RoPE; it exists because the whole-prefill TTFT report chains
 * the Linears through the model's glue ("the step either side of the Linear").  Columns n < rope_cols are heads of
 * head_dim = 128 channels (one 128-wide output tile each, rope_cols % 128 == 0); cos / sin: [M, 128] in the output dtype, row = token
 * position.  Result = mq_gemm_w4a8 followed by mq_rope_inplace on the same columns, bit for bit:
 *   x = cast(linear output);  out = cast(cast(x * cos) + cast(rotate_half(x) * sin)),  rotate_half(x) = cat(-x[64:], x[:64]).
 * Needs tiled activations (lda = MQ_LD_TILED), a 16-bit output, N % 8 == 0; other head sizes: MQ_EUNSUPPORTED (run mq_rope_inplace). */
int mq_gemm_w4a8_rope_ws(const int8_t *a, long lda, const void *w, int w_bits, long M, long N, long K_pad, float s_x0, float s_x1,
                         const uint8_t *row_sel, const float *s_w, const float *bias, const void *rope_cos, const void *rope_sin,
                         long rope_cols, int head_dim, void *out, int out_dtype, long ldo, void *stream);

/* Activation of the CONSUMER folded into the GEMM's store (round 6).  The reference wraps every Linear on its own
 * (fake_quant/quant_utils.py:330-391); between two wrapped Linears the HF module applies its activation --
 * Qwen2MLP: down_proj(act_fn(gate_proj(x)) * up_proj(x)), the vision MLP: fc2(quick_gelu(fc1(x))) -- and the down_proj / fc2 wrapper
 * then reads that tensor (through the online Hadamard, fake_quant/hadamard_utils.py:115-128).  Here the producer GEMM stores the
 * ACTIVATED tensor: the Linear's output y is formed and rounded to out_dtype exactly as mq_gemm_w4a8_ws stores it, then
 *   act = MQ_ACT_SILU_MUL (1):   w is the image of a fused gate|up projection, gate = channels 0 .. N/2-1, up = N/2 .. N-1 (N / 2 a
 *                                multiple of 32); out[m][c] = cast(cast(silu(y[m][c])) * y[m][N/2 + c]) -- [M, N/2], ldo >= N/2
 *   act = MQ_ACT_QUICK_GELU (2): out[m][n] = cast(x * cast(sigmoid(cast(1.702 x)))), x = y[m][n] -- [M, N]
 * every torch op rounding once to out_dtype, as F.silu(gate) * up / QuickGELUActivation evaluate on tensors of that dtype (the same
 * device expf): bit-identical to mq_gemm_w4a8_ws followed by those torch ops.  Static scale set(s) (s_x0, s_x1, row_sel) or per-row
 * scales (s_x_rows); bias optional; no rank-1 terms.  Needs tiled activations (lda = MQ_LD_TILED: MQ_EUNSUPPORTED otherwise),
 * output columns and ldo multiples of 8, out 16-byte aligned. */
int mq_gemm_w4a8_act_ws(const int8_t *a, long lda, const void *w, int w_bits, long M, long N, long K_pad, float s_x0, float s_x1,
                        const uint8_t *row_sel, const float *s_x_rows, const float *s_w, const float *bias, int act,
                        void *out, int out_dtype, long ldo, void *stream);

/* out[m][n] = cast(y32[m][n] + x[m] * w[n]): a THIRD rank-1 term behind mq_gemm_w4a8_rank2_ws, for a layer that combines the split
 * column (--visual_split, quant_utils.py:367-376), asymmetric weights (--w_asym) and asymmetric dynamic activations (--a_asym): run
 * the two-slot GEMM with out_dtype = MQ_F32, then this pass -- the sum continues in fp32 and is rounded to the output dtype once,
 * exactly what a third epilogue slot would compute.  y32: fp32 [M, ldy]; out: [M, ldo] of out_dtype (may alias y32 for MQ_F32). */
int mq_rank1_add_cast(const float *y32, long M, long N, long ldy, const float *x, const float *w, void *out, int out_dtype,
                      long ldo, void *stream);

/* The GEMM for group-wise WEIGHT scales (--w_groupsize g; reference fake_quant/gptq/gptq_utils.py:263-273: the GPTQ solver re-runs
 * WeightQuantizer.find_params on every group of g consecutive input channels, so channel n carries one scale per group; flag at
 * exam/quant_qwen2vl.py:327).  s_w_groups[g * N + n] = that scale.
 *   y[m][n] = (sum_g ((float(acc_g[m][n]) * s_x_groups[m][g]) * s_w_groups[g][n])) * s_x(m) + bias[n]
 * acc_g = the exact int32 sum over the group's k; groups added in ascending order in fp32 (one rounding per product and per sum);
 * s_x(m) = s_x_rows[m] (dynamic per-token), or s_x1 where row_sel[m] else s_x0 (static, MSQ), or 1 when s_x_groups (group-wise
 * activation scales of the SAME group size, quant_utils.py:181-203; NULL otherwise: the factor is then 1 inside the sum).
 * Replaces F.linear on fake-quantized tensors (quant_utils.py:384): outputs agree with the reference to the 1e-3 of the other modes,
 * the per-group accumulators are exact.  group_k: 64 or a multiple of 128; n_groups * group_k covers K_pad up to its zero padding. */
int mq_gemm_w4a8_wgroupscale(const int8_t *a, long lda, const void *w, int w_bits, long M, long N, long K_pad,
                             const float *s_w_groups, long n_groups, int group_k, float s_x0, float s_x1,
                             const uint8_t *row_sel, const float *s_x_rows, const float *s_x_groups,
                             const float *bias, void *out, int out_dtype, long ldo, void *stream);

/* Scaled row sums of the int8 activation levels, for ASYMMETRIC weights (--w_asym; WeightQuantizer with
 * sym = False, quant_utils.py:446-509).  With the weight levels stored as q - 2^(b-1) the fake-quantized
 * weight is s_w[n] (stored + 2^(b-1) - z_w[n]); the zero points come back through the rank-1 epilogue term
 *   x0[m] = s_x(m) * sum_k a[m][k]   (this function; s_x(m) as in the GEMM: s_x_rows[m], or s_x1 where
 *                                     row_sel[m], else s_x0)
 *   w0[n] = s_w[n] * (2^(b-1) - z_w[n])
 * of mq_gemm_w4a8 / _rowscale_ws.  a: [M, K_pad] int8 with lda bytes per row or lda = MQ_LD_TILED. */
int mq_act_rowsum_scaled(const int8_t *a, long lda, long M, long K_pad, float s_x0, float s_x1,
                         const uint8_t *row_sel, const float *s_x_rows, float *out, void *stream);

/* TEST-ONLY hook, not part of the drop-in surface: force the tile shape (-1 heuristic; ids as in
 * csrc/gemm_w4a8.hip dispatch_tile) and the split-K factor (0 heuristic; bits 8..15 of a positive value
 * force the number of m-groups of the XCD mapping, 0 = automatic) for the GEMM calls the CALLING THREAD
 * makes afterwards (thread-local state; other threads keep the heuristic). */
int mq_gemm_debug_force(int tile, int splits);
/* TEST-ONLY: the fused activations of the 16-bit dtypes (csrc/mq_common.h act_silu_16 / act_sigmoid_16 and the packed forms of the GEMM act
 * epilogues) against the reference forms (device expf + IEEE division, what torch's silu / sigmoid kernels execute), element by element:
 * which = 0 silu(x), 1 sigmoid(x), 2 silu(x) * u, 3 quick_gelu(x), 4 / 5 the packed two-at-a-time forms of 2 / 3.  in_bits / in2_bits /
 * out_*: n 16-bit patterns of dtype (MQ_F16 or MQ_BF16); in2_bits may be NULL (u = 1).  tests/test_gpu_act_exhaustive.py runs all 2^16 inputs. */
int mq_debug_act_table(int dtype, int which, const void *in_bits, const void *in2_bits, long n, void *out_fast, void *out_ref, void *stream);
/* TEST-ONLY: the plan (tile id, split-K factor) the dispatcher takes for a shape; host arithmetic only. */
int mq_gemm_debug_plan(long M, long N, long K_pad, int w_bits, int a_tiled, int have_workspace, int *tile, int *splits);

/* ---------------------------------------------------------------------------
 * Min/max observer reduction.  Replaces the two reductions of
 * MinmaxObserver.update, fake_quant/observer/minmax.py:13-28 (after
 * BaseObserver.reshape_tensor, observer/base.py:15-28): per-channel min and max of
 * x[M, C] over rows.  col_begin skips leading channels (split: the observer only
 * sees x[..., 1:], quant_utils.py:369).  Results are fp32 [C - col_begin].
 * mq_minmax_tensor reduces to two scalars (layer_wise): out2[0]=min, out2[1]=max.
 * ------------------------------------------------------------------------- */
int mq_minmax_channels(const void *x, int x_dtype, long M, long C, long ldx, long col_begin,
                       float *mn, float *mx, void *stream);
int mq_minmax_tensor(const void *x, int x_dtype, long M, long C, long ldx, long col_begin,
                     float *out2, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* MQUANT_HIP_H */
