/* mquant_bench.h -- BENCH-ONLY entry points (libmquant_bench.so, built from csrc/bench_probe.hip).
 * Not part of the drop-in surface and not linked into libmquant_hip.so: bench.py loads this library for its
 * roofline.peak_sustained_measured reading, nothing under fake_quant/ or mquant_amd/ ever does. */
#ifndef MQUANT_BENCH_H
#define MQUANT_BENCH_H
#ifdef __cplusplus
extern "C" {
#endif

/* The dense int8 matrix rate the device sustains under its package power limit.
 * Register-only V_MFMA_I32_32X32X32_I8 (kind 0) / V_MFMA_I32_16X16X64_I8 (kind 1) chains, two waves per SIMD on every CU, no memory or
 * LDS traffic; operands: 8 x 64 x 16 bytes on the device (four A and four B fragment register sets, rotated over the MFMAs --
 * the caller supplies bytes with the statistics of its workload: the clock the part holds depends on them,
 * profiles/r5_clock_reconciliation.txt).  Runs 3 + launches launches of iters x 16 (kind 0) / x 32 (kind 1) MFMAs per wave, times
 * the last `launches` with HIP events on `stream` (blocking) and returns the achieved int8 ops per second.  sink: 4 bytes of
 * device scratch.  bench.py reports the result as roofline.peak_sustained_measured beside the nominal peak. */
int mq_bench_mfma_burn(int kind, const void *operands, int iters, int launches, int *sink, double *ops_per_s, void *stream);

/* status text of the last failing call of this library on the calling thread */
const char *mq_bench_last_error(void);

#ifdef __cplusplus
}
#endif
#endif /* MQUANT_BENCH_H */
