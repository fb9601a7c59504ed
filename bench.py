#!/usr/bin/env python3
"""Headline benchmark: W4A8 prefill tokens/s of the hot path on the Qwen2-VL-7B workload
(1 x 448^2 image + 512-token prompt => 1024 vision tokens, 768 LLM positions).

A "step" is one pass of the hot path over one image+prompt: every wrapped Linear of the
prefill (ViT 32 blocks x 4 + patch_embed + merger x 2, LLM 28 layers x 7 = 327 Linears), each
as [online Hadamard +] static int8 quantize -> int8 x int4 MFMA GEMM with fused dequant,
inputs resident in HBM.  Attention, norms, RoPE and lm_head are outside the path (SURVEY.md
section 8) and are not executed.  Weights are random with the real shapes; data is synthetic.

    python bench.py --gpus N --steps K --warmup W
N > 1 is launched by torch.distributed.run, one rank per GPU; every rank prefills its own
sample (batch sharding, weak scaling) and the ranks exchange last-token logits with one RCCL
all_gather per step.  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

PEAK_INT8_TOPS = 5000.0   # dense int8 MFMA, 2 x the 2.5 PF bf16 dense peak (MI355X_MICROARCH.md)
VOCAB = 152064


def cpu_baseline(prefill_ops_total: float):
    """Time the CPU oracle ("port" of the reference's fake-quant path, integer form) on a
    bounded sample: one LLM layer's 7 Linears restricted to 256 of the 768 rows."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import numpy as np
    import oracle
    from golden_inputs import make_w, make_x

    rows = 256
    t = np.load(os.path.join(ROOT, "tests", "golden", "hadk_table.npz"))
    hk = np.unpackbits(t["had156"])[: 156 * 156].reshape(156, 156).astype(np.int8) * 2 - 1
    shapes = [(3584, 3584, 3584, 0), (3584, 3584, 512, 0), (3584, 3584, 512, 0),
              (3584, 3584, 3584, 0), (3584, 3584, 18944, 0), (3584, 3584, 18944, 0),
              (18944, 19968, 3584, 156)]
    prepared = []
    pool = np.random.RandomState(950).randint(-8, 8, size=19968 * 3584 + 8 * 19968, dtype=np.int8)
    for i, (k_in, k, n, hK) in enumerate(shapes):
        x = make_x(900 + i, (rows, k_in))
        w = pool[i * 19968: i * 19968 + n * k].reshape(n, k)   # untimed preparation
        s_w = np.full((n,), 0.003, dtype=np.float32)
        prepared.append((x, w, s_w, k, hK))
    ops_sample, reps = 0.0, 0
    t0 = time.perf_counter()
    while reps < 16 and (reps == 0 or time.perf_counter() - t0 < 10.0):     # ~10 s of CPU work
        for x, w, s_w, k, hK in prepared:
            if hK:
                x = oracle.hadamard(x, k, hK, hk, mid_round=1, out_round=1)
            q = oracle.quant_static(x, np.float32(0.05))
            acc = oracle.gemm_i32(q, w)
            oracle.epilogue(acc, np.float32(0.05), s_w)
            ops_sample += 2.0 * rows * k * w.shape[0]
        reps += 1
    dt = time.perf_counter() - t0
    est_step = dt * prefill_ops_total / ops_sample
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    return {"value": round(768.0 / est_step, 3), "unit": "tokens/s", "cores": cores, "kind": "port",
            "sample": f"oracle/mq_oracle.c (OpenMP, {cores} threads): LLM layer 0, 7 Linears incl. "
                      f"pad+Hadamard(156x128)+quant, {rows} of 768 rows, {reps} repetition(s), {dt:.1f} s measured; "
                      f"extrapolated by GEMM ops to the full 327-Linear prefill"}


def full_prefill_report(pf, dev, args):
    """TTFT of the WHOLE synthetic prefill -- the W4A8 Linears chained through
    torch glue (RMS norm, RoPE, SDPA, activations, fp16 lm_head on the last position) -- one
    hipGraph replay per sample, HIP events around each replay.  Secondary to ``value``."""
    import torch
    from mquant_amd import workload
    from mquant_amd.full_prefill import FullPrefill
    try:
        def measure(fused):
            fp = FullPrefill(pf, fused_glue=fused)
            fp.calibrate()
            if args.no_graph:
                run = fp.step
            else:
                fp.step()
                torch.cuda.synchronize(dev)
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g):
                    fp.step()
                run = g.replay
            for _ in range(20):
                run()
            torch.cuda.synchronize(dev)
            times = []
            for _ in range(max(args.ttft_iters, 5)):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                run()
                e1.record()
                e1.synchronize()
                times.append(e0.elapsed_time(e1))
            times.sort()
            med = times[len(times) // 2]
            p90 = times[min(len(times) - 1, int(round(0.9 * (len(times) - 1))))]
            return med, p90, len(times), bool(torch.isfinite(fp.logits.float()).all().item())
        med_u, p90_u, _, _ = measure(False)
        med, p90, iters, finite = measure(True)
        return {"what": "whole synthetic prefill: W4A8 Linears (this repo's kernels) + torch SDPA and fp16 lm_head "
                        "on the last position; RMS norm -> quantize, SiLU*up / QuickGELU -> Hadamard -> quantize, "
                        "residual adds (GEMM epilogue) and RoPE (one in-place launch) run fused; "
                        "ttft_ms_median_unfused_glue = the same dataflow with those steps as separate torch ops",
                "ttft_ms_median": round(med, 4), "ttft_ms_p90": round(p90, 4), "iters": iters,
                "ttft_ms_median_unfused_glue": round(med_u, 4), "ttft_ms_p90_unfused_glue": round(p90_u, 4),
                "llm_tokens_per_s": round(workload.M_LLM / (med * 1e-3), 1),
                "all_tokens_per_s": round((workload.M_LLM + workload.M_VIS) / (med * 1e-3), 1),
                "logits_finite": finite}
    except Exception as exc:      # a report, never a reason to lose the bench line
        return {"error": repr(exc)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--tiny", action="store_true", help="small shapes (debug only; not a valid bench line)")
    ap.add_argument("--no-graph", action="store_true", help="launch kernels one by one instead of replaying a hipGraph")
    ap.add_argument("--no-fuse", action="store_true", help="one GEMM per Linear (no q/k/v, gate/up fusion)")
    ap.add_argument("--no-full-prefill", action="store_true",
                    help="skip the secondary report: whole synthetic prefill incl. attention/norms (torch glue)")
    ap.add_argument("--ttft-iters", type=int, default=100)
    ap.add_argument("--batch", type=int, default=1,
                    help="image+prompt samples per GPU and step (the benchmark configuration is 1; >1 is a scaling study)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    distributed = world > 1
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if distributed:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="nccl", device_id=dev)

    from mquant_amd import workload

    specs = workload.tiny_specs() if args.tiny else workload.qwen2vl_7b_specs(msq=True, batch=args.batch)
    pf = workload.Prefill(specs, device=dev, dtype=torch.float16, share_groups=not args.no_fuse)
    tokens_per_step = workload.M_LLM * args.batch if not args.tiny else specs[-1].M

    logits_local = torch.zeros((1, VOCAB), dtype=torch.float16, device=dev)
    logits_all = torch.zeros((world, VOCAB), dtype=torch.float16, device=dev) if distributed else None

    def capture(fn):
        """Launch-bound inner loop -> one hipGraph (the kernels themselves are unchanged)."""
        if args.no_graph:
            return fn
        fn()                      # first-call allocations / attribute setup happen outside capture
        torch.cuda.synchronize(dev)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            fn()
        return g.replay

    run_prefill = capture(pf.step)

    def step():
        run_prefill()
        if distributed:
            dist.all_gather_into_tensor(logits_all, logits_local)

    def fence():
        if distributed:
            dist.barrier()
        torch.cuda.synchronize(dev)

    for _ in range(args.warmup):
        step()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    elapsed = time.perf_counter() - t0
    if distributed:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
    ms_per_step = elapsed / args.steps * 1e3
    value = world * tokens_per_step * args.steps / elapsed

    # ---- kernel attribution: the GEMM launches of a step alone, HIP events on the launch stream
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = max(3, min(args.steps, 10))
    run_gemms = capture(pf.step_gemm_only)
    run_quants = capture(pf.step_quant_only)
    run_gemms()
    torch.cuda.synchronize(dev)
    e0.record()
    for _ in range(reps):
        run_gemms()
    e1.record()
    torch.cuda.synchronize(dev)
    gemm_ms = e0.elapsed_time(e1) / reps
    e0.record()
    for _ in range(reps):
        run_quants()
    e1.record()
    torch.cuda.synchronize(dev)
    quant_ms = e0.elapsed_time(e1) / reps
    launches = pf.gemm_launches()
    traffic, traffic_note = None, "no profiles/r1_traffic.json"
    tpath = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "r1_traffic.json")
    if os.path.exists(tpath) and not args.tiny and args.batch == 1:
        # HBM bytes per GEMM launch from the PMC passes (tools/pmc_traffic.py); counters cannot be
        # read inside the timed run, so this is the committed measurement of the same command
        with open(tpath) as fh:
            tj = json.load(fh)
        traffic = tj["kernels"].get("gemm", {}).get("hbm_bytes_per_launch")
        traffic_note = tj["corrections"]
    achieved = pf.gemm_ops() / (gemm_ms * 1e-3) / 1e12
    roofline = {"bound": "mfma", "kernel": "gemm_w4a8_kernel (V_MFMA_I32_16X16X64_I8)",
                "achieved": round(achieved, 2), "peak": PEAK_INT8_TOPS, "unit": "TOP/s",
                "frac": round(achieved / PEAK_INT8_TOPS, 4), "traffic": traffic,
                "traffic_unit": "HBM bytes per launch (PMC FETCH_SIZE x2 + WRITE_SIZE)", "traffic_note": traffic_note,
                "algorithmic_bytes_per_launch": round(pf.gemm_bytes() / launches),
                "launches_per_step": launches,
                "avg_launch_us": round(gemm_ms * 1e3 / launches, 3),
                "algorithmic_ops_per_launch": round(pf.gemm_ops() / launches),
                "gemm_ms_per_step": round(gemm_ms, 4),
                "quant_hadamard_ms_per_step": round(quant_ms, 4),
                "quant_hadamard_GBps": round(pf.quant_bytes() / (quant_ms * 1e-3) / 1e9, 1)}

    line = {"metric": "W4A8 prefill tokens/sec (hot path: Hadamard + static quant + W4A8 Linear), "
                      "Qwen2-VL-7B 448px+512tok",
            "value": round(value, 1), "unit": "tokens/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "int8",
            "data": "synthetic (random weights with the real shapes, random activations with outlier channels)",
            "config": {"workload": "Qwen2-VL-7B W4A8 MSQ prefill, 1x448^2 image (1024 vision tokens) + "
                                   "512 text tokens, 327 wrapped Linears, M_llm=768" + ("" if args.batch == 1 else f", x{args.batch} samples per step (scaling study, not the benchmark configuration)") + ("" if args.no_fuse else " (q/k/v and gate/up share one quantization and one GEMM)") if not args.tiny
                       else "tiny debug shapes",
                       "tokens_per_step_per_gpu": tokens_per_step, "parallelism": f"batch-shard x{world}",
                       "ttft_hot_path_ms": round(ms_per_step, 4),
                       "gemm_TOP_per_step": round(pf.gemm_ops() / 1e12, 3),
                       "hip_graph": not args.no_graph,
                       "weights_GB": round(pf.weight_bytes() / 1e9, 3)},
            "roofline": roofline}
    if not (args.tiny or args.no_full_prefill or args.no_fuse or args.batch != 1):
        line["full_prefill"] = full_prefill_report(pf, dev, args)
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        try:
            line["cpu_baseline"] = cpu_baseline(float(pf.gemm_ops()))
        except Exception as exc:  # the baseline is a report, never a reason to lose the line
            line["cpu_baseline"] = {"value": None, "unit": "tokens/s", "cores": 0, "kind": "port",
                                    "sample": f"failed: {exc!r}"}
    if rank == 0:
        print(json.dumps(line), flush=True)
    if distributed:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
