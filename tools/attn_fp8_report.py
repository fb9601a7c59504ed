#!/usr/bin/env python3
"""mq_attn_prefill_fp8kv against the path it replaces (mq_kv_dequant_fp8 -> fp16 K/V in HBM -> torch SDPA) and against
plain fp16 SDPA, at the attention shapes of the benchmark prefill (768 tokens; 7B: 28 / 4 heads, 72B: 64 / 8).
Output -> profiles/r3_attn_fp8kv.txt."""
import os
import sys

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
torch.set_grad_enabled(False)
from mquant_amd import ops  # noqa: E402

dev = "cuda:0"


def timed(fn, iters=50):
    fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(10):
            fn()
    g.replay()
    torch.cuda.synchronize()
    ts = []
    for _ in range(iters):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        g.replay()
        e1.record()
        e1.synchronize()
        ts.append(e0.elapsed_time(e1) * 100.0)          # us per call
    ts.sort()
    return ts[len(ts) // 2]


for name, T, H, HKV in (("Qwen2-VL-7B", 768, 28, 4), ("Qwen2-VL-72B", 768, 64, 8), ("7B heads, 4096 tokens", 4096, 28, 4)):
    D = 128
    g = torch.Generator(device=dev).manual_seed(1)
    qkv = torch.randn(T, (H + 2 * HKV) * D, generator=g, device=dev).half()
    q = qkv[:, :H * D].view(T, H, D)
    kv = qkv[:, H * D:].view(T, 2 * HKV, D)
    scale = ops.kv_scale_from_absmax(kv)
    cache = ops.kv_quant_fp8(kv, scale)
    out = torch.empty(T, H * D, device=dev, dtype=torch.float16)
    kvd = torch.empty(T, 2 * HKV, D, device=dev, dtype=torch.float16)
    rep = H // HKV

    def sdpa(kv16):
        k, v = kv16[:, :HKV], kv16[:, HKV:]
        o = F.scaled_dot_product_attention(q.permute(1, 0, 2)[None], k.permute(1, 0, 2)[None], v.permute(1, 0, 2)[None],
                                           is_causal=True, enable_gqa=True)
        return o[0].permute(1, 0, 2).reshape(T, H * D)

    t_ours = timed(lambda: ops.attn_prefill_fp8kv(q, cache, scale, out=out))
    kk, vv = kv[:, :HKV], kv[:, HKV:]
    t_16 = timed(lambda: ops.attn_prefill(q, kk, vv, out=out))
    t_deq = timed(lambda: sdpa(ops.kv_dequant_fp8(cache, scale, torch.float16, out=kvd)))
    t_f16 = timed(lambda: sdpa(kv))
    a, b = ops.attn_prefill_fp8kv(q, cache, scale).float(), sdpa(ops.kv_dequant_fp8(cache, scale, torch.float16)).float()
    flops = 4.0 * T * T * D * H / 2                       # causal: half of the score matrix
    print(f"{name}: T={T} heads={H}/{HKV} head_dim=128 causal, fp16, hipGraph replay, median of 50")
    print(f"  mq_attn_prefill_fp8kv (reads e4m3)            : {t_ours:8.2f} us   ({flops / t_ours / 1e6:6.1f} TFLOP/s causal-useful)")
    print(f"  mq_attn_prefill (16-bit K / V, same kernel)   : {t_16:8.2f} us   ({flops / t_16 / 1e6:6.1f} TFLOP/s causal-useful)")
    print(f"  mq_kv_dequant_fp8 + torch SDPA (fp16 copy)    : {t_deq:8.2f} us")
    print(f"  torch SDPA on the unquantised fp16 K/V        : {t_f16:8.2f} us")
    print(f"  max |ours - dequant+SDPA| / max|.| = {float((a - b).abs().max() / b.abs().max()):.2e}")


# the vision tower's attention: 1024 patches, 16 heads of 80, non-causal (q / k / v = the three column blocks of the qkv GEMM output)
T, H, D = 1024, 16, 80
qkv = torch.randn(T, 3 * H * D, device=dev).half()
q, k, v = qkv.view(T, 3, H, D).unbind(1)
out = torch.empty(T, H * D, device=dev, dtype=torch.float16)
t_own = timed(lambda: ops.attn_prefill(q, k, v, causal=False, out=out))
t_sdpa = timed(lambda: F.scaled_dot_product_attention(q.transpose(0, 1)[None], k.transpose(0, 1)[None], v.transpose(0, 1)[None])[0].transpose(0, 1).reshape(T, H * D))
a = ops.attn_prefill(q, k, v, causal=False).float()
b = F.scaled_dot_product_attention(q.transpose(0, 1)[None], k.transpose(0, 1)[None], v.transpose(0, 1)[None])[0].transpose(0, 1).reshape(T, H * D).float()
print(f"Qwen2-VL vision tower: T={T} heads={H} head_dim={D} non-causal, fp16")
print(f"  mq_attn_prefill                               : {t_own:8.2f} us   ({4.0 * T * T * D * H / t_own / 1e6:6.1f} TFLOP/s)")
print(f"  torch SDPA (+ the transpose copy proj needs)  : {t_sdpa:8.2f} us")
print(f"  max |ours - SDPA| / max|.| = {float((a - b).abs().max() / b.abs().max()):.2e}")
