#!/usr/bin/env python3
"""Where the fixed cost of a wave-specialised GEMM launch goes: per-workgroup cycle stamps from the
timeline build of gemm_ws.hip (-DMQ_WS_TL -> mquant_amd/libmquant_hip_wstl.so), G launches back to back from one hipGraph,
every launch with its own stamp area (a slice of the split-K workspace) and its own copy of the weight image (cold weights,
as in the prefill).  s_memrealtime (100 MHz, chip-wide) places the workgroups of consecutive launches on one time axis;
s_memtime gives the phases inside a workgroup; their ratio is the shader clock the launch ran at.

usage (GPU box): MQUANT_HIP_LIB=mquant_amd/libmquant_hip_wstl.so python3 tools/gemm_timeline.py [--shapes name,...]"""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mquant_amd import ops  # noqa: E402

SHAPES = {  # name: (M, N, K)   the launches of the Qwen2-VL-7B prefill that run on gemm_ws_kernel
    "vit_proj": (1024, 1280, 1280),
    "vit_fc2": (1024, 1280, 5120),
    "vit_qkv": (1024, 3840, 1280),
    "vit_fc1": (1024, 5120, 1280),
    "llm_o": (768, 3584, 3584),
    "llm_qkv": (768, 4608, 3584),
    "llm_down": (768, 3584, 19968),
}


def u32(t):
    return t.to(torch.int64) & 0xFFFFFFFF


def run(name, M, N, K, G, dev):
    a = ops.TiledAct.from_rows(torch.randint(-128, 128, (M, K), dtype=torch.int8, device=dev))
    q = torch.randint(-8, 8, (N, K), dtype=torch.int8, device=dev)
    imgs = [ops.prepack(q, 4) for _ in range(G)]
    s_w = torch.full((N,), 0.01, device=dev)
    outs = [torch.empty((M, N), dtype=torch.float16, device=dev) for _ in range(2)]
    ws = torch.zeros((G, 1 << 20), dtype=torch.int32, device=dev)            # 4 MiB of stamp area per launch
    tile = torch.zeros(2, dtype=torch.int32)
    ops.call("mq_gemm_debug_plan", M, N, a.K_pad, 4, 1, 1, tile[0:].data_ptr(), tile[1:].data_ptr())
    aptr, lda, _, K_pad = ops._a_args(a)

    def launches():
        for g in range(G):
            ops.call("mq_gemm_w4a8_ws", aptr, lda, imgs[g].data_ptr(), 4, M, N, K_pad, 0.02, 0.02, 0, s_w.data_ptr(),
                     0, 0, 0, outs[g & 1].data_ptr(), ops.dtype_code(torch.float16), outs[g & 1].stride(0),
                     ws[g].data_ptr(), ws[g].numel() * 4, ops._stream())
    launches()
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        with torch.cuda.graph(graph, stream=st, capture_error_mode="thread_local"):
            launches()
        for _ in range(5):
            graph.replay()
        st.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(st)
        R = 20
        for _ in range(R):
            graph.replay()
        e1.record(st)
        st.synchronize()
    host_us = e0.elapsed_time(e1) / (R * G) * 1e3
    t = ws.cpu()
    # number of workgroups = rows with a non-zero entry stamp
    nwg = int((t[0, : 4096 * 16].view(-1, 16)[:, 1] != 0).sum().item())
    tl = t[:, : nwg * 16].view(G, nwg, 16)
    ent_c, ent_r = u32(tl[..., 0]), u32(tl[..., 1])
    end_c, end_r = u32(tl[..., 11]), u32(tl[..., 12])
    nk = int(tl[0, 0, 13].item())

    def d(i, j):                                  # stamp i - stamp j in ticks, wrap-safe
        return (u32(tl[..., i]) - u32(tl[..., j])) & 0xFFFFFFFF
    wg_cyc = (end_c - ent_c) & 0xFFFFFFFF
    wg_real = ((end_r - ent_r) & 0xFFFFFFFF).clamp_min(1)
    ghz = (wg_cyc.double() / (wg_real.double() * 10.0))      # ticks per ns
    clk = ghz[1:].median().item()
    ns = lambda ticks: ticks.double() / clk                  # noqa: E731

    first = ent_r.min(dim=1).values
    last_end = end_r.max(dim=1).values
    dur = ((last_end - first) & 0xFFFFFFFF).double() * 10.0
    gap = ((first[1:] - last_end[:-1]) & 0xFFFFFFFF).double() * 10.0
    ramp = ((ent_r - first[:, None]) & 0xFFFFFFFF).double() * 10.0
    tail = ((last_end[:, None] - end_r) & 0xFFFFFFFF).double() * 10.0

    def row(label, x):
        x = x[1:].flatten()                       # launch 0 follows an idle queue: not the steady state
        print(f"    {label:58s} median {x.median().item():8.0f}  min {x.min().item():8.0f}  max {x.max().item():8.0f}  ns")
    print(f"== {name}: {M} x {N} x {K}, tile {tile[0].item()}, {nwg} workgroups, {nk} k-steps of 128; host-timed {host_us:.2f} us per launch "
          f"(graph of {G}, cold weights); shader clock in the launch {clk:.2f} GHz (s_memtime / s_memrealtime per workgroup, median)")
    row("launch: first workgroup entry -> last workgroup done", dur[:, None])
    row("gap: last workgroup of launch g done -> first entry of g+1", torch.cat([gap[:1], gap])[:, None])
    row("dispatch ramp: workgroup entry after the launch's first entry", ramp)
    row("math wave 0's entry -> loader wave 0's first instruction (wave launch order)", ns(d(3, 0)))
    row("loader 0: first instruction -> piece addresses ready (tile map, 64-bit address arithmetic)", ns(d(14, 3)))
    row("entry -> loader 0 has its piece addresses", ns(d(14, 0)))
    row("   -> stage 0 requested (LPW LDS-DMA instructions)", ns(d(15, 14)))
    row("   -> stages 1, 2 requested", ns(d(4, 15)))
    row("entry -> first 3 stages requested (loader 0: addresses + issue)", ns(d(4, 0)))
    row("requested -> stage 0 landed (loader 0's counted vmcnt)", ns(d(5, 4)))
    row("landed -> math wave 0 past B(0)", ns(d(6, 5)))
    row(f"k-loop, {nk} steps (math wave 0)", ns(d(7, 6)))
    row("   per k-step", ns(d(7, 6)) / max(nk, 1))
    row("loop end -> accumulators parked in the slab (2 barriers)", ns(d(8, 7)))
    row("slab -> wave 0's last store issued (reads, arithmetic, stores)", ns(d(9, 8)))
    row("issued -> wave 0's stores acknowledged", ns(d(10, 9)))
    row("-> every wave's stores acknowledged (barrier)", ns(d(11, 10)))
    row("whole workgroup: entry -> done", ns(wg_cyc))
    row("tail: workgroup done -> the launch's last workgroup done", tail)
    sys.stdout.flush()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--shapes", default=",".join(SHAPES))
    ap.add_argument("--graph", type=int, default=6)
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    ops.splitk_workspace(dev)
    for name in args.shapes.split(","):
        M, N, K = SHAPES[name]
        run(name, M, N, K, args.graph, dev)


if __name__ == "__main__":
    main()
