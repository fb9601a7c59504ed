#!/usr/bin/env python3
"""Clock readings of the product's own GEMM launches, next to tools/probes/clock_recon.hip.

For the gate|up launch (768 x 37888 x 3584, ping-pong 256 x 256, stamp build libmquant_hip_ppst.so) and the down_proj launch
(768 x 3584 x 19968, ws 96 x 128, timeline build libmquant_hip_wstl.so), with operand DATA of several kinds (the instruction
stream is identical; only a power-managed clock can tell them apart):
  * host-timed us per launch (HIP events, rotating weight copies = cold weights);
  * in-kernel s_memtime ticks per s_memrealtime tick (100 MHz): the shader clock each workgroup saw;
  * rocm-smi sclk / package power sampled by a side thread while the launch runs back to back for ~2.5 s.
usage (GPU box): python3 tools/clock_recon.py   (spawns itself per library)"""
import os
import subprocess
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def smi():
    try:
        out = subprocess.run("/opt/rocm/bin/rocm-smi --showpower --showclocks 2>/dev/null | grep -E 'sclk|Power' | tr -s ' \\t' ' ' | tr '\\n' ';'",
                             shell=True, capture_output=True, text=True, timeout=10).stdout
        return out.strip()
    except Exception as e:  # noqa: BLE001
        return f"error {e}"


def operands(kind, M, N, K, dev, torch):
    g = torch.Generator(device="cpu").manual_seed(7)
    if kind in ("random", "zeroA"):
        q = torch.randint(-8, 8, (N, K), dtype=torch.int8, generator=g)
    elif kind == "zero":
        q = torch.zeros((N, K), dtype=torch.int8)
    else:   # "bench-like": RTN levels of Gaussian weights (per-channel absmax / 7), Gaussian activations with outlier channels
        w = torch.randn((N, K), generator=g) * 0.02
        q = torch.clamp(torch.round(w / (w.abs().amax(dim=1, keepdim=True) / 7)), -8, 7).to(torch.int8)
    if kind == "random":
        a = torch.randint(-128, 128, (M, K), dtype=torch.int8, generator=g)
    elif kind in ("zero", "zeroA"):
        a = torch.zeros((M, K), dtype=torch.int8)
    else:
        x = torch.randn((M, K), generator=g)
        x[:, torch.randperm(K, generator=g)[: max(1, K // 1000)]] *= 20.0
        a = torch.clamp(torch.round(x / (x.abs().max() / 127)), -128, 127).to(torch.int8)
    return a.to(dev), q.to(dev)


def worker(which):
    import torch
    from mquant_amd import ops
    dev = torch.device("cuda:0")
    M = 768
    if which == "pp":
        N, K, tile, per, label = 37888, 3584, 14, 8, "gate|up 768 x 37888 x 3584, ping-pong 256 x 256 (444 tiles, persistent on 256 CUs)"
    else:
        N, K, tile, per, label = 3584, 19968, 40, 16, "down_proj 768 x 3584 x 19968, ws 96 x 128 (224 workgroups)"
    print(f"== {label}")
    for kind in ("random", "bench-like", "zeroA", "zero"):
        a8, q = operands(kind, M, N, K, dev, torch)
        a = ops.TiledAct.from_rows(a8)
        copies = [ops.prepack(q, 4) for _ in range(6)]
        s_w = torch.full((N,), 0.01, device=dev)
        out = torch.empty((M, N), dtype=torch.float16, device=dev)
        ops.gemm_debug_force(tile, 1)
        ws = ops.splitk_workspace(dev)
        ws.zero_()
        st = {"i": 0}

        def call():
            st["i"] = (st["i"] + 1) % len(copies)
            ops.gemm_w4a8(a, copies[st["i"]], 4, N, 0.02, s_w, out=out)
        for _ in range(10):
            call()
        torch.cuda.synchronize()
        samples, stop = [], threading.Event()

        def watcher():
            time.sleep(0.5)
            while not stop.is_set():
                samples.append(smi())
                time.sleep(0.3)
        th = threading.Thread(target=watcher)
        th.start()
        t_end = time.time() + 2.5
        best, n = 1e9, 0
        while time.time() < t_end:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(200):
                call()
            e1.record()
            torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) / 200 * 1e3)
            n += 200
        stop.set()
        th.join()
        t = ws[: 512 * per * 4].view(torch.int32).view(-1, per).cpu().to(torch.int64)
        if which == "pp":
            rows = t[t[:, 1] != 0]
            cyc, real = rows[:, 0] & 0xFFFFFFFF, (rows[:, 2] & 0xFFFFFFFF).clamp_min(1)
            ghz = (cyc.double() / (real.double() * 10.0))
            extra = f"k-loop {cyc.double().median().item() / rows[0, 1].item():.0f} ticks per k-tile (MFMAs need 1024)"
        else:
            rows = t[t[:, 1] != 0]
            cyc = (rows[:, 11] - rows[:, 0]) & 0xFFFFFFFF
            real = ((rows[:, 12] - rows[:, 1]) & 0xFFFFFFFF).clamp_min(1)
            ghz = (cyc.double() / (real.double() * 10.0))
            loop = ((rows[:, 7] - rows[:, 6]) & 0xFFFFFFFF).double()
            extra = f"k-loop {loop.median().item() / rows[0, 13].item() / 2:.0f} ticks per 64-wide k-tile (MFMAs need 192)"
        ops_ = 2.0 * M * N * K
        print(f"  data {kind:10s}: {best:7.2f} us per launch ({ops_ / best * 1e-6 / 1e3:.0f} TOP/s, {ops_ / best * 1e-6 / 5e6:.3f} of 5 POP/s), in-kernel s_memtime / s_memrealtime "
              f"median {ghz.median().item():.3f} GHz (min {ghz.min().item():.3f} max {ghz.max().item():.3f}, {len(rows)} stamps), {extra}")
        for s in samples[:3]:
            print(f"      rocm-smi: {s}")
        sys.stdout.flush()
        del copies, a, q


def main():
    if len(sys.argv) > 1:
        worker(sys.argv[1])
        return
    for which, lib in (("pp", "libmquant_hip_ppst.so"), ("ws", "libmquant_hip_wstl.so")):
        env = dict(os.environ, MQUANT_HIP_LIB=os.path.join(ROOT, "mquant_amd", lib))
        r = subprocess.run([sys.executable, __file__, which], env=env, capture_output=True, text=True)
        print(r.stdout, end="")
        if r.returncode:
            print(r.stderr[-2000:])
        sys.stdout.flush()


if __name__ == "__main__":
    main()
