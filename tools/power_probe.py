#!/usr/bin/env python3
"""What the chip reports while one kernel family runs back to back: shader clock and package power (rocm-smi), sampled from a
side thread during ~3 s of (a) the 256 x 256 gate|up GEMM, (b) the down_proj GEMM, (c) the exact Hadamard + quantize kernel of
down_proj, (d) an idle pause.  Evidence for DESIGN 4.1: the GEMM family runs at the power limit (clock well below 2.4 GHz)."""
import json
import os
import subprocess
import sys
import threading
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mquant_amd import ops  # noqa: E402

dev = torch.device("cuda:0")


def sample():
    try:
        out = subprocess.run(["/opt/rocm/bin/rocm-smi", "--showpower", "--showclocks", "--showtemp", "--json"], capture_output=True, text=True, timeout=10).stdout
        d = json.loads(out)
        card = d[sorted(d)[0]]
        keep = {k: v for k, v in card.items() if any(t in k.lower() for t in ("power", "sclk", "mclk", "junction", "fclk"))}
        return keep
    except Exception as e:  # noqa: BLE001
        return {"error": str(e)[:200]}


def run(name, fn, seconds=3.0):
    samples, stop = [], threading.Event()

    def watcher():
        while not stop.is_set():
            samples.append(sample())
            time.sleep(0.2)
    th = threading.Thread(target=watcher)
    th.start()
    t0 = time.time()
    n = 0
    while time.time() - t0 < seconds:
        for _ in range(50):
            fn()
        torch.cuda.synchronize()
        n += 50
    dt = time.time() - t0
    stop.set()
    th.join()
    print(f"== {name}: {n} calls, {dt / n * 1e6:.1f} us per call")
    for s in samples[2:8]:
        print("   ", s)
    sys.stdout.flush()


def main():
    print("rocm-smi caps:", subprocess.run(["/opt/rocm/bin/rocm-smi", "--showmaxpower", "--showperflevel"], capture_output=True, text=True).stdout[-600:])
    M = 768
    for name, N, K in (("gate|up 768x37888x3584", 37888, 3584), ("down_proj 768x3584x19968", 3584, 19968)):
        a = ops.TiledAct.from_rows(torch.randint(-128, 128, (M, K), dtype=torch.int8, device=dev))
        q = torch.randint(-8, 8, (N, K), dtype=torch.int8, device=dev)
        imgs = [ops.prepack(q, 4) for _ in range(6)]
        s_w = torch.full((N,), 0.01, device=dev)
        out = torch.empty((M, N), dtype=torch.float16, device=dev)
        st = {"i": 0}

        def call():
            st["i"] = (st["i"] + 1) % len(imgs)
            ops.gemm_w4a8(a, imgs[st["i"]], 4, N, 0.02, s_w, out=out)
        run(name, call)
        del imgs, q
    from fake_quant import hadamard_utils
    n, Kh = 19968, 156
    bits = hadamard_utils.had_sign_bits(Kh, dev)
    x = torch.randn((M, 18944), dtype=torch.float16, device=dev)
    run("hadamard+quant 768x18944->19968 (K=156)", lambda: ops.hadamard_quant_i8(x, n, Kh, bits, 0.05, tiled=True))
    run("idle (sleep)", lambda: time.sleep(0.001), seconds=1.5)


if __name__ == "__main__":
    main()
