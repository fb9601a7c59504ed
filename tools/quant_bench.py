#!/usr/bin/env python3
"""Static activation quantizer alone (mq_quantize_act_i8, tiled int8 output) at the two geometries the Qwen2-VL-7B prefill
launches it on, replayed from a hipGraph over ROTATING input / output buffers (the input of a launch was written by another
kernel a moment ago: it sits in the Infinity Cache or HBM, not in the L2 of the XCD that reads it).
Usage (GPU box): python tools/quant_bench.py [--libs main,v1,...]   (variant libraries: tools/build_variant_lib.sh)"""
import argparse
import hashlib
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def worker():
    import torch
    sys.path.insert(0, ROOT)
    from mquant_amd import ops
    dev = torch.device("cuda:0")
    shapes = [("llm 768 x 3584", 768, 3584), ("vit 1024 x 1280", 1024, 1280), ("llm 768 x 18944 (no Hadamard)", 768, 18944)]
    for name, M, K in shapes:
        for dt in (torch.float16, torch.bfloat16):
            for msq in (False, True):
                g = torch.Generator(device="cpu").manual_seed(7)
                NB = 16
                xs = [(torch.randn((M, K), generator=g) * 2.0).to(dt).to(dev) for _ in range(NB)]
                outs = [ops.TiledAct.empty(M, (K + 127) // 128 * 128, dev) for _ in range(NB)]
                sel = None
                if msq:
                    sel = torch.zeros((M,), dtype=torch.uint8, device=dev)
                    sel[: M // 3] = 1

                def one(i):
                    ops.quantize_act_i8(xs[i], 0.0371, 0.0212, row_sel=sel, out=outs[i])
                for i in range(NB):
                    one(i)
                torch.cuda.synchronize()
                h = hashlib.sha256()
                for o in outs[:4]:
                    h.update(o.data.cpu().numpy().tobytes())
                graph = torch.cuda.CUDAGraph()
                s = torch.cuda.Stream()
                with torch.cuda.stream(s):
                    with torch.cuda.graph(graph, stream=s):
                        for r in range(10):
                            for i in range(NB):
                                one(i)
                for _ in range(3):
                    graph.replay()
                torch.cuda.synchronize()
                best = 1e9
                for _ in range(5):
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    for _ in range(5):
                        graph.replay()
                    e1.record()
                    torch.cuda.synchronize()
                    best = min(best, e0.elapsed_time(e1) / (5 * 10 * NB) * 1e3)
                byts = M * K * xs[0].element_size() + M * K
                print(f"  {name:30s} {str(dt)[6:]:9s} {'two scales' if msq else 'one scale ':10s}: {best:6.2f} us per launch "
                      f"({byts / best / 1e6:5.2f} TB/s)  levels sha {h.hexdigest()[:12]}", flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--libs", default="main")
    ap.add_argument("--worker", action="store_true")
    args = ap.parse_args()
    if args.worker:
        worker()
        return
    for lib in args.libs.split(","):
        env = dict(os.environ)
        if lib != "main":
            env["MQUANT_HIP_LIB"] = os.path.join(ROOT, "mquant_amd", f"libmquant_hip_{lib}.so")
        print(f"== {lib}", flush=True)
        subprocess.call([sys.executable, os.path.abspath(__file__), "--worker"], env=env)


if __name__ == "__main__":
    main()
