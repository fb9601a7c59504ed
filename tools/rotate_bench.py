"""Time of the offline rotation W <- (W.double() @ Q).to(dtype) on the GPU: the
structured kernel (mq_rotate_f64: sign flip + fast Hadamard per row) beside the dense fp64 product
the reference evaluates, on Qwen2-VL-7B's weight shapes.  Prints one table; run on the GPU box."""
import sys
import time

import torch

sys.path.insert(0, ".")
from fake_quant import rotation_utils as ru  # noqa: E402

DEV = torch.device("cuda:0")
# (name, rows, cols, side, count): side "in" = W Q over the input features, "out" = Q^T W
SHAPES = [("embed_tokens / lm_head", 152064, 3584, "in", 2),
          ("llm q_proj", 3584, 3584, "in", 28), ("llm k/v_proj", 512, 3584, "in", 56),
          ("llm o_proj", 3584, 3584, "out", 28), ("llm gate/up_proj", 18944, 3584, "in", 56),
          ("llm down_proj", 3584, 18944, "out", 28),
          ("vis qkv", 3840, 1280, "in", 32), ("vis proj", 1280, 1280, "out", 32),
          ("vis fc1", 5120, 1280, "in", 32), ("vis fc2", 1280, 5120, "out", 32)]


def timed(fn, reps):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


def main():
    torch.manual_seed(0)
    qs = {}
    total_k = total_d = 0.0
    print(f"{'weight':24s} {'shape':>14s} side   kernel ms   dense fp64 ms   speed-up   kernel GB/s")
    for name, rows, cols, side, count in SHAPES:
        n = cols if side == "in" else rows
        if n not in qs:
            qs[n] = ru.get_orthogonal_matrix(n, "hadamard", device=DEV)
        Q = qs[n]
        dense = Q.clone()
        W = (torch.randn(rows, cols, device=DEV) * 0.02).to(torch.bfloat16)
        big = rows * cols > 2e8
        if side == "in":
            tk = timed(lambda: ru.mul_q(W, Q), 3 if big else 10)
            td = timed(lambda: ru.mul_q(W, dense), 1 if big else 3)
        else:
            tk = timed(lambda: ru.mul_qt(Q, W), 10)
            td = timed(lambda: ru.mul_qt(dense, W), 3)
        total_k += tk * count
        total_d += td * count
        print(f"{name:24s} {rows:>7d}x{cols:<6d} {side:4s} {tk * 1e3:10.3f} {td * 1e3:15.2f} {td / tk:10.1f} "
              f"{2 * 2 * rows * cols / tk / 1e9:12.0f}")
    print(f"whole model (counts applied): kernel {total_k:.3f} s, dense fp64 {total_d:.3f} s")


if __name__ == "__main__":
    main()
