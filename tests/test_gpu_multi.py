"""Two-GPU checks (skipped on a one-GPU box): batch sharding through bench.py's own launcher with the
gathered logits compared to the single-GPU result, and kernels launched on a device that is not the
current one (per-device kernel attributes, device guard of mquant_amd.ops)."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
two_gpus = pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs")


@two_gpus
def test_bench_two_ranks_gather_the_single_gpu_logits():
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
                        "--no-cpu-baseline", "--no-full-prefill"], env=env, capture_output=True, text=True, timeout=1800)
    assert r.returncode == 0, r.stderr[-3000:]
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["n_gpus"] == 2
    assert line["logits_check"]["samples"] == 2 and line["logits_check"]["equal_to_single_gpu"] is True


@two_gpus
def test_kernels_follow_the_tensor_device_not_the_current_one(had_table):
    """One 256 x 256 GEMM (> 64 KiB of dynamic LDS) and one K = 156 Hadamard on cuda:1 after cuda:0, with
    cuda:0 current: the reference's device_map='auto' placement."""
    import oracle
    from mquant_amd import ops
    rng = np.random.default_rng(0)
    M, N, K = 512, 1024, 256
    a = rng.integers(-128, 128, size=(M, K), dtype=np.int8)
    w = rng.integers(-8, 8, size=(N, K), dtype=np.int8)
    ref = oracle.gemm_i32(a, w)
    bits = had_table["words"][156]
    x = rng.normal(size=(4, 19968)).astype(np.float32)
    outs = []
    torch.cuda.set_device(0)
    try:
        for dev in ("cuda:0", "cuda:1"):
            at, wt = torch.from_numpy(a).to(dev), torch.from_numpy(w).to(dev)
            ops.splitk_workspace(torch.device(dev), 64 << 20)
            ops.gemm_debug_force(3, 1)
            acc = ops.gemm_w4a8_i32(ops.TiledAct.from_rows(at), ops.prepack(wt, 4), 4, N)
            np.testing.assert_array_equal(acc.cpu().numpy(), ref, err_msg=dev)
            ops.gemm_debug_force(-1, 0)
            xt = torch.from_numpy(x).to(dev).half()
            q, _ = ops.hadamard_quant_i8(xt, 19968, 156, torch.from_numpy(bits).to(dev), 0.05)
            outs.append(q.cpu())
        assert torch.equal(outs[0], outs[1])
        with pytest.raises(Exception):
            ops.gemm_w4a8_i32(torch.from_numpy(a).to("cuda:0"), ops.prepack(torch.from_numpy(w).to("cuda:1"), 4), 4, N)
    finally:
        ops.gemm_debug_force(-1, 0)


def test_bench_as_a_single_torchrun_rank_initialises_rccl_and_exchanges_logits_on_this_gpu():
    """The multi-GPU code path on a one-GPU box: bench.py started the way the driver starts its ranks
    (torch.distributed.run, here with one process).  The rank initialises the RCCL process group on its
    device, every timed step ends in the fp16 lm_head + all_gather on the side stream, the whole-prefill
    logits are gathered with shard.gather_logits and compared with the single-GPU computation."""
    import socket
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=1", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "1",
           "--no-cpu-baseline", "--no-full-prefill"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=1800)
    assert r.returncode == 0, r.stderr[-3000:]
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["n_gpus"] == 1 and line["value"] > 0
    assert "ActQuantWrapper.forward" in line["config"]["path"] and "FAILED" not in line["config"]["path"]
    chk = line["logits_check"]
    assert chk["step_exchange"]["own_row_intact"] is True and chk["step_exchange"]["finite"] is True
    assert chk["samples"] == 1 and chk["equal_to_single_gpu"] is True
    assert line["roofline"]["launches_per_step"] == 243


def test_bench_single_torchrun_rank_with_four_samples_per_gpu_on_internvl2():
    """BASELINE configuration 4's per-GPU share (batch 32 over 8 GPUs = 4 samples per rank, InternVL2-8B): every step ends in
    4 x 92 553 last-position logits per rank, the all_gather on the side stream exchanges [world x 4, vocab], and
    shard.gather_logits restores sample order for B_local > 1 -- executed on this box's one GPU under torch.distributed.run."""
    import socket
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=1", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1",
           "--batch", "4", "--workload", "internvl2_8b", "--no-cpu-baseline", "--no-full-prefill"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=1800)
    assert r.returncode == 0, r.stderr[-3000:]
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["n_gpus"] == 1 and line["value"] > 0
    chk = line["logits_check"]
    ex = chk["step_exchange"]
    assert ex["samples_per_rank"] == 4 and ex["all_gather_bytes"] == 4 * 92553 * 2
    assert ex["own_row_intact"] is True and ex["finite"] is True
    assert chk["gather_logits"] == {"samples": 4, "shape": [4, 92553], "own_samples_in_place": True}
