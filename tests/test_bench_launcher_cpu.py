"""bench.py --gpus N starts N ranks itself (fresh children, before anything touches a GPU) and the
ranks gather per-sample logits; exercised on CPU with the gloo self-test mode."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _run(args, env=None):
    e = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        e.pop(k, None)
    e.update(env or {})
    return subprocess.run([sys.executable, BENCH] + args, env=e, capture_output=True, text=True, timeout=300)


def test_gpus_flag_spawns_that_many_ranks():
    r = _run(["--gpus", "2", "--cpu-selftest"])
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["gathered_ok"] is True


def test_single_rank_needs_no_launcher():
    r = _run(["--cpu-selftest"])
    assert r.returncode == 0, r.stderr[-2000:]
    assert json.loads(r.stdout.strip().splitlines()[-1])["n_gpus"] == 1


def test_mismatch_between_gpus_and_world_size_fails_loudly():
    r = _run(["--gpus", "2", "--cpu-selftest"], env={"WORLD_SIZE": "4", "RANK": "0", "LOCAL_RANK": "0", "MASTER_PORT": "29999"})
    assert r.returncode != 0 and "WORLD_SIZE=4" in (r.stderr + r.stdout)


def test_a_stray_world_size_without_the_rendezvous_environment_runs_as_a_single_process():
    """A scheduler that exports WORLD_SIZE (but no RANK / LOCAL_RANK / MASTER_PORT) must not push a plain `python bench.py` into
    init_process_group(env://): the process is not a torch.distributed.run rank and runs alone."""
    import json
    r = _run(["--cpu-selftest"], env={"WORLD_SIZE": "4"})
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["n_gpus"] == 1 and line["gathered_ok"] is True


def test_launcher_runs_before_any_gpu_import():
    """The spawn decision sits above every torch import of main(): the parent must never initialise HIP."""
    src = open(BENCH).read()
    main = src[src.index("def main():"):]
    assert main.index("subprocess.call(cmd") < main.index("import torch")


def test_secondary_block_respects_its_time_budget():
    """The bounded secondary block of the default bench line (the other BASELINE configurations as child processes): with no
    time left every entry is reported as skipped and nothing is started."""
    import time
    sys.path.insert(0, ROOT)
    import bench
    out = bench.run_secondary(0.0, time.perf_counter())
    assert set(out) == set(bench.SECONDARY) and all("skipped" in v for v in out.values())
    assert {"qwenvl_7b", "internvl2_8b_batch4", "qwen2vl_72b_kv_fp8", "qwen2vl_7b_visual_w8"} <= set(out)
