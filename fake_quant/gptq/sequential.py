"""Layer-sequential GPTQ (the capture / replay loop shared by every ``*_gptq_plus`` driver).

Upstream repeats this loop once per tower and model (reference
``fake_quant/gptq/qwen2vl_gptq_plus.py:41-246,268-360,409-556`` and the InternVL / Qwen-VL /
MiniCPM-V twins): (1) swap a module for a catcher, run the calibration prompts through
``model.generate`` and keep what reaches that module; (2) per block and per *sequential group* of
Linears: hook the Linears, replay the block on the captured inputs so every GPTQ object sees the
activations produced by the ALREADY quantized predecessors, solve, write the weights back;
(3) replay once more to get the next block's inputs.

Here the loop exists once.  A captured sample is the full ``(args, kwargs)`` of the call (upstream
keeps a hand-picked subset: attention_mask / position_ids / cache_position, cu_seqlens /
rotary_pos_emb, ...), so the same code serves every tower.
"""
from __future__ import annotations

from typing import Callable, Dict, List, Sequence, Tuple

import torch
import tqdm

from fake_quant import quant_utils, utils
from .gptq_utils import GPTQ

Sample = Tuple[tuple, dict]


class _Stop(Exception):
    pass


def run_calibration_prompts(model, dataset, dataset_name, args, stop_after: Callable[[], bool]) -> None:
    """Drive the VLMEvalKit-style wrapper over the dataset until ``stop_after()``."""
    for i in tqdm.tqdm(range(len(dataset.data))):
        if stop_after():
            break
        record = dataset.data.iloc[i]
        if hasattr(model, "use_custom_prompt") and model.use_custom_prompt(dataset_name):
            struct = model.build_prompt(record, dataset=dataset_name)
        else:
            struct = dataset.build_prompt(record)
        try:
            model.generate(message=struct, dataset=args.dataset_name)
        except _Stop:
            pass


def capture_inputs(target: torch.nn.Module, feed: Callable[[Callable[[], bool]], None], nsamples: int) -> List[Sample]:
    """Record the first ``nsamples`` calls of ``target`` (positional + keyword arguments) while
    ``feed`` pushes calibration data through the model; each call is aborted at ``target``."""
    samples: List[Sample] = []

    def catcher(*a, **kw):
        samples.append((a, kw))
        raise _Stop()

    target.forward = catcher                    # instance attribute shadows the class's forward
    try:
        feed(lambda: len(samples) >= nsamples)
    finally:
        del target.forward
    return samples[:nsamples]


def _first(out):
    return out[0] if isinstance(out, (tuple, list)) else out


def new_quantizer(bits: int, sym: bool, mse: bool):
    q = quant_utils.WeightQuantizer()
    q.configure(bits, perchannel=True, sym=sym, mse=mse)
    return q


def gptq_group(block_call: Callable[[Sample], object], samples: Sequence[Sample], subset: Dict[str, torch.nn.Module],
               bits: int, sym: bool, mse: bool, args, key: Callable[[str], str], quantizers: dict) -> None:
    """One sequential group: accumulate Hessians over a replay, then solve every member."""
    solvers = {}
    for name, lin in subset.items():
        print(f"{name}", end="  ", flush=True)
        solvers[name] = GPTQ(lin)
        solvers[name].quantizer = new_quantizer(bits, sym, mse)
    handles = [lin.register_forward_hook(
        (lambda nm: lambda _m, inp, out: solvers[nm].add_batch(inp[0].data, out.data))(name))
        for name, lin in subset.items()]
    try:
        for s in samples:
            block_call(s)
    finally:
        for h in handles:
            h.remove()
    for name, solver in solvers.items():
        solver.fasterquant(percdamp=args.percdamp, groupsize=args.w_groupsize, actorder=args.act_order,
                           static_groups=False)
        quantizers[key(name)] = solver.quantizer
        # with --w_groupsize > 0 fasterquant re-runs find_params per column group: ``scale`` holds the LAST group's as in the
        # reference, every group's is kept in ``group_scales`` (gptq_utils.py here) and the wrapper runs
        # mq_gemm_w4a8_wgroupscale.  With --act_order on top the groups are runs of PERMUTED columns: the solver keeps the
        # permutation next to the scales (``group_perm``), the wrapper's engine gathers the activation columns the same way
        # (engine.W4A8Linear.col_perm) and stays on the integer path (ActQuantWrapper.extra_repr says which backend runs).
        _attach(subset[name], solver.quantizer)
        solver.free()


def _attach(lin, quantizer) -> None:
    """Tell the owning ActQuantWrapper which quantizer produced these weights (real-integer path)."""
    owner = getattr(lin, "_mq_owner", None)
    if owner is not None:
        wrapper, sub = owner
        quant_utils.attach_weight_quantizer(wrapper, sub, quantizer)


def mark_owners(root: torch.nn.Module) -> None:
    """Remember, on every wrapped Linear, the wrapper (and sub-module name) it belongs to."""
    for wrapper in quant_utils.find_qlayers(root, layers=[quant_utils.ActQuantWrapper]).values():
        for sub in ("module", "L2"):
            if hasattr(wrapper, sub):
                object.__setattr__(getattr(wrapper, sub), "_mq_owner", (wrapper, sub))


def gptq_blocks(blocks: Sequence[torch.nn.Module], samples: List[Sample], sequential: Sequence[Sequence[str]],
                bits: int, sym: bool, mse: bool, args, key_fmt: str, quantizers: dict) -> List[Sample]:
    """Quantize a stack of blocks one after the other.  ``key_fmt % (i, name)`` is the dict key.
    Returns the samples that would enter the block after the last one."""
    for i, block in enumerate(blocks):
        print(f"\nLayer {i}:", flush=True, end=" ")
        mark_owners(block)
        full = quant_utils.find_qlayers(block, layers=[torch.nn.Linear])
        call = lambda s, blk=block: blk(*s[0], **s[1])          # noqa: E731
        for names in sequential:
            if any(p in n for n in names for p in args.skip_names):
                continue
            gptq_group(call, samples, {n: full[n] for n in names}, bits, sym, mse, args,
                       lambda n, i=i: key_fmt % (i, n), quantizers)
        samples = [((_first(call(s)),) + tuple(s[0][1:]), s[1]) for s in samples]
        if torch.cuda.is_available():
            torch.cuda.empty_cache()
    return samples


def gptq_single(module_call: Callable[[Sample], object], root: torch.nn.Module, samples: Sequence[Sample],
                sequential: Sequence[Sequence[str]], bits: int, sym: bool, mse: bool, args,
                key: Callable[[str], str], quantizers: dict, layers=(torch.nn.Linear,)) -> None:
    """Same for one module whose Linears are solved group after group (merger / mlp1 / patch conv)."""
    mark_owners(root)
    full = quant_utils.find_qlayers(root, layers=list(layers))
    for names in sequential:
        gptq_group(module_call, samples, {n: full[n] for n in names}, bits, sym, mse, args, key, quantizers)
    utils.cleanup_memory(verbos=False)
