"""Round-to-nearest weight pass shared by the per-model drivers."""
import torch

from fake_quant import quant_utils


def _owner_wrapper(root, dotted):
    """The ActQuantWrapper that owns sub-module ``dotted`` ('...wrapper.module' / '...L2')."""
    parent_path, _, leaf = dotted.rpartition(".")
    mod = root
    for part in [p for p in parent_path.split(".") if p]:
        mod = getattr(mod, part) if not part.isdigit() else mod[int(part)]
    return (mod, leaf) if isinstance(mod, quant_utils.ActQuantWrapper) else (None, leaf)


def rtn_module(root, key_prefix, bits, sym, mse, skip_names, quantizers, groupsize=-1):
    """RTN over every nn.Linear under ``root`` (exact type; wrapper sub-modules ``module`` /
    ``L2``; ``L1`` -- the unquantized split column -- is skipped like upstream).

    ``groupsize`` > 0 (an extension: the reference has group-wise scales in its GPTQ solver only,
    gptq/gptq_utils.py:263-273): ``find_params`` + ``quantize`` on every group of ``groupsize`` consecutive
    input channels, and the quantizer left behind carries what this repository's GPTQ records
    (``group_scales`` / ``group_zeros`` [rows, groups], ``groupsize``) -- weights of the form a
    ``--w_groupsize`` GPTQ run produces, without its error feedback; synthetic benchmarks use it."""
    subset = quant_utils.find_qlayers(root, layers=[torch.nn.Linear])
    for name, lin in subset.items():
        if any(p in name for p in skip_names) or "L1" in name:
            continue
        qz = quant_utils.WeightQuantizer()
        qz.configure(bits, perchannel=True, sym=sym, mse=mse)
        W = lin.weight.data
        if groupsize > 0 and W.shape[1] % groupsize == 0:
            Q = torch.empty_like(W)
            scales, zeros = [], []
            for i in range(0, W.shape[1], groupsize):
                qz.find_params(W[:, i:i + groupsize])
                scales.append(qz.scale.reshape(-1).float().clone())
                zeros.append(qz.zero.reshape(-1).float().clone())
                Q[:, i:i + groupsize] = qz.quantize(W[:, i:i + groupsize]).to(W.dtype)
            qz.groupsize = int(groupsize)
            qz.group_permuted = False
            qz.group_scales = torch.stack(scales, dim=1)
            qz.group_zeros = torch.stack(zeros, dim=1)
            lin.weight.data = Q
        else:
            qz.find_params(W)
            lin.weight.data = qz.quantize(W).to(W.dtype)
        owner, leaf = _owner_wrapper(root, name)
        if owner is not None:
            quant_utils.attach_weight_quantizer(owner, leaf, qz)
        quantizers[f"{key_prefix}.{name}" if name else key_prefix] = qz.cpu()
    if torch.cuda.is_available():
        torch.cuda.empty_cache()


def rtn_wrapped_conv(wrapper, key, bits, sym, mse, quantizers):
    """RTN of a wrapped patch-embedding convolution (per output channel over the flattened
    kernel)."""
    qz = quant_utils.WeightQuantizer()
    qz.configure(bits, perchannel=True, sym=sym, mse=mse)
    W = wrapper.module.weight.data
    qz.find_params(W)
    wrapper.module.weight.data = qz.quantize(W).to(W.dtype)
    quant_utils.attach_weight_quantizer(wrapper, "module", qz)
    quantizers[key] = qz.cpu()
