"""mq_gptq_block (the GPTQ column loop in one launch) against the oracle restatement of that loop, and the
solver end to end on the GPU against the reference goldens."""
import os

import numpy as np
import pytest
import torch

from golden_inputs import make_w, make_x
from test_gptq_cpu import build

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
torch.set_grad_enabled(False)


@pytest.mark.parametrize("N,cols,bits", [(200, 128, 4), (64, 100, 4), (1, 1, 4), (333, 37, 8), (4096, 128, 4),
                                         (65, 128, 2), (130, 64, 6)])
def test_block_kernel_equals_the_oracle(N, cols, bits):
    """mq_gptq_block against oracle.gptq_block (orc_gptq_block, pinned on the CPU to the reference's
    GPTQ goldens in test_gptq_cpu.py): Q and Err bit for bit."""
    import oracle
    from mquant_amd import ops
    columns = cols + 40
    W = torch.from_numpy(make_w(N + cols, (N, columns))).to(DEV) * 3.0
    X = torch.from_numpy(make_x(N, (columns + 64, columns))).to(DEV)
    H = X.T @ X / X.shape[0]
    H += 0.01 * torch.mean(torch.diag(H)) * torch.eye(columns, device=DEV)
    Hinv = torch.linalg.cholesky(torch.cholesky_inverse(torch.linalg.cholesky(H)), upper=True).contiguous()
    scale = (W.abs().amax(1) / (2 ** (bits - 1) - 1)).contiguous()
    i1, i2 = 8, 8 + cols
    Q = torch.zeros_like(W)
    E = torch.empty((N, 128), device=DEV)
    ops.gptq_block(W, i1, i2, Hinv, scale, bits, Q, E)
    q_ref, e_ref = oracle.gptq_block(W[:, i1:i2].cpu().numpy(), Hinv[i1:i2, i1:i2].cpu().numpy(),
                                     scale.cpu().numpy(), bits)
    np.testing.assert_array_equal(Q[:, i1:i2].cpu().numpy(), q_ref)
    np.testing.assert_array_equal(E[:, :cols].cpu().numpy(), e_ref)
    assert float(Q[:, :i1].abs().max()) == 0.0 and float(Q[:, i2:].abs().max()) == 0.0


@pytest.mark.parametrize("case", ["plain", "actorder", "mse_w8", "wide"])
def test_solver_on_gpu_fused_equals_unfused_and_stays_near_the_reference(golden_dir, case):
    from fake_quant import quant_utils as qu
    from fake_quant.gptq.gptq_utils import GPTQ
    g = np.load(os.path.join(golden_dir, f"gptq_{case}.npz"))
    seed, n_out, n_in, bits, mse, actorder, groupsize = [int(v) for v in g["meta"]]
    outs = {}
    for fused in (True, False):
        layer, xs = build("linear", n_out, n_in, seed)
        layer = layer.to(DEV)
        solver = GPTQ(layer)
        solver.use_kernel = fused
        solver.quantizer = qu.WeightQuantizer()
        solver.quantizer.configure(bits, perchannel=True, sym=True, mse=bool(mse))
        for x in xs:
            solver.add_batch(x.to(DEV), None)
        solver.fasterquant(percdamp=0.01, groupsize=groupsize, actorder=bool(actorder), static_groups=False)
        outs[fused] = layer.weight.data.clone()
    torch.testing.assert_close(outs[True], outs[False], rtol=0, atol=0)
    # vs the reference's CPU run: Hessian / Cholesky / trailing GEMM orders differ between the CPU
    # and the GPU BLAS, so a few weights land on a neighbouring level; the bulk must agree
    same = float((outs[True].cpu().numpy() == g["Q"]).mean())
    assert same > 0.97, same


def test_large_layer_runs_and_reduces_error():
    from fake_quant import quant_utils as qu
    from fake_quant.gptq.gptq_utils import GPTQ
    lin = torch.nn.Linear(1024, 768, bias=False).to(DEV)
    lin.weight.data = torch.from_numpy(make_w(1, (768, 1024))).to(DEV) * 2.0
    w0 = lin.weight.data.clone()
    x = torch.from_numpy(make_x(2, (2048, 1024))).to(DEV)
    solver = GPTQ(lin)
    solver.quantizer = qu.WeightQuantizer()
    solver.quantizer.configure(4, perchannel=True, sym=True, mse=False)
    solver.add_batch(x, None)
    solver.fasterquant()
    rtn = qu.WeightQuantizer()
    rtn.configure(4, perchannel=True, sym=True, mse=False)
    rtn.find_params(w0)
    e_gptq = float(((x @ (lin.weight.data - w0).T) ** 2).mean())
    e_rtn = float(((x @ (rtn.quantize(w0) - w0).T) ** 2).mean())
    assert e_gptq < 0.8 * e_rtn, (e_gptq, e_rtn)
