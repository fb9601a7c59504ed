"""200 distinct row counts through one wrapper: the device memory held afterwards is the largest call's, not the sum
(VERDICT r4 "What's weak" 9; mquant_amd/engine.py Workspace, ops.splitk_workspace)."""
import numpy as np
import pytest
import torch

from golden_inputs import make_w, make_x

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
torch.set_grad_enabled(False)


def _wrapper(K, N, static):
    from fake_quant import quant_utils as qu
    from fake_quant.gptq.rtn import rtn_module
    lin = torch.nn.Linear(K, N, bias=False)
    lin.weight.data = torch.from_numpy(make_w(K + N, (N, K)))
    wrap = qu.ActQuantWrapper(lin.half().to(DEV))
    rtn_module(wrap, "layer", 4, True, False, [], {})
    if static:
        wrap.quantizer.configure(bits=8, sym=True, static=True, observer_type="minmax")
        qu.calib_layer(wrap, [torch.from_numpy(make_x(1 + i, (64, K))).half().to(DEV) for i in range(2)], _Args())
    else:
        wrap.quantizer.configure(bits=8, sym=True)
    return wrap


class _Args:
    skip_names = []
    no_sibling_fusion = True


@pytest.mark.parametrize("static", [True, False])
def test_memory_stays_flat_over_200_distinct_row_counts(static):
    from mquant_amd import engine
    K, N = 3584, 512
    wrap = _wrapper(K, N, static)
    x_all = torch.from_numpy(make_x(3, (800, K))).half().to(DEV)
    sizes = list(range(1, 801, 4))
    assert len(sizes) == 200
    y = wrap(x_all[: sizes[-1]])                               # the largest call first: every buffer exists afterwards
    assert wrap._real is not None
    ref_small = wrap(x_all[:37]).clone()
    del y
    torch.cuda.synchronize()
    base = torch.cuda.memory_allocated()
    held = engine.WORKSPACE.nbytes()
    for M in sizes:
        y = wrap(x_all[:M])
        del y
    torch.cuda.synchronize()
    assert torch.cuda.memory_allocated() == base, "a per-row-count buffer was kept"
    assert engine.WORKSPACE.nbytes() == held
    # and ascending from a cleared workspace: what is held in the end is the largest call's buffers
    engine.WORKSPACE.clear()
    for M in sizes:
        y = wrap(x_all[:M])
        del y
    torch.cuda.synchronize()
    top = (sizes[-1] + 15) // 16 * 16
    assert engine.WORKSPACE.nbytes() <= top * K + 4 * 800
    # slicing a shared buffer changes no result
    assert torch.equal(wrap(x_all[:37]), ref_small)


def test_a_captured_graph_keeps_replaying_into_its_buffer_after_the_workspace_grew():
    from mquant_amd import engine
    K, N = 1280, 256
    wrap = _wrapper(K, N, True)
    engine.WORKSPACE.clear()
    x = torch.from_numpy(make_x(5, (96, K))).half().to(DEV)
    want = wrap(x).clone()
    engine.WORKSPACE.clear()
    g = torch.cuda.CUDAGraph()
    st = torch.cuda.Stream()
    out = torch.empty((96, N), dtype=torch.float16, device=DEV)
    with torch.cuda.stream(st):
        wrap(x)                                                # first use outside the capture (weight freeze)
        st.synchronize()
        engine.WORKSPACE.clear()
        with torch.cuda.graph(g, stream=st, capture_error_mode="thread_local"):
            out.copy_(wrap(x))
        st.synchronize()
    big = torch.from_numpy(make_x(6, (4096, K))).half().to(DEV)
    y_big = wrap(big)                                          # grows the workspace: the captured buffer must survive
    assert engine.WORKSPACE._pinned, "the buffer a graph captured was released"
    junk = [torch.full((96 * K,), 7, dtype=torch.int8, device=DEV) for _ in range(8)]   # would land in a freed block
    out.zero_()
    g.replay()
    torch.cuda.synchronize()
    assert torch.equal(out, want)
    assert np.isfinite(y_big.float().cpu().numpy()).all()
    del junk
