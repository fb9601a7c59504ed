import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "tests"))
import numpy as np, torch
torch.set_grad_enabled(False)
from mquant_amd import workload
from fake_quant import quant_utils as qu
DEV = "cuda:0"
specs = workload.tiny_specs()
pf = workload.Prefill(specs, device=DEV, share_groups=True)
wp = workload.WrapperPrefill(specs, device=DEV, fuse_siblings=True)
ys = wp.outputs()
keys = workload.execution_order(specs)
got = {k: y.float().cpu().numpy() for k, y in zip(keys, ys)}
# direct engine outputs
direct = {}
for L in pf.layers:
    a, x0 = L.lin.quantize(L.x, L.row_sel)
    y = L.lin.gemm(a, x0, pf.dtype, L.row_sel, None)
    direct[(getattr(L, "order_name", L.spec.name), L.idx)] = y.float().cpu().numpy()
for (wrap, x, spec), key in zip(wp.calls, keys):
    eng = qu.real_engine(wrap)
    d = direct.get(key)
    line = f"{key} wrapper s0={eng.s_x0!r} s1={eng.s_x1!r} split={eng.split} had={eng.had is not None} "
    for L in pf.layers:
        if (getattr(L, "order_name", L.spec.name), L.idx) == key:
            line += f"| direct s0={L.lin.s_x0!r} s1={L.lin.s_x1!r} "
            n = min(eng.s_w.numel(), L.lin.s_w.numel())
            line += f"s_w equal={bool(torch.equal(eng.s_w[:n], L.lin.s_w[:n]))} "
            if eng.bias is not None:
                line += f"bias maxdiff={float((eng.bias[:n]-L.lin.bias[:n]).abs().max()):.2e} "
            if eng.w0 is not None:
                line += f"w0 equal={bool(torch.equal(eng.w0, L.lin.w0))} "
            line += f"w_img equal={bool(torch.equal(eng.w_img, L.lin.w_img)) if eng.w_img.shape == L.lin.w_img.shape else 'shape'} "
    if d is not None:
        g = got[key]
        line += f"| out maxdiff={np.abs(g - d[:, :g.shape[1]]).max():.3e}"
    print(line)
