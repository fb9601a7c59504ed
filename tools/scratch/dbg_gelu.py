import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import torch
from test_gpu_act_gemm import _case, _torch_act
torch.set_grad_enabled(False)
for tile in (14, 46, 48, 45):
    act = 2
    M, N, K = 300, 448, 384
    ops, levels, s_w, b, a, img, sel, s0, s1 = _case(M, N, K, seed=tile + 3 * act)
    y = ops.gemm_w4a8(a, img, 4, N, s0, s_w, s_x1=s1, row_sel=sel, bias=b, out_dtype=torch.float16)
    want = _torch_act(y, act, ops)
    ops.gemm_debug_force(tile, 0)
    got = ops.gemm_w4a8_act(a, img, 4, N, s0, s_w, act, s_x1=s1, row_sel=sel, bias=b, out_dtype=torch.float16)
    ops.gemm_debug_force(-1, 0)
    bad = (got != want).nonzero()
    print("tile", tile, "mismatches", bad.shape[0])
    for (r, c) in bad[:8].tolist():
        x = y[r, c]
        z = (1.702 * y)[r, c]
        sg = torch.sigmoid(1.702 * y)[r, c]
        print("  ", r, c, "x", float(x), hex(x.view(torch.int16).item() & 0xffff), "z", float(z), "sig", float(sg), "want", float(want[r, c]), "got", float(got[r, c]))
    # the same through the prologue kernel's arithmetic on the stored y
    q = torch.empty_like(y)
