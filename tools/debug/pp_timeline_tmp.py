import sys, torch, collections
import numpy as np
sys.path.insert(0, '/root/repo')
from mquant_amd import ops
dev = torch.device('cuda:0')
M, N, K = 768, 37888, 3584
a = ops.TiledAct.from_rows(torch.randint(-128, 128, (M, K), dtype=torch.int8, device=dev))
q = torch.randint(-8, 8, (N, K), dtype=torch.int8, device=dev)
ws = [ops.prepack(q, 4) for _ in range(4)]
s_w = torch.full((N,), 0.01, device=dev)
out = torch.empty((M, N), dtype=torch.float16, device=dev)
wsb = ops.splitk_workspace(dev)
for i in range(6):
    ops.gemm_w4a8(a, ws[i % 4], 4, N, 0.02, s_w, out=out)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for i in range(20):
    ops.gemm_w4a8(a, ws[i % 4], 4, N, 0.02, s_w, out=out)
e1.record(); torch.cuda.synchronize()
print(f"gate|up {M}x{N}x{K}: {e0.elapsed_time(e1) / 20 * 1e3:.1f} us per launch")
st = wsb.view(torch.int32)[8192:8192 + 8 * 444].view(444, 8).cpu().numpy().astype('int64') & 0xffffffff
t0 = st[:, 0] + (st[:, 1] << 32)
hw, xcc = st[:, 5], st[:, 6] & 0xf
cu = ((hw >> 8) & 0xf) | (((hw >> 12) & 1) << 4) | (((hw >> 13) & 7) << 5) | (xcc << 8)
per = collections.defaultdict(list)
for r in range(444):
    per[int(cu[r])].append((int(t0[r]), int(st[r, 2]), int(st[r, 3]), int(st[r, 4]), int(st[r, 7])))
print("CUs:", len(per), "tiles per CU:", dict(collections.Counter(len(v) for v in per.values())))
gaps, pro, loop, epi, tot1, tot2 = [], [], [], [], [], []
shown = 0
for c, v in sorted(per.items()):
    v.sort()
    for (e, l0, l1, x, b) in v:
        pro.append(l0); loop.append(l1 - l0); epi.append(x - l1)
    if len(v) == 2:
        gaps.append(v[1][0] - (v[0][0] + v[0][3]))
        tot1.append(v[0][3]); tot2.append(v[1][3])
    if shown < 5 and len(v) == 2:
        shown += 1
        b = v[0][0]
        print(f"  CU {c}: " + "   ".join(f"[tile {bb}: entry {e - b}, first step landed +{l0}, loop done +{l1}, stores retired +{x}]" for (e, l0, l1, x, bb) in v))
f = lambda a: f"median {int(np.median(a))} min {int(np.min(a))} max {int(np.max(a))}"
print("entry -> first step landed:", f(pro)); print("loop:", f(loop)); print("epilogue (to stores retired):", f(epi))
print("first tile total:", f(tot1)); print("second tile total:", f(tot2)); print("gap between the end of a CU's first tile and the entry of its second:", f(gaps))
one = [v[0][3] for v in per.values() if len(v) == 1]
if one: print("CUs with one tile: total", f(one))
