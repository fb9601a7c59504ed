cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
HAD_ROWS=1,4,16,64,256 HAD_SHAPES=vis.fc2,llm.down timeout 300 python3 tools/had_bench.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r5_hadamard_small_m.txt
timeout 300 python3 - <<'PY' 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r5_hadamard_small_m.txt
import torch, sys
sys.path.insert(0, '.')
from mquant_amd import ops
dev = torch.device('cuda:0')
def bench(fn, iters=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
for M in (1, 16):
    for K in (3584,):
        x = torch.randn((M, K), device=dev, dtype=torch.float16)
        out = ops.TiledAct.empty(M, K, dev)
        print(f"static quantizer M={M} K={K}: {bench(lambda: ops.quantize_act_i8(x, 0.05, out=out)):.1f} us")
PY
