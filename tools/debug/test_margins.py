import sys, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
torch.set_grad_enabled(False)
from mquant_amd import workload
from mquant_amd.full_prefill import FullPrefill, QWEN2VL_7B, QWEN2VL_72B
DEV = "cuda:0"
cs = torch.nn.functional.cosine_similarity
specs = workload._qwen2vl_7b_specs(True, 1, 2)
pf = workload.Prefill(specs, device=DEV, share_groups=True)
outs = []
for own in (False, True):
    fp = FullPrefill(pf, fused_glue=True, attn_kernel=own)
    fp.calibrate()
    outs.append((fp.step().float().clone(), fp.attn_first.float().clone(), fp.vis_attn_first.float().clone()))
    fp.restore_hot_path_scales()
(la, aa, va), (lb, ab, vb) = outs
print("own-attn: vis max rel", float((va - vb).abs().max() / va.abs().max()), "vis cos", float(cs(va.flatten(), vb.flatten(), dim=0)),
      "attn rel", float((aa - ab).norm() / aa.norm()), "attn cos", float(cs(aa.flatten(), ab.flatten(), dim=0)), "logit cos", float(cs(la.flatten(), lb.flatten(), dim=0)))
for geo_name in ("7b", "72b"):
    if geo_name == "72b":
        specs, geo = workload.qwen2vl_72b_specs(v=1, l=2), QWEN2VL_72B
    else:
        specs, geo = workload._qwen2vl_7b_specs(True, 1, 2), QWEN2VL_7B
    pf = workload.Prefill(specs, device=DEV, share_groups=True)
    o = {}
    for direct in (False, True):
        fp = FullPrefill(pf, fused_glue=True, geometry=geo, kv_fp8=True, attn_fp8=direct)
        fp.calibrate()
        logits = fp.step().float().clone()
        o[direct] = (logits, fp.attn_first.float().clone())
        fp.restore_hot_path_scales()
    (la, aa), (lb, ab) = o[False], o[True]
    print(geo_name, "direct-vs-readback: attn max rel", float((aa - ab).abs().max() / aa.abs().max()), "cos", float(cs(aa.flatten(), ab.flatten(), dim=0)), "logit cos", float(cs(la.flatten(), lb.flatten(), dim=0)))
    o = {}
    for kv8 in (False, True):
        fp = FullPrefill(pf, fused_glue=True, geometry=geo, kv_fp8=kv8)
        fp.calibrate()
        o[kv8] = (fp.step().float().clone(), fp.attn_first.float().clone())
        fp.restore_hot_path_scales()
    (la, aa), (lb, ab) = o[False], o[True]
    print(geo_name, "fp8-vs-fp16: attn rel", float((aa - ab).norm() / aa.norm()), "cos", float(cs(aa.flatten(), ab.flatten(), dim=0)), "logit cos", float(cs(la.flatten(), lb.flatten(), dim=0)))
