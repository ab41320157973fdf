import sys
sys.path.insert(0,'/root/repo/tests'); sys.path.insert(0,'/root/repo/llm-quantization_amd'); sys.path.insert(0,'/root/repo')
import torch
from helpers import golden, rel_err
from miq import reorder_utils as R
from miq.calibrate import CalibConfig, calibrate_layer, reorder_block
import test_gpu_block_calibration as T
DEV="cuda:0"
d = golden("block")["opt_tiny_w4a4_reorder"]; m=d["meta"]
q=T.build(d)
for name in ("self_attn_layer_norm", "self_attn.q_proj", "self_attn.k_proj", "final_layer_norm"):
    R.oc_maxmin_dict[name] = (d[f"stat_oc_{name}_max"].to(DEV), d[f"stat_oc_{name}_min"].to(DEV))
for name in ("self_attn.out_proj", "fc2"):
    R.ic_maxmin_dict[name] = (d[f"stat_ic_{name}_max"].to(DEV), d[f"stat_ic_{name}_min"].to(DEV))
indices = reorder_block(q, m["H"], m["F"], m["heads"], {f"R{k}": True for k in range(1, 6)}, m["clusters"], DEV)
cfg = CalibConfig(epochs=m["epochs"], nsamples=m["N"], batch_size=1, let=True, lwc=True, deactive_amp=True, reorder="012345", n_clusters=m["clusters"])
prefix = "model.decoder.layers.0"
act_scales = {f"{prefix}.{n}": d[f"act_scale_{n}"] for n in ("self_attn.q_proj", "self_attn.out_proj", "fc1")}
act_shifts = {f"{prefix}.{n}": d[f"act_shift_{n}"] for n in ("self_attn.q_proj", "self_attn.out_proj", "fc1")}
qi = d["inps"].clone().to(DEV); fi = d["inps"].clone().to(DEV)
res = calibrate_layer(q, qi, fi, d["mask"].to(DEV), cfg, act_scales, act_shifts, prefix, dims=(m["H"], m["F"], m["heads"]), indices=indices)
print("losses", res["losses"]); print("ref   ", d["losses"].tolist())
for j in range(m["N"]):
    a=qi[j].cpu(); b=d["quant_out"][j]
    print("sample",j,"mean rel",((a-b).abs().mean()/b.abs().mean()).item(),"max",(a-b).abs().max().item())
for name, mod in q.named_modules():
    k=f"wq_{name}"
    if k in d:
        w=mod.weight.float().cpu(); r=d[k]
        print(name,"wq differing frac",(w!=r).float().mean().item(),"max",(w-r).abs().max().item(), "mean rel", ((w-r).abs().mean()/r.abs().mean()).item())
    k=f"actscale_{name}"
    if k in d:
        s=mod.act_quantizer.scale.float().cpu()
        print(name,"actscale rel",rel_err(s,d[k]))
