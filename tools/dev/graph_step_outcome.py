#!/usr/bin/env python3
"""How many steps of the reordered tiny-OPT fixture are replayed from the HIP graph (and what the loop noted)."""
import os
import sys

import torch

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in ("llm-quantization_amd", "tests", ""):
    sys.path.insert(0, os.path.join(REPO, p))
from helpers import golden  # noqa: E402
import test_gpu_block_calibration as opt  # noqa: E402
from miq import reorder_utils as R  # noqa: E402
from miq.calibrate import CalibConfig, calibrate_layer, reorder_block  # noqa: E402

DEV = "cuda:0"
d = golden("block")["opt_tiny_w4a4_reorder"]
m = d["meta"]
prefix = "model.decoder.layers.0"
sites = ("self_attn.q_proj", "self_attn.out_proj", "fc1")
act_scales = {f"{prefix}.{n}": d[f"act_scale_{n}"] for n in sites}
act_shifts = {f"{prefix}.{n}": d[f"act_shift_{n}"] for n in sites}
q = opt.build(d)
R.oc_maxmin_dict.clear()
R.ic_maxmin_dict.clear()
for name in ("self_attn_layer_norm", "self_attn.q_proj", "self_attn.k_proj", "final_layer_norm"):
    R.oc_maxmin_dict[name] = (d[f"stat_oc_{name}_max"].to(DEV), d[f"stat_oc_{name}_min"].to(DEV))
for name in ("self_attn.out_proj", "fc2"):
    R.ic_maxmin_dict[name] = (d[f"stat_ic_{name}_max"].to(DEV), d[f"stat_ic_{name}_min"].to(DEV))
indices = reorder_block(q, m["H"], m["F"], m["heads"], {f"R{k}": True for k in range(1, 6)}, m["clusters"], DEV)
cfg = CalibConfig(epochs=m["epochs"], nsamples=m["N"], batch_size=1, let=True, lwc=True, deactive_amp=True,
                  autocast_passes=False, reorder="012345", n_clusters=m["clusters"], graph_step=True)
quant, fp = d["inps"].clone().to(DEV), d["inps"].clone().to(DEV)
res = calibrate_layer(q, quant, fp, d["mask"].to(DEV), cfg, act_scales, act_shifts, prefix,
                      dims=(m["H"], m["F"], m["heads"]), indices=indices)
print("epochs", m["epochs"], "N", m["N"], "steps", res["steps"], "graph_steps", res["graph_steps"], "notes", res["notes"])
