import os, sys
import torch
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(REPO, "llm-quantization_amd")); sys.path.insert(0, os.path.join(REPO, "tests")); sys.path.insert(0, REPO)
from helpers import golden
from test_gpu_block_llama import build, config
from miq.calibrate import calibrate_layer
DEV = "cuda:0"; PREFIX = "model.layers.0"; LET_SITES = ("self_attn.q_proj", "self_attn.o_proj", "mlp.up_proj")
d = golden("block_llama")["llama_tiny_w4a4_pertensor"]; m = d["meta"]
act_scales = {f"{PREFIX}.{n}": d[f"act_scale_{n}"] for n in LET_SITES}
out = {}
for tag, graph in (("eager", False), ("eager2", False), ("graph", True)):
    q = build(d); cfg = config(m); cfg.deactive_amp = False; cfg.graph_step = graph; cfg.epochs = 3
    quant, fp = d["inps"].clone().to(DEV), d["inps"].clone().to(DEV)
    res = calibrate_layer(q, quant, fp, d["mask"].to(DEV), cfg, act_scales, {}, PREFIX, is_llama=True)
    out[tag] = res["losses"]
for i in range(len(out["eager"])):
    print(i, out["eager"][i], out["eager2"][i], out["graph"][i], out["eager"][i] == out["eager2"][i], out["eager"][i] == out["graph"][i])
