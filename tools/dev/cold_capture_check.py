#!/usr/bin/env python3
"""Which first-use action is illegal inside a HIP-graph capture of the fresh-statistics activation quantizer?
Every scenario runs in its own child process (a failed capture can abort the process) and captures ONE call of
act_fresh_quant_fwd on a side stream with exactly one piece of per-process state left cold:

  warm        everything warmed up on the capture stream first (control)
  cold_layout the ClusterLayout is built inside the capture (torch.tensor(list, device=...): a pageable
              host-to-device copy)
  cold_table  the statistics table is allocated and zero-filled inside the capture (the library's guard is
              bypassed for this scenario)
  cold_lib    the very first call into libmiq.so happens inside the capture (code-object load)

Prints one line per scenario: ok / the Python exception / the child's exit status.  Stops at the first child that
dies from a signal (nothing more is started on the GPU after an abort)."""
import os
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

CHILD = r'''
import os, sys
sys.path.insert(0, os.path.join(%(repo)r, "llm-quantization_amd"))
import torch
scenario = %(scenario)r
dev = torch.device("cuda:0")
torch.cuda.set_device(0)
x = torch.randn(256, 1024, device=dev)
counts = [32] * 32
side = torch.cuda.Stream(device=dev)
side.wait_stream(torch.cuda.current_stream())
from miq import functional as F
if scenario == "cold_table":
    real = torch.cuda.is_current_stream_capturing
    import miq.functional as MF
    # bypass the guard: allocate inside the capture as the code did before the guard existed
    def ws(dev_, nbytes):
        key = (dev_.index, torch._C._cuda_getCurrentRawStream(dev_.index))
        t = MF._FRESH_WS.get(key)
        if t is None or t.numel() < nbytes:
            t = MF._FRESH_WS[key] = torch.zeros(max(nbytes, 1 << 16), dtype=torch.uint8, device=dev_)
        return t
    MF._fresh_workspace = ws
if scenario != "cold_lib":
    warm_layout = F.ClusterLayout.get([64] * 16, 1024, dev)   # a DIFFERENT layout: loads the library, warms nothing else
    with torch.cuda.stream(side):
        if scenario in ("warm", "cold_layout"):
            F.act_fresh_quant_fwd([x], 4, warm_layout)         # creates the stream's table
        else:
            F.colquant_fwd(x, torch.ones(1024, device=dev), torch.zeros(1024, device=dev), 4)  # library loaded, no table
    torch.cuda.synchronize()
# only cold_layout builds the layout (host lists -> device tensors) inside the capture
layout = None if scenario == "cold_layout" else F.ClusterLayout.get(counts, 1024, dev)
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
try:
    with torch.cuda.graph(g, stream=side):
        if layout is None:
            layout = F.ClusterLayout.get(counts, 1024, dev)
        out = F.act_fresh_quant_fwd([x], 4, layout)[0][0]
    g.replay()
    torch.cuda.synchronize()
    print("RESULT ok", float(out.abs().sum()))
except Exception as e:
    print("RESULT", type(e).__name__, str(e)[:300].replace("\n", " | "))
'''


def main():
    for scenario in ("warm", "cold_layout", "cold_table", "cold_lib"):
        code = CHILD % {"repo": REPO, "scenario": scenario}
        env = dict(os.environ, AMD_LOG_LEVEL="1")
        p = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=170, env=env)
        res = [ln for ln in p.stdout.splitlines() if ln.startswith("RESULT")]
        err = [ln for ln in p.stderr.splitlines() if "rror" in ln or "hip" in ln.lower()][-6:]
        print(f"{scenario:12s} rc={p.returncode} {res[-1] if res else '(no result line)'}")
        for ln in err:
            print("    stderr:", ln[:240])
        sys.stdout.flush()
        if p.returncode < 0 or p.returncode in (134, 139):
            print("child died from a signal: stopping here")
            return 1
    return 0


if __name__ == "__main__":
    sys.exit(main())
