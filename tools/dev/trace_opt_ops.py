import os, sys
import torch
from torch.profiler import ProfilerActivity, profile
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench_block as BB
torch.cuda.set_device(0)
kw = dict(BB.REFERENCE_LOG); kw.update(samples=2, epochs=1, warmup=False); kw.pop("reference_s_per_layer")
BB.run(**kw)
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    kw.update(epochs=2)
    BB.run(**kw)
for e in prof.key_averages(group_by_input_shape=True):
    n = e.key
    if any(t in n for t in ("sum", "mean", "norm", "amax", "amin", "max", "min", "Sum", "Mean")) and e.device_time_total > 0:
        print(f"{n[:40]:40s} calls {e.count:4d} cuda {e.device_time_total/max(e.count,1):7.1f}us shapes {str(e.input_shapes)[:110]}")
