"""torch._foreach_copy_ over lists whose pairs differ in dtype (fp16 gradient -> fp32 bucket view and back)."""
import torch
g = torch.Generator(device="cuda").manual_seed(3)
shapes = [((4096, 1), torch.float32), ((11008, 1), torch.float32), ((4096,), torch.float16), ((1,), torch.float32)]
src = [torch.randn(s, device="cuda", generator=g).to(dt) for s, dt in shapes]
flat = torch.empty(sum(t.numel() for t in src), dtype=torch.float32, device="cuda")
views, off = [], 0
for t in src:
    views.append(flat[off:off + t.numel()].view(t.shape)); off += t.numel()
torch._foreach_copy_(views, src)
torch.cuda.synchronize()
print("pack   :", [float((v - t.float()).abs().max()) for v, t in zip(views, src)])
dst = [torch.empty_like(t) for t in src]
torch._foreach_copy_(dst, views)
torch.cuda.synchronize()
print("unpack :", [float((d.float() - t.float()).abs().max()) for d, t in zip(dst, src)])
for v, t in zip(views, src): v.copy_(t)
for d, v in zip(dst, views): d.copy_(v)
torch.cuda.synchronize()
print("loop   :", [float((d.float() - t.float()).abs().max()) for d, t in zip(dst, src)])
