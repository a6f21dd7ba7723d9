import torch, time
dev="cuda:0"
n=32*256*4*55*55
a=torch.randn(n,device=dev); b=torch.randn(n,device=dev); c=torch.empty_like(a)
x=torch.randn(n//4,device=dev)
def t(fn,reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s,e=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e)/reps
ms=t(lambda: torch.add(a,b,out=c)); print(f"add: {ms*1e3:.0f} us  {3*n*4/ms/1e9:.2f} TB/s")
ms=t(lambda: c.copy_(a)); print(f"copy: {ms*1e3:.0f} us  {2*n*4/ms/1e9:.2f} TB/s")
ms=t(lambda: torch.relu(a,out=c) if False else torch.clamp_min(a,0,out=c)); print(f"relu: {ms*1e3:.0f} us  {2*n*4/ms/1e9:.2f} TB/s")
ms=t(lambda: a.sum()); print(f"sum(read only): {ms*1e3:.0f} us  {n*4/ms/1e9:.2f} TB/s")
ms=t(lambda: c.zero_()); print(f"fill(write only): {ms*1e3:.0f} us  {n*4/ms/1e9:.2f} TB/s")
