import torch
dev = torch.device('cuda:0')
n = 8192*64*2*576
a = torch.empty(n, device=dev); b = torch.empty(n, device=dev)
def t(f, reps=10):
    for _ in range(3): f()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for s,e in ev:
        s.record(); f(); e.record()
    torch.cuda.synchronize()
    return sorted(s.elapsed_time(e) for s,e in ev)[reps//2]
ms = t(lambda: a.fill_(1.0)); print(f"fill  {n*4/1e9:.2f} GB: {ms:.3f} ms -> {n*4/ms/1e9:.2f} TB/s write")
ms = t(lambda: b.copy_(a)); print(f"copy  {n*4/1e9:.2f} GB: {ms:.3f} ms -> {2*n*4/ms/1e9:.2f} TB/s read+write")
ms = t(lambda: a.sum()); print(f"sum   {n*4/1e9:.2f} GB: {ms:.3f} ms -> {n*4/ms/1e9:.2f} TB/s read")
