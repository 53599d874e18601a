import torch, time
for mb in (57, 256, 1024):
    n = mb * 1024 * 1024 // 4
    x = torch.empty(n, device="cuda"); y = torch.randn(n, device="cuda")
    for name, fn in (("fill", lambda: x.fill_(1.0)), ("copy", lambda: x.copy_(y)), ("read-sum", lambda: y.sum())):
        for _ in range(3): fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): fn()
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / 20
        print(f"{mb} MB {name}: {us:.1f} us -> {mb * 1.048576 / us:.2f} TB/s of {('written' if name=='fill' else 'one-way')}")
