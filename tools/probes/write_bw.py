"""Pure-write / copy bandwidth of the device (torch kernels), to price the convolution epilogues against."""
import torch

for mb in (51, 103, 205, 822):
    n = mb * 1024 * 1024 // 2
    y = torch.empty(n, device="cuda", dtype=torch.float16)
    x = torch.randn(n, device="cuda", dtype=torch.float16)
    for name, fn in (("fill", lambda: y.fill_(1.0)), ("copy", lambda: y.copy_(x)), ("relu", lambda: torch.relu_(y))):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            fn()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 20 * 1e3
        moved = n * 2 * (1 if name == "fill" else 2)
        print(f"{mb:4d} MiB {name}: {us:8.1f} us  {moved / us / 1e6:7.2f} TB/s ({'write only' if name == 'fill' else 'read + write'})")
