"""dev: what a pure write stream / copy reaches on this MI355X (context for K2's 201 MB write and the upsample)."""
import torch, time
dev = torch.device("cuda:0")
n = 8 * 96 * 16 * 64 * 64
x = torch.empty(n, device=dev)
y = torch.empty(n, device=dev)
def t(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
ms = t(lambda: x.zero_())
print(f"fill  201 MB: {ms*1e3:.1f} us = {n*4/ms/1e9:.2f} TB/s written")
ms = t(lambda: y.copy_(x))
print(f"copy  201 MB: {ms*1e3:.1f} us = {n*4/ms/1e9:.2f} TB/s read + {n*4/ms/1e9:.2f} TB/s written")
big = torch.empty(4 * n, device=dev)
ms = t(lambda: big.zero_())
print(f"fill  805 MB: {ms*1e3:.1f} us = {4*n*4/ms/1e9:.2f} TB/s written")
