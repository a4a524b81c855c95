"""Root-cause probe for the stale graphed loss: pure PyTorch, no mphip kernel involved.
A hipGraph that contains ATen's multi-block reduction (`x.mean()` over > ~64 k elements: Reduce.cuh zeroes its semaphores with
hipMemsetAsync — a MEMSET NODE in the captured graph — and the LAST block to finish writes the result) is replayed with changing inputs;
a stale result means the memset node did not run before the reduction kernel of the same replay (the semaphore still holds the
previous replay's count, no block sees itself as the last one, the output is never written).
usage: repro_graph_memset.py [n_elements] [replays] [filler kernels before the reduction] [kernels after it]"""
import sys
import torch
n = int(sys.argv[1]) if len(sys.argv) > 1 else 196608
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 200
filler = int(sys.argv[3]) if len(sys.argv) > 3 else 20
after = int(sys.argv[4]) if len(sys.argv) > 4 else 0
dev = torch.device("cuda:0")
x = torch.zeros(n, device=dev)
w = torch.randn(256, 256, device=dev)
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(3):
        a = w
        for _ in range(filler): a = a @ w * 1e-2
        l = (x * 2.0).mean() + 0.0 * a.sum()
torch.cuda.current_stream().wait_stream(s)
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    a = w
    for _ in range(filler): a = a @ w * 1e-2
    big = (x * 2.0).mean()          # multi-block reduction: semaphores zeroed by a memset node
    small = (x[:1024] * 2.0).mean() # single-block reduction: no semaphores
    l = big + 0.0 * a.sum()
    b = w
    for _ in range(after): b = b @ w * 1e-2    # (a backward pass worth of kernels behind the reduction)
stale_big = stale_small = 0
for i in range(reps):
    v = float(i % 7 + 1)
    x.fill_(v)
    g.replay()
    torch.cuda.synchronize()
    stale_big += abs(big.item() - 2.0 * v) > 1e-6
    stale_small += abs(small.item() - 2.0 * v) > 1e-6
print(f"n={n} replays={reps} filler={filler} after={after}: stale multi-block reduction results {stale_big}, stale single-block results {stale_small}")
