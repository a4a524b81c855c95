"""LDS-DMA weight-stream ceiling (mphip_debug_dma_stream): bytes per clock per CU a workgroup gets from an L2-resident tensor."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from megaportrait_hack_amd import _lib
lib = _lib.load()
dev = torch.device("cuda:0")
sink = torch.empty(256 * 512, device=dev)
st = torch.cuda.current_stream().cuda_stream
for slabs_in_tensor in (54, 108):          # 96->96 (1.33 MB), 192->192 per co tile (2.65 MB)
    w = torch.randn(slabs_in_tensor * 12288 // 2, device=dev).view(torch.float16) if False else torch.randn(slabs_in_tensor * 6144, device=dev)
    for lag in (1, 2, 3):
        for barrier in (1, 0):
            slabs = 54 * 8
            for _ in range(2):
                _lib.check(lib.mphip_debug_dma_stream(w.data_ptr(), slabs_in_tensor, slabs, lag, barrier, sink.data_ptr(), 256, st), "dma_stream")
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                lib.mphip_debug_dma_stream(w.data_ptr(), slabs_in_tensor, slabs, lag, barrier, sink.data_ptr(), 256, st)
            e1.record(); torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / 10
            gb = 256 * slabs * 24576 / 1e9
            print(f"tensor {slabs_in_tensor * 24576 / 1e6:.2f} MB  lag {lag} barrier {barrier}: {ms:.3f} ms for {gb:.2f} GB = {gb / ms:.2f} TB/s = {gb / ms * 1e3 / 256:.1f} GB/s per CU "
                  f"(~{gb / ms * 1e12 / 256 / 2.4e9:.1f} B/clk at 2.4 GHz); per slab {ms * 1e3 / slabs:.3f} us", flush=True)
