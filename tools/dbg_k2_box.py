import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch, ctypes
from megaportrait_hack_amd import ops, _lib
from bench_warps import fields
lib = _lib.load()
B = 8
f = fields(B)["faithful"].cuda()
v = torch.randn(B, 96, 16, 64, 64, device="cuda")
out, coords, idx = ops.warp_volume(v, f, return_coords=True)
box = torch.zeros(B * 8, dtype=torch.int32, device="cuda")
lib.mphip_warp_sample_box(ctypes.c_void_p(coords.data_ptr()), ctypes.c_void_p(box.data_ptr()), B, 16, 64, 64, None)
torch.cuda.synchronize()
print(box.view(B, 8).cpu())
fl = coords.floor()
print("floor max per axis", fl.amax(dim=(0, 1, 2, 3)), "min", fl.amin(dim=(0, 1, 2, 3)))
