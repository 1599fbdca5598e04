#!/usr/bin/env python3
"""What rank 0 of an 8-GPU run does per frame besides its kernel (one GPU stands in; the gather itself is not included):
render rank 0's tiles, k_resolve, quantise them, un-tile 8 packed 8-bit buffers."""
import os, sys, time, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from raytracer_amd import api, scenes, _ffi as F
from raytracer_amd import dist as rdist
spec = scenes.c4_blob63k()
W, H, world = spec.width, spec.height, 8
g = api.GpuScene(spec, n_threads=8)
stream = torch.cuda.current_stream().cuda_stream
n_max = rdist.max_tiles(W, H, world)
packed = torch.zeros((n_max * 64, 3), dtype=torch.float64, device="cuda")
packed8 = torch.zeros((n_max * 64, 3), dtype=torch.uint8, device="cuda")
bufs = [torch.zeros_like(packed8) for _ in range(world)]
rgb8 = torch.empty((H, W, 3), dtype=torch.uint8, device="cuda")
def step():                                     # bench.py's step without the gather
    g.render_tiles_device(0, world, d_color=packed.data_ptr(), stream=stream)
    F.check(F.core().rt_quantize_device(C.c_void_p(packed.data_ptr()), C.c_void_p(packed8.data_ptr()), packed.numel(), C.c_void_p(stream)))
    bufs[0].copy_(packed8)
    for r in range(world):
        rdist.untile_into(rgb8, bufs[r], W, H, r, world)
for _ in range(3): step()
torch.cuda.synchronize(); g.kernel_times()
t0 = time.perf_counter()
for _ in range(20): step()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 20 * 1e3
ms, n = g.kernel_times()
print(f"rank 0 of 8: step {dt:.3f} ms, render kernel {ms / n:.3f} ms, everything else {dt - ms / n:.3f} ms")
