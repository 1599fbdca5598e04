#!/usr/bin/env python3
"""Kernel time vs amount of work: full frame at reduced spp (world 1) and one rank's share (world 8)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from raytracer_amd import api, scenes
from raytracer_amd import dist as rdist
spec = scenes.c4_blob63k()
W, H = spec.width, spec.height
g = api.GpuScene(spec, n_threads=8)
stream = torch.cuda.current_stream().cuda_stream
def run(rank, world, **over):
    n_max = rdist.max_tiles(W, H, world)
    packed = torch.zeros((n_max * 64, 3), dtype=torch.float64, device="cuda")
    for _ in range(2):
        g.render_tiles_device(rank, world, d_color=packed.data_ptr(), stream=stream, **over)
    torch.cuda.synchronize(); g.kernel_times()
    for _ in range(4):
        g.render_tiles_device(rank, world, d_color=packed.data_ptr(), stream=stream, **over)
    torch.cuda.synchronize()
    ms, n = g.kernel_times()
    return ms / n
full = run(0, 1)
print(f"world 1 spp 100: {full:.3f} ms")
for spp in (50, 25, 12, 6):
    t = run(0, 1, samples_per_pixel=spp)
    print(f"world 1 spp {spp}: {t:.3f} ms, ideal {full * spp / 100:.3f}", flush=True)
for world in (2, 4, 8):
    t = run(0, world)
    print(f"world {world} rank 0 spp 100: {t:.3f} ms, ideal {full / world:.3f}", flush=True)
