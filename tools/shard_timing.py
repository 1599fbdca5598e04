#!/usr/bin/env python3
"""Kernel time of ONE rank's share of the frame for world = 1, 2, 4, 8 (all on this GPU, one after the other):
how far the per-rank launch is from 1/world of the full frame (tail of the persistent kernel, fixed costs)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from raytracer_amd import api, scenes
from raytracer_amd import dist as rdist
spec = scenes.c4_blob63k()
W, H = spec.width, spec.height
g = api.GpuScene(spec, n_threads=8)
stream = torch.cuda.current_stream().cuda_stream
base = None
for world in (1, 2, 4, 8):
    n_max = rdist.max_tiles(W, H, world)
    packed = torch.zeros((n_max * 64, 3), dtype=torch.float64, device="cuda")
    for rank in sorted({0, world - 1}):
        for _ in range(2):
            g.render_tiles_device(rank, world, d_color=packed.data_ptr(), stream=stream)
        torch.cuda.synchronize(); g.kernel_times()
        t0 = time.perf_counter()
        for _ in range(5):
            g.render_tiles_device(rank, world, d_color=packed.data_ptr(), stream=stream)
        torch.cuda.synchronize()
        wall = (time.perf_counter() - t0) / 5 * 1e3
        ms, n = g.kernel_times()
        if base is None:
            base = ms / n
        print(f"world {world} rank {rank}: kernel {ms / n:.3f} ms, call-to-sync {wall:.3f} ms, ideal {base / world:.3f} ms, efficiency {base / world / wall:.3f}", flush=True)
