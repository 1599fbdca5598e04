import os, sys
sys.path.insert(0, os.getcwd())
import torch
from raytracer_amd import api, scenes
for (w, h, spp) in [(64, 64, 1), (256, 256, 1), (1920, 1080, 1), (1920, 1080, 2), (1920, 1080, 4)]:
    spec = scenes.c4_blob63k(width=w, height=h, spp=spp)
    g = api.GpuScene(spec, n_threads=8)
    for _ in range(3):
        g.render(want=("color",))
    g.kernel_times()
    for _ in range(5):
        res, st = g.render(want=("color",))
    ms, n = g.kernel_times()
    print(f"{w}x{h} spp {spp}: kernel {ms / n:.3f} ms (items {w * h * spp})", flush=True)
    g.close()
