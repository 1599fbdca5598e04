"""Kernel time of C4 as shipped against the size of its sky texture (how much of the frame is the sky's cache footprint?)."""
import sys, numpy as np
sys.path.insert(0, '.')
from raytracer_amd import api, scenes
for size in [(2048, 1024), (1184, 592), (512, 256), (64, 32)]:
    spec = scenes.c4_shipped(sky_size=size)
    g = api.GpuScene(spec, n_threads=16)
    ms = []
    for it in range(6):
        g.render(want=("rgb8",))
        t, n = g.kernel_times()
        ms.append(round(t / max(n, 1), 3))
    print(size, "texture MB", size[0] * size[1] * 12 / 1e6, "kernel ms", ms[2:], flush=True)
    g.close()
