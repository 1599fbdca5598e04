#!/usr/bin/env python3
"""How a rank's launch ENDS (needs the -DRT_TIMERS build: RTCORE_LIB=raytracer_amd/lib/variants/timers.so): waves alive
against time over the last millisecond of rank 0's share of C4 for world = argv[1] (default 8), and each wave's
drain (end - the moment it found the chunk counter exhausted).      python tools/tail_timeline.py [world] [c4|c4v]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
world = int(sys.argv[1]) if len(sys.argv) > 1 else 8
os.environ["RT_WAVE_DUMP"] = "/tmp/waves_tail.bin"
from raytracer_amd import api, scenes
from raytracer_amd import dist as rdist
spec = (scenes.c4_blob63k if (len(sys.argv) > 2 and sys.argv[2] == "c4v") else scenes.c4_shipped)()
g = api.GpuScene(spec, n_threads=8)
n_max = rdist.max_tiles(spec.width, spec.height, world)
packed = torch.zeros((n_max * 64, 3), dtype=torch.float64, device="cuda")
stream = torch.cuda.current_stream().cuda_stream
for _ in range(3):
    g.render_tiles_device(0, world, d_color=packed.data_ptr(), stream=stream)
    torch.cuda.synchronize()
ms, n = g.kernel_times()
raw = np.fromfile("/tmp/waves_tail.bin", dtype=np.uint64)
d = raw[: len(raw) - 2 * 262144].reshape(-1, 4).astype(np.float64)
d = d[d[:, 2] > 0]
t0 = d[:, 0].min()
start, exh, end = (d[:, 0] - t0) / 1e5, (d[:, 1] - t0) / 1e5, (d[:, 2] - t0) / 1e5       # 100 MHz wall clock -> ms
T = end.max()
print(f"world {world} rank 0: kernel {ms / n:.3f} ms (HIP events, avg of {n}); {len(d)} waves; last wave ends at {T:.3f} ms")
print(f"first wave finds the counter exhausted at {exh[exh > 0].min():.3f} ms, median {np.median(exh):.3f}, last {exh.max():.3f}")
print(f"drain per wave (end - exhausted) ms: median {np.median(end - exh):.3f} p90 {np.percentile(end - exh, 90):.3f} max {(end - exh).max():.3f}")
for back in (1.0, 0.8, 0.6, 0.5, 0.4, 0.3, 0.2, 0.1, 0.05):
    print(f"  {back:4.2f} ms before the end: {(end > T - back).sum():5d} of {len(d)} waves alive, {((end > T - back) & (exh < T - back)).sum():5d} of them already draining")
