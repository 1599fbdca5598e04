#!/usr/bin/env python3
"""Multi-GPU readiness on ONE GPU (VERDICT r01 #4a): the GPU plays every rank of world 2 / 4 / 8 in turn and times each
rank's share of the frame (k_render_tiles, HIP events).  The frame of an N-GPU run is set by the SLOWEST rank, so the
figure of merit is max over ranks against the ideal (full frame / N).  RT_TILE_SKEW=k selects the tile -> rank map
(tx + k ty) mod N for both the library and this script.

    python tools/rank_timing.py [c4|c4v|c5] [width height spp]
"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from raytracer_amd import api, scenes
from raytracer_amd import dist as rdist

which = sys.argv[1] if len(sys.argv) > 1 else "c4"
mk = {"c4": scenes.c4_shipped, "c4v": scenes.c4_blob63k, "c5": scenes.c5_multi}[which]
spec = mk(*(int(x) for x in sys.argv[2:5])) if len(sys.argv) > 4 else mk()
W, H = spec.width, spec.height
g = api.GpuScene(spec, n_threads=8)
stream = torch.cuda.current_stream().cuda_stream
reps = 1 if which == "c5" else 3


def run(rank, world):
    n_max = rdist.max_tiles(W, H, world)
    packed = torch.zeros((n_max * 64, 3), dtype=torch.float64, device="cuda")
    g.render_tiles_device(rank, world, d_color=packed.data_ptr(), stream=stream)
    torch.cuda.synchronize(); g.kernel_times()
    for _ in range(reps):
        g.render_tiles_device(rank, world, d_color=packed.data_ptr(), stream=stream)
    torch.cuda.synchronize()
    ms, n = g.kernel_times()
    return ms / n


out = {"workload": spec.name, "size": [W, H, spec.spp], "tile_skew": rdist.tile_skew(), "worlds": {}}
full = run(0, 1)
out["full_frame_ms"] = round(full, 3)
print(f"{spec.name} {W}x{H} spp {spec.spp}, tile skew {rdist.tile_skew()}: full frame {full:.3f} ms", flush=True)
for world in (2, 4, 8):
    t = [run(r, world) for r in range(world)]
    mx, mean = max(t), sum(t) / world
    out["worlds"][world] = {"rank_ms": [round(x, 3) for x in t], "max_ms": round(mx, 3), "mean_ms": round(mean, 3),
                            "ideal_ms": round(full / world, 3), "max_over_mean": round(mx / mean, 4),
                            "kernel_efficiency": round(full / (world * mx), 4)}
    print(f"world {world}: ranks {' '.join(f'{x:.2f}' for x in t)} ms | max {mx:.3f} mean {mean:.3f} ideal {full / world:.3f} | "
          f"max/mean {mx / mean:.3f} | kernel-only efficiency {full / (world * mx):.3f}", flush=True)
print(json.dumps(out))
