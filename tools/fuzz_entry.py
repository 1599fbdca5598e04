#!/usr/bin/env python3
"""Entry-point equivalence on the random scenes of tools/fuzz_oracle.py (GPU only, no oracle): for every seed the frame of
rt_render must be reproduced BIT FOR BIT by the sharded entry points - rt_render_tiles_device and its no-allocation twin,
world sizes 1 / 2 / 3 / 5 with one GPU playing every rank, colour + radiance + ids, un-tiled with rt_untile_device - and by
the same frame rendered by the single-process device group (rt_init_devices, 2-5 virtual devices) and in several launches
(RT_SAMPLE_BUDGET_MB=1 on a fresh scene handle).
    python tools/fuzz_entry.py <first seed> <last seed + 1> [--extreme]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import fuzz_oracle as FZ                            # noqa: E402
from raytracer_amd import api                       # noqa: E402
from raytracer_amd import dist as rdist             # noqa: E402


def main():
    extreme = "--extreme" in sys.argv
    if extreme:
        sys.argv.remove("--extreme")
    lo, hi = int(sys.argv[1]), int(sys.argv[2])
    bad = 0
    stream = torch.cuda.current_stream().cuda_stream
    for seed in range(lo, hi):
        spec, rr = FZ.scene(seed, False, extreme)
        W, H = spec.width, spec.height
        g = api.GpuScene(spec)
        full, _ = g.render(want=("color", "radiance", "prim_id"), rr_start_bounce=rr)
        msgs = []
        for world in (1, 2, 3, 5):
            n_max = rdist.max_tiles(W, H, world)
            img = torch.full((H, W, 3), float("nan"), dtype=torch.float64, device="cuda")
            rad = torch.full((H, W, 3), float("nan"), dtype=torch.float64, device="cuda")
            ids = torch.full((H, W, 1), -7, dtype=torch.int32, device="cuda")
            for rank in range(world):
                pc = torch.zeros((n_max * 64, 3), dtype=torch.float64, device="cuda")
                pr = torch.zeros((n_max * 64, 3), dtype=torch.float64, device="cuda")
                pi = torch.zeros((n_max * 64, 1), dtype=torch.int32, device="cuda")
                g.render_tiles_device(rank, world, d_color=pc.data_ptr(), d_radiance=pr.data_ptr(), d_prim_id=pi.data_ptr(), stream=stream,
                                      rr_start_bounce=rr, noalloc=bool((rank + world + seed) & 1))
                rdist.untile_into(img, pc, W, H, rank, world)
                rdist.untile_into(rad, pr, W, H, rank, world)
                rdist.untile_into(ids, pi, W, H, rank, world)
            torch.cuda.synchronize()
            ok = (np.array_equal(img.cpu().numpy(), full["color"], equal_nan=True) and np.array_equal(rad.cpu().numpy(), full["radiance"], equal_nan=True)
                  and np.array_equal(ids.cpu().numpy()[..., 0], full["prim_id"]))
            bad += not ok
            msgs.append(f"w{world} {'OK' if ok else 'MISMATCH'}")
        g.close()
        gN = api.GpuScene(spec, devices=[0] * (2 + seed % 4))      # the single-process device group, on virtual devices
        grp, _ = gN.render(want=("color", "radiance", "prim_id"), rr_start_bounce=rr)
        gN.close()
        ok = all(np.array_equal(grp[k], full[k], equal_nan=True) for k in ("color", "radiance", "prim_id"))
        bad += not ok
        msgs.append(f"group{2 + seed % 4} {'OK' if ok else 'MISMATCH'}")
        os.environ["RT_SAMPLE_BUDGET_MB"] = "1"
        g = api.GpuScene(spec)
        many, _ = g.render(want=("color", "radiance", "prim_id"), rr_start_bounce=rr)
        g.close()
        del os.environ["RT_SAMPLE_BUDGET_MB"]
        ok = all(np.array_equal(many[k], full[k], equal_nan=True) for k in ("color", "radiance", "prim_id"))
        bad += not ok
        msgs.append(f"launches {'OK' if ok else 'MISMATCH'}")
        print(seed, f"{W}x{H}x{spec.spp} rr{rr}", " ".join(msgs), flush=True)
    print("mismatches:", bad)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
