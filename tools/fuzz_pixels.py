#!/usr/bin/env python3
"""For one seed of tools/fuzz_oracle.py: the pixels where the GPU (box_precision = argv[2]) and the oracle differ.
    python tools/fuzz_pixels.py <seed> <mode> [--big] [--extreme]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import fuzz_oracle as F                            # noqa: E402
from oracle import oracle as O                     # noqa: E402
from raytracer_amd import api                      # noqa: E402

seed, mode = int(sys.argv[1]), int(sys.argv[2])
spec, rr = F.scene(seed, "--big" in sys.argv, "--extreme" in sys.argv)
print("camera", spec.camera, "aa", spec.antialiasing, "sky", None if spec.sky is None else (spec.sky[0], spec.sky[1].shape), "mb", spec.max_bounces)
o = O.OracleScene(spec)
ref = o.render(rr_start_bounce=rr)
g = api.GpuScene(spec)
res, st = g.render(box_precision=mode, rr_start_bounce=rr, collect_stats=1)
rel = np.abs(res["radiance"] - ref["radiance"]) / np.maximum(1.0, np.abs(ref["radiance"]))
ys, xs = np.nonzero((res["prim_id"] != ref["prim_id"]) | np.any(rel > 1e-12, axis=-1))
print("keys", sorted(res.keys()), "lookFrom", spec.camera["lookFrom"])
print("differing pixels:", len(ys))
for y, x in list(zip(ys, xs))[:20]:
    print((x, y), "gpu id", res["prim_id"][y, x], res["radiance"][y, x], "oracle id", ref["prim_id"][y, x], ref["radiance"][y, x])
for k in ("depth", "normal"):
    if k in res and k in ref:
        for y, x in list(zip(ys, xs))[:20]:
            print(k, (x, y), res[k][y, x], ref[k][y, x])
