#!/usr/bin/env python3
"""How often does the last bit of Schlick's `(1 - cosine) ** 5` send a path another way (Trace.hs:54-57: `randD < reflectProb`)?
A frame filled by C2's glass sphere (one glass decision per path: Sphere.hs tests the near root only, so a refracted ray leaves the
sphere without a second hit); GPU against the CPU oracle, pixels whose radiance differs by more than 1e-12 are counted and listed.
Measured (2048 x 2048 x 96 = 403 M paths, 350 M glass decisions): no differing pixel, with the multiplication chain x*x*x*x*x as with the
correctly rounded pow5_cr - the last bit of x^5 reaches `0.04 + 0.96 x^5` one time in twenty and a random number's 2^-32 grid lies
within an ulp of that 3e-8 of the time: ~0.4 expected flips per 400 M paths for the chain, 450 times fewer for pow5_cr.
    python tools/schlick_flip_probe.py [size] [spp]          (RTCORE_LIB=<other build> to compare builds)"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import oracle as O                     # noqa: E402
from raytracer_amd import api, scenes              # noqa: E402


def glass_scene(size, spp):
    spec = scenes.c2_spheres(width=size, height=size, spp=spp, max_bounces=7)
    spec.name = "glass_probe"
    spec.camera = {"lookFrom": [-3.2, 1.0, 0.0], "lookAt": [0.0, 1.0, 0.0], "vUp": [0.0, 1.0, 0.0], "vfov": 34.0, "aperture": 0.0, "focusDist": 3.0}
    return spec


def main():
    size = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
    spp = int(sys.argv[2]) if len(sys.argv) > 2 else 32
    spec = glass_scene(size, spp)
    t0 = time.time()
    o = O.OracleScene(spec)
    ref = o.render(n_threads=min(16, os.cpu_count() or 8))
    o.close()
    t1 = time.time()
    g = api.GpuScene(spec)
    res, st = g.render(collect_stats=1)
    g.close()
    ids, n = np.unique(ref["prim_id"], return_counts=True)
    glass_first = float(n.max() / ref["prim_id"].size)          # the glass sphere is the most frequent first hit
    with np.errstate(invalid="ignore"):
        rel = np.abs(res["radiance"] - ref["radiance"]) / np.maximum(1.0, np.abs(ref["radiance"]))
    bad = np.argwhere((rel > 1e-12).any(-1))
    print(f"{size}x{size}x{spp}: oracle {t1 - t0:.1f} s; {st.paths} paths, {st.shaded_hits} shaded hits, glass at the first hit in {glass_first:.2f} of the pixels")
    print(f"pixels differing by more than 1e-12: {len(bad)}; ids differing: {int((res['prim_id'] != ref['prim_id']).sum())}; worst {float(np.nanmax(rel)):.2e}")
    for py, px in bad[:10]:
        print(f"    pixel ({px}, {py}): gpu {res['radiance'][py, px]} oracle {ref['radiance'][py, px]}")
    return 1 if len(bad) else 0


if __name__ == "__main__":
    sys.exit(main())
