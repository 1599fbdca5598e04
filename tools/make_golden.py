#!/usr/bin/env python3
"""Generates tests/golden/*.npz: small input/output vectors of the hot path produced by the CPU oracle
(oracle/oracle.cpp) in this container.  The reference itself cannot run here (Haskell, no ghc — SURVEY §8c)
and holds no fixtures, so these goldens pin the ORACLE's current behaviour (regression guard) and give the
GPU tests oracle-independent data files; they are not reference outputs: parity stays "unpinned".

    python tools/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import oracle as O            # noqa: E402
from raytracer_amd import scenes          # noqa: E402


def cases():
    from tests.golden_cases import golden_cases
    yield from golden_cases()
    return
    yield "c1_det_96x54", scenes.c1_testing(width=96, height=54), {}
    yield "c3_det_64x36", scenes.c3_blob988(width=64, height=36, spp=1, antialiasing=False, max_bounces=1), {}
    yield "c4_det_64x36", scenes.c4_blob63k(width=64, height=36, spp=1, antialiasing=False, max_bounces=1), {}
    yield "c2_stoch_48x27_spp4", scenes.c2_spheres(width=48, height=27, spp=4), {}
    yield "c3_stoch_48x27_spp4_seed9", scenes.c3_blob988(width=48, height=27, spp=4), {"seed": 9}
    s = scenes.c1_testing(width=48, height=27, spp=2, antialiasing=True, max_bounces=3)
    s.sky = scenes.sky_gradient_texture()
    yield "c1_sky_48x27_spp2", s, {}


def main():
    out = os.path.join(ROOT, "tests", "golden")
    os.makedirs(out, exist_ok=True)
    for name, spec, over in cases():
        o = O.OracleScene(spec)
        res, cnt = o.render(n_threads=4, counters=True, **over)
        info = o.bvh_info()
        np.savez_compressed(os.path.join(out, name + ".npz"), radiance=res["radiance"], color=res["color"], rgb8=res["rgb8"],
                            prim_id=res["prim_id"], counters=np.array([cnt[k] for k in sorted(cnt)], np.int64),
                            counter_names=np.array(sorted(cnt)), bvh=np.array([info["n_nodes"], info["n_leaves"], info["depth"], info["n_prims"]]))
        print(name, res["radiance"].shape, os.path.getsize(os.path.join(out, name + ".npz")), "bytes")


if __name__ == "__main__":
    main()
