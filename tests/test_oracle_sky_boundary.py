"""The ORACLE's sky texel at texel boundaries (CPU only): frames whose primary rays sweep through a column boundary of a 64 x 32 sky
(tools/sky_boundary_probe.py) must show, pixel by pixel, the texel that SkySphere.hs:35-40 gives with GHC's atan2 - the RealFloat
class default, pyref.hs_atan2 - and the test only counts if libm's atan2 would have given a different texel for some of them."""
import math
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))


def test_oracle_uses_ghc_atan2_at_column_boundaries():
    import sky_boundary_probe as P
    from oracle import oracle as O
    from oracle.pyref import hs_atan2
    from raytracer_amd import scenes
    sw, sh = 64, 32
    img = np.random.default_rng(5).uniform(0.1, 1.0, size=(sh, sw, 3)).astype(np.float32)
    spec = scenes.c2_spheres(width=P.W, height=P.H, spp=1, max_bounces=0)
    spec.config_objects = [("sphere", [0.0, -1e6, 0.0], 1.0, None)]
    spec.obj_meshes = []
    spec.antialiasing = False
    spec.sky = ("rgbf", img)
    o = O.OracleScene(spec)
    libm_would_differ = 0
    try:
        for k in (1, 5, 8, 16, 24, 40, 56, 63):
            phi = 2 * math.pi * (k / sw - 0.5)
            for step in (1e-17, 1e-16):
                cam = P.sweep_camera((math.cos(phi), 0.4, math.sin(phi)), (-math.sin(phi), 0.0, math.cos(phi)), (0.0, 1.0, 0.0), step)
                got = o.render(camera=cam, n_threads=4)["radiance"]
                d = P.directions(cam)
                for py in range(P.H):
                    for px in range(P.W):
                        x, y, z = d[py, px]
                        j = min(sh - 1, max(0, math.floor((0.5 - math.asin(y) / math.pi) * sh)))
                        i = min(sw - 1, max(0, math.floor((0.5 + hs_atan2(z, x) / (2 * math.pi)) * sw)))
                        assert (got[py, px] == img[j, i].astype(np.float64)).all(), (k, step, px, py)
                        i_libm = min(sw - 1, max(0, math.floor((0.5 + math.atan2(z, x) / (2 * math.pi)) * sw)))
                        libm_would_differ += i_libm != i
    finally:
        o.close()
    assert libm_would_differ > 100
