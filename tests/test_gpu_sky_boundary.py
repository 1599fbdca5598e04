"""Sky texels AT texel boundaries (SkySphere.hs:35-41): frames whose primary rays sweep through a column or row boundary of the
equirect sky in steps of 1e-17 ... 1e-14 - where the texel is decided by how GHC's atan2, the division by 2 pi, the sum and the
product round.  Every pixel must equal the oracle's (tools/sky_boundary_probe.py; DESIGN.md section 8, sky texel)."""
import os
import sys

import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))


@pytest.mark.gpu
@pytest.mark.parametrize("sw,sh", [(64, 32), (100, 50)])
def test_boundary_directions_get_the_reference_texel(need_gpu, sw, sh):
    import sky_boundary_probe as P
    bad, total = P.probe(sw, sh, steps=(1e-17, 1e-16, 1e-15, 1e-14), verbose=False)
    assert total > 1_000_000 and bad == 0


@pytest.mark.gpu
@pytest.mark.parametrize("sw,sh", [(1, 1), (2, 1), (1, 2), (3, 2), (4, 4), (5, 3), (7, 1)])
def test_tiny_sky_textures(need_gpu, sw, sh):
    """Skies of a few texels: column boundaries half a turn apart (the table tests' bracketing cannot hold, every lane takes the
    exact path), boundaries ON the coordinate axes, one-row and one-column textures."""
    import numpy as np
    from oracle import oracle as O
    from raytracer_amd import api, scenes
    img = np.random.default_rng(sw * 16 + sh).uniform(0.1, 1.0, size=(sh, sw, 3)).astype(np.float32)
    for fmt, tex in (("rgbf", img), ("rgb8", (img * 255).astype(np.uint8))):
        for look_at in ([1.0, 0.0, 0.0], [-1.0, 0.2, 0.0], [0.0, 0.1, 1.0], [0.0, -0.3, -1.0], [0.0, 1.0, 1e-3], [0.6, -0.8, 0.0]):
            spec = scenes.c2_spheres(width=96, height=64, spp=2, max_bounces=1)
            spec.config_objects = [("sphere", [0.0, -1e6, 0.0], 1.0, None)]
            spec.obj_meshes = []
            spec.sky = (fmt, tex)
            spec.camera = {"lookFrom": [0.0, 0.0, 0.0], "lookAt": look_at, "vUp": [0.0, 1.0, 0.0], "vfov": 120.0, "aperture": 0.0, "focusDist": 1.0}
            o = O.OracleScene(spec)
            ref = o.render(n_threads=8)
            o.close()
            g = api.GpuScene(spec)
            try:
                for mode in (0, 1):
                    res, _ = g.render(box_precision=mode, want=("radiance",))
                    assert np.array_equal(res["radiance"], ref["radiance"]), (fmt, look_at, mode)
            finally:
                g.close()
