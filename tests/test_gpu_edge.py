"""GPU parity on edge inputs the reference accepts without complaint: degenerate geometry, inside-out boxes, frames
smaller than a tile, nothing traced at all, directional lights only.  Same bar as tests/test_gpu_parity.py: ids equal on
every pixel, radiance within 1e-12."""
import numpy as np
import pytest

from oracle import oracle as O
from raytracer_amd import api, scenes

pytestmark = pytest.mark.gpu


def both(spec, **over):
    g = api.GpuScene(spec)
    res = {m: g.render(box_precision=m, **over)[0] for m in (0, 1, 2)}
    g.close()
    o = O.OracleScene(spec)
    ref = o.render(**over)
    o.close()
    return res, ref


def check(res, ref):
    for mode, r in res.items():
        assert (r["prim_id"] == ref["prim_id"]).all(), f"mode {mode}: {(r['prim_id'] != ref['prim_id']).sum()} ids differ"
        d = np.abs(r["radiance"] - ref["radiance"])
        assert np.nanmax(d) <= 1e-12 and np.isnan(r["radiance"]).sum() == np.isnan(ref["radiance"]).sum(), mode
        assert (r["rgb8"] == ref["rgb8"]).all(), mode


def test_negative_and_zero_radius_spheres(need_gpu):
    # Sphere.hs:20-22 boxes a sphere as c -+ r: a negative radius gives an inside-out box that the reference's slab test and
    # box unions treat in their own way; the core keeps to the reference's tree for such scenes (no fast tree)
    spec = scenes.c2_spheres(width=120, height=68, spp=3)
    spec.config_objects.append(("sphere", [-0.5, 0.6, 0.8], -0.6, "red"))
    spec.config_objects.append(("sphere", [0.5, 0.3, -0.8], 0.0, "blue"))
    spec.leaf_threshold = 2
    check(*both(spec))


def test_degenerate_triangles_and_duplicates(need_gpu):
    spec = scenes.c3_blob988(width=96, height=54, spp=3, max_bounces=4)
    spec.config_objects += [("triangle", [0.0, 1.0, 4.0], [0.0, 1.0, 4.0], [0.0, 1.0, 4.0], "blob"),        # a point
                            ("triangle", [-1.0, 0.5, 3.0], [0.0, 0.5, 3.0], [1.0, 0.5, 3.0], "blob"),       # a segment
                            ("triangle", [-0.4, 1.2, 3.2], [0.4, 1.2, 3.2], [0.0, 1.8, 3.2], "wall"),
                            ("triangle", [-0.4, 1.2, 3.2], [0.4, 1.2, 3.2], [0.0, 1.8, 3.2], "floor")]     # an exact duplicate, another material
    check(*both(spec))


@pytest.mark.parametrize("size", [(1, 1), (7, 3), (8, 8), (9, 17)])
def test_frames_around_the_tile_size(need_gpu, size):
    spec = scenes.c1_testing(width=size[0], height=size[1], spp=5, antialiasing=True, max_bounces=3)
    check(*both(spec))


def test_nothing_traced_and_directional_lights_only(need_gpu):
    spec = scenes.c1_testing(width=64, height=40, spp=2, antialiasing=True, max_bounces=0)      # traceLoop _ 0: pure background
    spec.sky = scenes.sky_gradient_texture()
    res, ref = both(spec)
    check(res, ref)
    assert (ref["prim_id"] == -1).all()
    spec = scenes.c2_spheres(width=96, height=54, spp=3)
    spec.lights = [{"tag": "DirectionalLight", "direction": [0.2, -1.0, 0.1], "intensity": [0.7, 0.7, 0.6]},
                   {"tag": "DirectionalLight", "direction": [0.0, 1.0, 0.0], "intensity": [0.3, 0.3, 0.3]}]   # one from below: every term culled or blocked
    check(*both(spec))
