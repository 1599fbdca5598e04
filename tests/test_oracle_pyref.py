"""Two independent restatements of the reference must agree bit for bit.

The C++ oracle (oracle/oracle.cpp) is what the GPU is compared with, and nothing from the reference pins it (no ghc,
no tests, no assets: parity unpinned).  oracle/pyref.py is a second reading of the same Haskell sources — recursive and
list-shaped like the reference, written separately — so a slip in either restatement (a tie rule, an interval bound, the
order of a sum, ray0 at depth 0 ...) shows up here as a difference.  Frames are tiny: pyref is pure Python."""
import numpy as np
import pytest

from oracle import oracle as O
from oracle import pyref
from raytracer_amd import scenes


def both(spec):
    scene, cam = pyref.scene_from_spec(spec)
    rad, col, rgb, ids = pyref.render(scene, cam, spec.width, spec.height, spec.spp, spec.antialiasing, spec.max_bounces,
                                      spec.exposure, spec.gamma, seed=spec.seed)
    o = O.OracleScene(spec)
    ref = o.render(n_threads=2)
    o.close()
    return (np.array(rad), np.array(col), np.array(rgb, dtype=np.uint8), np.array(ids, dtype=np.int32)), ref


def check(spec):
    (rad, col, rgb, ids), ref = both(spec)
    assert (ids == ref["prim_id"]).all(), f"{(ids != ref['prim_id']).sum()} primary ids differ"
    assert np.array_equal(rad, ref["radiance"]), f"radiance differs by up to {np.abs(rad - ref['radiance']).max():.3e}"
    assert np.abs(col - ref["color"]).max() <= 1e-15          # `**` goes through two pow implementations
    assert (rgb == ref["rgb8"]).all()


def test_c1_deterministic_and_stochastic():
    check(scenes.c1_testing(width=24, height=14, spp=1))                                       # deterministic: AA off, maxBounces 1
    check(scenes.c1_testing(width=20, height=12, spp=2, antialiasing=True, max_bounces=4))     # emissive lamp, two lights


def test_c2_glass_metal_planes_duplicates():
    spec = scenes.c2_spheres(width=24, height=14, spp=3)                                       # glass, metal, 3 lights, duplicated ground
    spec.config_objects.append(("plane", [0.0, 0.0005, 0.0], [0.0, 1.0, 0.0], "blue"))
    spec.leaf_threshold = 2
    check(spec)


def test_c3_mesh_depth_limited_leaves():
    check(scenes.c3_blob988(width=20, height=12, spp=2, max_bounces=3, bvh_max_depth=5))       # 31-triangle leaves
    check(scenes.c3_blob988(width=16, height=10, spp=1, max_bounces=5, leaf_threshold=1))


@pytest.mark.parametrize("fmt", ["rgbf", "rgb8"])
def test_sky_sphere_and_aperture(fmt):
    spec = scenes.c1_testing(width=18, height=10, spp=2, antialiasing=True, max_bounces=3)
    spec.sky = scenes.sky_gradient_texture(fmt=fmt)
    spec.camera["aperture"] = 0.3
    check(spec)


def test_single_leaf_root_and_no_lights():
    spec = scenes.c2_spheres(width=16, height=10, spp=2)
    spec.leaf_threshold = 64                                                                   # the whole scene in the root leaf: no box test at all
    spec.lights = []
    check(spec)


def test_planes_outside_their_slab_box():
    # a leaf visited second after its sibling produced a hit receives [enter, t firstHit] (BVH.hs:137-138): both restatements
    # must see the plane there, beyond its leaf box's exit
    from plane_cases import random_plane_scene, two_object_scene
    check(two_object_scene(width=32, height=18, spp=1))
    check(two_object_scene(width=20, height=12, spp=2, max_bounces=3, antialiasing=True))
    check(random_plane_scene(5, n_tris=60, width=20, height=12, spp=1))


def test_sphere_hit_where_it_touches_its_box():
    """Sphere.hs:20-27 + BVH.hs:124-139: the centre ray's t1 is one ulp below the leaf's slab entry, so with
    two leaves under an internal root both restatements see THROUGH the sphere there; with a single-leaf root both see it."""
    from sphere_cases import two_sphere_scene
    for lt, want in ((1, -1), (4, 0)):
        spec = two_sphere_scene(lt)
        (rad, col, rgb, ids), ref = both(spec)
        assert ids[8, 16] == want and ref["prim_id"][8, 16] == want
        check(spec)
    check(two_sphere_scene(1, lights=True, max_bounces=3, width=17, height=9))


def test_nan_bounce_terms_follow_ghc_min_max():
    """VERDICT r02 "missing" #6: Haskell's max / min on a NaN.  clamp NaN 0 10 = 0 in the reference (min hi NaN = NaN, max lo NaN =
    lo), a path without a hit is returned unclamped.  Both restatements must agree - NaN positions included."""
    from nan_cases import nan_scene
    assert pyref.clamp3((float("nan"), 11.0, -1.0), 0, 10.0) == (0, 10.0, 0)
    spec = nan_scene(width=24, height=14, spp=2, max_bounces=3)
    (rad, col, rgb, ids), ref = both(spec)
    assert np.isnan(ref["radiance"]).any() and np.isinf(ref["radiance"]).any() and np.isfinite(ref["radiance"]).any()
    assert np.array_equal(rad, ref["radiance"], equal_nan=True)
    assert (ids == ref["prim_id"]).all() and (rgb == ref["rgb8"]).all()
    # the ground (black: attenuation 0, so every deeper term is 0 x inf = NaN -> 0) shows its emission + direct light only
    ground = (ref["prim_id"] <= 1) & (ref["prim_id"] >= 0)
    assert ground.any() and np.isfinite(ref["radiance"][ground]).all()
