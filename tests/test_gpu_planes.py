"""GPU parity for planes outside their constant slab box.

Where a plane is crossed beyond its leaf box's exit the reference's answer depends on its traversal history (the leaf
must be the child visited second and its sibling must have produced a hit, BVH.hs:137-138); the kernel answers those
queries with literal_hit, the literal walk of the reference's tree, in all three box_precision modes.  Bar: primitive
ids equal on every pixel, radiance within 1e-12 of the oracle."""
import numpy as np
import pytest

from oracle import oracle as O
from plane_cases import random_plane_scene, two_object_scene
from raytracer_amd import api

pytestmark = pytest.mark.gpu


def oracle_frame(spec):
    o = O.OracleScene(spec)
    ref = o.render()
    o.close()
    return ref


@pytest.mark.parametrize("box_precision", [0, 1, 2])
def test_two_object_scene(need_gpu, box_precision):
    for spec in (two_object_scene(), two_object_scene(width=80, height=45, spp=3, max_bounces=4, antialiasing=True)):
        ref = oracle_frame(spec)
        g = api.GpuScene(spec)
        res, st = g.render(box_precision=box_precision, collect_stats=1)
        g.close()
        assert (ref["prim_id"] == 1).sum() > 100                  # the oracle sees the plane (beyond its slab) ...
        assert st.rays_literal > 0                                # ... and the kernel had to ask the literal walk
        assert (res["prim_id"] == ref["prim_id"]).all(), f"{(res['prim_id'] != ref['prim_id']).sum()} ids differ"
        assert np.abs(res["radiance"] - ref["radiance"]).max() <= 1e-12
        assert (res["rgb8"] == ref["rgb8"]).all()


@pytest.mark.parametrize("seed", [1, 2, 3, 4, 5, 6])
def test_random_planes(need_gpu, seed):
    spec = random_plane_scene(seed)
    ref = oracle_frame(spec)
    g = api.GpuScene(spec)
    lit = 0
    for mode in (0, 1, 2):
        res, st = g.render(box_precision=mode, collect_stats=1)
        lit += st.rays_literal
        assert (res["prim_id"] == ref["prim_id"]).all(), f"mode {mode}: {(res['prim_id'] != ref['prim_id']).sum()} ids differ"
        assert np.abs(res["radiance"] - ref["radiance"]).max() <= 1e-12, mode
    g.close()
    assert lit > 0
