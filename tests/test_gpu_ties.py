"""GPU parity where the traversal ORDER decides the result: primitives hit at exactly the same distance.

The reference keeps the primitive it tests last (`t <= closest`, HitHelpers.hs:14-16; the child visited second
wins an equal t, BVH.hs:137-141).  The render core traverses its own tree (rt_fastbvh.h) in another order and
settles such ties with the reference tree (tie_candidate_wins, rt_render_kernel.h).  These scenes are made of
exact duplicates with DIFFERENT emissive materials, so a wrong winner changes both the primitive id and the
radiance.  leafThreshold 1..4 puts the duplicates into different reference leaves (decided by the FP64 entry
distances at their lowest common ancestor) or into the same one (the later element wins).
The oracle follows the reference's recursion literally, so it is the ground truth here; bar: ids equal on every
pixel, radiance within 1e-12 (far inside the 1e-4 of north_star).
"""
import numpy as np
import pytest

from oracle import oracle as O
from raytracer_amd import api, scenes

pytestmark = pytest.mark.gpu


def duplicate_scene(n_tris=160, copies=3, leaf_threshold=1, seed=7, width=160, height=96, spp=2, max_bounces=3,
                    shuffle=False):
    rng = np.random.default_rng(seed)
    ctr = rng.uniform([-2.5, 0.2, -2.5], [2.5, 3.0, 2.5], size=(n_tris, 3))
    tri = ctr[:, None, :] + rng.uniform(-0.9, 0.9, size=(n_tris, 3, 3))
    verts = tri.reshape(-1, 3)
    base = np.arange(3 * n_tris).reshape(n_tris, 3)
    faces = np.concatenate([base] * copies)
    mats = sum(([f"m{c}"] * n_tris for c in range(copies)), [])
    if shuffle:                                   # duplicates far apart in the object list (still adjacent after the stable sort)
        perm = rng.permutation(len(faces))
        faces = faces[perm]
        mats = [mats[i] for i in perm]
    materials = [(f"m{c}", {"diffuseColor": [0.3 + 0.2 * c, 0.8 - 0.25 * c, 0.4],
                            "emissionColor": [0.15 * (c + 1), 0.05 + 0.1 * c, 0.3 - 0.1 * c]}) for c in range(copies)]
    floor = [("triangle", [-6.0, 0.0, -6.0], [-6.0, 0.0, 6.0], [6.0, 0.0, 6.0], "m0"),     # config triangles are listed twice
             ("triangle", [-6.0, 0.0, -6.0], [6.0, 0.0, 6.0], [6.0, 0.0, -6.0], "m1")]     # by the reference itself
    return scenes.SceneSpec(
        name=f"dups_{leaf_threshold}", width=width, height=height, spp=spp, antialiasing=True, gamma=1.0, exposure=1.0,
        background={"tag": "Gradient", "color1": [1.0, 1.0, 1.0], "color2": [0.5, 0.7, 1.0]},
        camera={"lookFrom": [-6.0, 3.0, -6.0], "lookAt": [0.0, 1.2, 0.0], "vUp": [0.0, 1.0, 0.0], "vfov": 55.0,
                "aperture": 0.0, "focusDist": 3.0},
        max_bounces=max_bounces, bvh_max_depth=32, leaf_threshold=leaf_threshold, materials=materials,
        lights=[{"tag": "PointLight", "position": [4.0, 9.0, -3.0], "intensity": [30.0, 30.0, 28.0]}],
        config_objects=floor, obj_meshes=[scenes.ObjMesh(verts, faces, mats, (0.0, 0.0, 0.0))])


def both(spec, **over):
    g = api.GpuScene(spec)
    res, _ = g.render(**over)
    exact, _ = g.render(box_precision=1, **over)
    g.close()
    o = O.OracleScene(spec)
    ref = o.render(**over)
    o.close()
    return res, exact, ref


@pytest.mark.parametrize("leaf_threshold", [1, 2, 3, 4])
@pytest.mark.parametrize("shuffle", [False, True])
def test_duplicates_resolve_like_the_reference(need_gpu, leaf_threshold, shuffle):
    spec = duplicate_scene(leaf_threshold=leaf_threshold, shuffle=shuffle)
    res, exact, ref = both(spec)
    assert (ref["prim_id"] >= 0).mean() > 0.5                     # the frame is mostly geometry, i.e. mostly ties
    assert (exact["prim_id"] == ref["prim_id"]).all()
    assert (res["prim_id"] == ref["prim_id"]).all(), f"{(res['prim_id'] != ref['prim_id']).sum()} tie(s) resolved differently"
    assert np.abs(res["radiance"] - ref["radiance"]).max() <= 1e-12
    assert np.abs(exact["radiance"] - ref["radiance"]).max() <= 1e-12


def test_duplicates_deterministic_centre_rays(need_gpu):
    # antialiasing off: every sample of a pixel is the same centre ray; maxBounces 1 makes the frame deterministic
    spec = duplicate_scene(leaf_threshold=1, copies=4, width=128, height=72, spp=1, max_bounces=1)
    spec.antialiasing = False
    res, exact, ref = both(spec)
    assert (res["prim_id"] == ref["prim_id"]).all()
    assert np.abs(res["radiance"] - ref["radiance"]).max() <= 1e-12


def test_winner_depends_on_the_ray(need_gpu):
    """The same pair of duplicates, in different reference leaves, must be able to resolve BOTH ways depending on
    the ray (which child box is entered first): otherwise the test would not exercise the entry-distance rule."""
    spec = duplicate_scene(leaf_threshold=1, copies=2, n_tris=200, width=192, height=108, spp=1, max_bounces=1)
    spec.antialiasing = False
    res, _, ref = both(spec)
    assert (res["prim_id"] == ref["prim_id"]).all()
    ids = ref["prim_id"][ref["prim_id"] >= 0]
    n = 200
    mesh = ids[(ids >= 2) & (ids < 2 + 2 * n)] - 2              # allObjects = 2 config triangles ++ mesh ++ config objects
    first, second = np.unique(mesh[mesh < n]), np.unique(mesh[mesh >= n] - n)
    assert len(first) > 0 and len(second) > 0                    # both copies win somewhere in the frame
