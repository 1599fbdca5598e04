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


@pytest.mark.parametrize("case", ["far", "axis", "huge_scene", "tiny_scene"])
def test_conservative_fp32_boxes_at_the_extremes(need_gpu, case):
    """The FP32 slab test (origin margin, 1/d by v_rcp_f32, rounded interval end) must stay a superset of the reference's FP64 test
    where its error terms are largest: a camera 20,000 units away looking through a 0.05-degree lens, rays exactly along an
    axis (direction components 0 and 1e-17-ish), a scene 500 times larger around coordinates of 1e6 (one FP32 ulp there is 0.25),
    and a scene a hundredth of the size (distances near the lower end 0.001 of the accepted interval)."""
    spec = scenes.c3_blob988(width=96, height=64, spp=2)
    spec.max_bounces = 3
    cam = dict(spec.camera)
    if case == "far":
        look_from = np.array(cam["lookFrom"], dtype=float)
        look_at = np.array(cam["lookAt"], dtype=float)
        d = (look_from - look_at) / np.linalg.norm(look_from - look_at)
        cam["lookFrom"] = list(look_at + 9000.0 * d)       # (beyond tmax = 10000 nothing is hit at all)
        cam["vfov"] = 0.05
    elif case == "axis":
        cam["lookFrom"] = [0.0, 1.0, 6.0]; cam["lookAt"] = [0.0, 1.0, 0.0]; cam["vfov"] = 1e-6    # every primary ray is (0, 0, -1) to 1e-8
        spec.antialiasing = False
    elif case in ("huge_scene", "tiny_scene"):
        s = 500.0 if case == "huge_scene" else 0.01        # (hits are accepted in [0.001, 10000] only: the scene has to stay inside)
        off = np.array([3.0e6, -2.0e6, 1.0e6]) if case == "huge_scene" else np.zeros(3)
        for m in spec.obj_meshes:                        # (c3: all geometry is in the OBJ meshes, no config objects)
            m.verts = np.asarray(m.verts, dtype=float) * s
            m.position = tuple(np.asarray(m.position, dtype=float) * s + off)
        assert not spec.config_objects
        cam["lookFrom"] = list(np.asarray(cam["lookFrom"], dtype=float) * s + off)
        cam["lookAt"] = list(np.asarray(cam["lookAt"], dtype=float) * s + off)
        if "focusDist" in cam:
            cam["focusDist"] = cam["focusDist"] * s
        for L in spec.lights:
            if L["tag"] == "PointLight":
                L["position"] = list(np.asarray(L["position"], dtype=float) * s + off)
                L["intensity"] = list(np.asarray(L["intensity"], dtype=float) * s * s)
    spec.camera = cam
    check(*both(spec))


@pytest.mark.parametrize("box_precision", [0, 1, 2])
def test_nan_bounce_terms_follow_ghc_min_max(need_gpu, box_precision):
    """Haskell's max / min on a NaN (Trace.hs:91-94 with GHC's Ord Double defaults): a NaN bounce term is clamped to 0, a path that
    never hit anything is returned unclamped (NaN / inf primary misses stay NaN / inf).  Kernel = oracle, NaN positions included."""
    import numpy as np
    from nan_cases import nan_scene
    from oracle import oracle as O
    from raytracer_amd import api
    spec = nan_scene()
    o = O.OracleScene(spec)
    ref = o.render()
    o.close()
    g = api.GpuScene(spec)
    res, _ = g.render(box_precision=box_precision)
    g.close()
    assert np.isnan(ref["radiance"]).any() and np.isinf(ref["radiance"]).any()
    assert (res["prim_id"] == ref["prim_id"]).all()
    assert np.array_equal(np.isnan(res["radiance"]), np.isnan(ref["radiance"]))
    assert np.array_equal(np.isinf(res["radiance"]), np.isinf(ref["radiance"]))
    ok = np.isfinite(ref["radiance"])
    assert np.abs(res["radiance"][ok] - ref["radiance"][ok]).max() <= 1e-12
    assert (res["rgb8"] == ref["rgb8"]).all()
