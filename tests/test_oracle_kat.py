"""Known-answer tests that pin the CPU oracle to the reference's semantics.

The reference has no tests and no golden vectors ("parity unpinned"), so every
expected value here is derived BY HAND from the cited Haskell lines (paths relative to
/root/reference/app).  Philox vectors are the published Random123 known-answer values.
"""
import math

import numpy as np
import pytest

from oracle import oracle as O
from raytracer_amd import scenes
from raytracer_amd.scenes import SceneSpec

TRI, SPH, PLN = 0, 1, 2
UNIT_TRI = [0, 0, 0, 1, 0, 0, 0, 1, 0]


def test_philox4x32_10_known_answers():
    # Random123 kat_vectors: philox4x32 10 rounds
    assert O.philox([0, 0, 0, 0], [0, 0]) == [0x6627E8D5, 0xE169C58D, 0xBC57AC4C, 0x9B00DBD8]
    assert O.philox([0xFFFFFFFF] * 4, [0xFFFFFFFF] * 2) == [0x408F276D, 0x41C83B0E, 0xA20BC7C6, 0x6D5451FD]
    assert O.philox([0x243F6A88, 0x85A308D3, 0x13198A2E, 0x03707344], [0xA4093822, 0x299F31D0]) == \
        [0xD16CFE09, 0x94FDCCEB, 0x5001E420, 0x24126EA1]


# ---------------------------------------------------------------- Hittable/BoundingBox.hs:18-37
def test_aabb_interval_clipping_and_parallel_rule():
    lo, hi = [-1, -1, -1], [1, 1, 1]
    # +x ray through the centre: y,z are parallel axes (|d| < 1e-8) with the origin inside -> (-inf, inf)
    assert O.hit_aabb(lo, hi, [-5, 0, 0], [1, 0, 0], 0.001, 1e4) == (4.0, 6.0)
    # interval narrowing: tMin/tMax take part in tEnter/tExit (:30-31)
    assert O.hit_aabb(lo, hi, [-5, 0, 0], [1, 0, 0], 4.5, 5.0) == (4.5, 5.0)
    assert O.hit_aabb(lo, hi, [-5, 0, 0], [1, 0, 0], 0.001, 3.9) is None
    # parallel axis with the origin outside the slab -> Nothing (:22); on the face -> still inside
    assert O.hit_aabb(lo, hi, [-5, 2, 0], [1, 0, 0], 0.001, 1e4) is None
    assert O.hit_aabb(lo, hi, [-5, 1, 0], [1, 0, 0], 0.001, 1e4) == (4.0, 6.0)
    # |d| = 1e-8 is NOT parallel (strict <): the slab is computed, here it excludes the ray
    assert O.hit_aabb(lo, hi, [-5, 2, 0], [1, 1e-8, 0], 0.001, 1e4) is None
    # negative direction swaps t0/t1 (:27)
    assert O.hit_aabb(lo, hi, [5, 0, 0], [-1, 0, 0], 0.001, 1e4) == (4.0, 6.0)
    # tEnter == tExit still hits (<=, :32): a ray touching the corner edge
    r = O.hit_aabb(lo, hi, [-2, 0, 0], [1, 1, 0], 0.0, 1e4)
    assert r == (1.0, 1.0)
    # box behind the ray
    assert O.hit_aabb(lo, hi, [5, 0, 0], [1, 0, 0], 0.001, 1e4) is None


def test_primitive_boxes():
    lo, hi = O.prim_box(TRI, UNIT_TRI)            # padded by epsilon = 1e-8 (Triangle.hs:24-28)
    assert np.array_equal(lo, [-1e-8, -1e-8, -1e-8]) and np.array_equal(hi, [1 + 1e-8, 1 + 1e-8, 1e-8])
    lo, hi = O.prim_box(SPH, [1, 2, 3, 0.5, 0, 0, 0, 0, 0])   # exact (Sphere.hs:20-22)
    assert np.array_equal(lo, [0.5, 1.5, 2.5]) and np.array_equal(hi, [1.5, 2.5, 3.5])
    lo, hi = O.prim_box(PLN, [7, 7, 7, 1, 0, 0, 0, 0, 0])     # the constant slab (Plane.hs:20)
    assert np.array_equal(lo, [-1e6, -1e-3, -1e6]) and np.array_equal(hi, [1e6, 1e-3, 1e6])


# ---------------------------------------------------------------- Hittable/Objects/Triangle.hs:30-64
def test_triangle_moller_trumbore():
    h = O.hit_prim(TRI, UNIT_TRI, [0.25, 0.25, 1], [0, 0, -1])
    assert h["t"] == 1.0 and np.array_equal(h["point"], [0.25, 0.25, 0]) and np.array_equal(h["normal"], [0, 0, 1]) and h["front"]
    h = O.hit_prim(TRI, UNIT_TRI, [0.25, 0.25, -1], [0, 0, 1])      # two-sided: normal flipped against the ray, frontFace False
    assert h["t"] == 1.0 and np.array_equal(h["normal"], [0, 0, -1]) and not h["front"]
    assert O.hit_prim(TRI, UNIT_TRI, [0.0, 0.5, 1], [0, 0, -1]) is not None      # u = 0 is inside (u < 0 rejects)
    assert O.hit_prim(TRI, UNIT_TRI, [0.5, 0.5, 1], [0, 0, -1]) is not None      # u + v = 1 is inside (> 1 rejects)
    assert O.hit_prim(TRI, UNIT_TRI, [0.6, 0.6, 1], [0, 0, -1]) is None
    assert O.hit_prim(TRI, UNIT_TRI, [-0.1, 0.5, 1], [0, 0, -1]) is None
    assert O.hit_prim(TRI, UNIT_TRI, [0.25, 0.25, 1], [1, 0, 0]) is None         # |a| < 1e-8: parallel
    assert O.hit_prim(TRI, UNIT_TRI, [0.25, 0.25, 1], [0, 0, -1], 0.001, 0.5) is None   # t outside the interval
    assert O.hit_prim(TRI, UNIT_TRI, [0.25, 0.25, 1], [0, 0, -1], 1.0, 1.0) is not None  # closed interval
    # un-normalised direction: t scales, the point does not
    h = O.hit_prim(TRI, UNIT_TRI, [0.25, 0.25, 1], [0, 0, -4])
    assert h["t"] == 0.25 and np.array_equal(h["point"], [0.25, 0.25, 0])


# ---------------------------------------------------------------- Hittable/Objects/Sphere.hs:24-40
def test_sphere_near_root_only():
    s = [0, 0, 0, 1, 0, 0, 0, 0, 0]
    h = O.hit_prim(SPH, s, [0, 0, 5], [0, 0, -1])
    assert h["t"] == 4.0 and np.array_equal(h["normal"], [0, 0, 1]) and h["front"]
    assert O.hit_prim(SPH, s, [0, 0, 0], [0, 0, -1]) is None           # from inside: t1 = -1, the far root is never tried
    assert O.hit_prim(SPH, s, [0, 0, 1], [0, 0, -1]) is None           # from the surface: t1 = 0 < 0.001
    assert O.hit_prim(SPH, s, [0, 2, 5], [0, 0, -1]) is None           # discriminant < 0
    h = O.hit_prim(SPH, s, [0, 0, 5], [0, 0, -2])                      # a = |d|^2 = 4
    assert h["t"] == 2.0 and np.array_equal(h["point"], [0, 0, 1])
    h = O.hit_prim(SPH, [0, 0, 0, 2, 0, 0, 0, 0, 0], [0, 0, 5], [0, 0, -1])   # normal = (p - c)/r
    assert np.array_equal(h["normal"], [0, 0, 1])


# ---------------------------------------------------------------- Hittable/Objects/Plane.hs:21-32
def test_plane():
    p = [0, 0, 0, 0, 1, 0, 0, 0, 0]
    h = O.hit_prim(PLN, p, [0, 1, 0], [0, -1, 0])
    assert h["t"] == 1.0 and np.array_equal(h["normal"], [0, 1, 0]) and h["front"]
    h = O.hit_prim(PLN, p, [0, -1, 0], [0, 1, 0])
    assert h["t"] == 1.0 and np.array_equal(h["normal"], [0, -1, 0]) and not h["front"]
    assert O.hit_prim(PLN, p, [0, 1, 0], [1, 0, 0]) is None            # |denom| <= 1e-6
    assert O.hit_prim(PLN, p, [0, 1, 0], [1, -1e-6, 0]) is None        # not strictly greater than 1e-6
    assert O.hit_prim(PLN, [0, 0, 0, 0, 2, 0, 0, 0, 0], [0, 1, 0], [0, -1, 0])["normal"][1] == 2.0   # normal is NOT normalised


def _spec(objs, materials=(), lights=(), bg=None, leaf=4, depth=32, maxb=1):
    return SceneSpec(name="kat", width=4, height=2, spp=1, antialiasing=False, gamma=1.0, exposure=1.0,
                     background=bg or {"tag": "SolidColor", "color": [2.0, 2.0, 2.0]},
                     camera={"lookFrom": [0, 0, 0], "lookAt": [0, 0, -1], "vUp": [0, 1, 0], "vfov": 90.0, "aperture": 0.0, "focusDist": 1.0},
                     max_bounces=maxb, bvh_max_depth=depth, leaf_threshold=leaf, materials=list(materials), lights=list(lights),
                     config_objects=list(objs))


# ---------------------------------------------------------------- ties: HitHelpers.hs:11-16, BVH.hs:128-143
def test_tie_breaking_in_leaf_and_across_children():
    tri = ("triangle", [0, 0, 0], [1, 0, 0], [0, 1, 0], None)
    # allObjects = configTriangles ++ hittables: the config triangle appears TWICE (SceneParser.hs:67-73): ids 0 and 1
    one_leaf = O.OracleScene(_spec([tri], leaf=4))
    assert one_leaf.bvh_info()["n_nodes"] == 1 and one_leaf.n_objs == 2
    pid, out = one_leaf.hit([0.25, 0.25, 1], [0, 0, -1])
    assert pid == 1 and out[0] == 1.0            # equal t inside a leaf: the LATER list element wins
    two_leaves = O.OracleScene(_spec([tri], leaf=1))
    assert two_leaves.bvh_info()["n_nodes"] == 3
    pid, _ = two_leaves.hit([0.25, 0.25, 1], [0, 0, -1])
    assert pid == 0                              # lEnter == rEnter -> right first; the second (left) child wins the tie


def test_root_box_is_never_tested_and_planes_live_in_leaf_boxes():
    # single-leaf root: a plane far outside its own (bogus) AABB is still hit, because no box is tested (BVH.hs:121-123)
    pl = ("plane", [0, 50, 0], [0, 1, 0], None)
    s = O.OracleScene(_spec([pl], leaf=4))
    pid, out = s.hit([0, 60, 0], [0, -1, 0])
    assert pid == 0 and out[0] == 10.0
    # once the plane sits in a child leaf, it is only visible where the hit lies inside that leaf's box
    sph = ("sphere", [0, 0, -5], 1.0, None)
    s2 = O.OracleScene(_spec([pl, sph], leaf=1))
    pid, _ = s2.hit([0, 60, 0], [0, -1, 0])
    assert pid == -1


# ---------------------------------------------------------------- Trace.hs:85-105, Vec3.hs:98-107
def test_schlick_reflect_refract_tonemap():
    assert O.schlick(1.0, 1.5) == pytest.approx(0.04, abs=1e-15)
    assert O.schlick(0.0, 1.5) == pytest.approx(1.0, abs=1e-15)
    assert np.array_equal(O.reflect([1, -1, 0], [0, 1, 0]), [1, 1, 0])
    assert np.allclose(O.refract([0, -1, 0], [0, 1, 0], 1 / 1.5), [0, -1, 0], atol=1e-15)
    # refract with the un-clamped cos and the abs() under the root (Vec3.hs:104-106)
    v = np.array([math.sin(0.3), -math.cos(0.3), 0.0])
    r = O.refract(v, [0, 1, 0], 1 / 1.5)
    assert r[0] == pytest.approx(math.sin(0.3) / 1.5, abs=1e-15)
    assert r[1] == pytest.approx(-math.sqrt(1 - (math.sin(0.3) / 1.5) ** 2), abs=1e-15)
    c = O.tonemap_gamma(1.5, 0.8, [1.0, 0.0, 4.0])
    assert c[0] == pytest.approx(0.6 ** 0.8, abs=1e-15) and c[1] == 0.0 and c[2] == pytest.approx((6 / 7) ** 0.8, abs=1e-15)


# ---------------------------------------------------------------- Camera.hs:26-60
def test_camera_and_row_orientation():
    cam = O.default_camera({"lookFrom": [0, 0, 0], "lookAt": [0, 0, -1], "vUp": [0, 1, 0], "vfov": 90.0, "aperture": 0.0, "focusDist": 1.0}, 2.0)
    assert np.allclose(cam.w[:], [0, 0, 1]) and np.allclose(cam.u[:], [1, 0, 0]) and np.allclose(cam.v[:], [0, 1, 0])
    assert np.allclose(cam.horizontal[:], [4, 0, 0], atol=1e-15) and np.allclose(cam.vertical[:], [0, 2, 0], atol=1e-15)
    assert np.allclose(cam.lower_left[:], [-2, -1, -1], atol=1e-15)
    o, d = O.generate_ray(cam, 0, 0, 4, 2, False, 0, 0, 0)      # bottom-left pixel centre: u = .125, v = .25
    e = np.array([-1.5, -0.5, -1.0])
    assert np.allclose(o, 0) and np.allclose(d, e / np.linalg.norm(e), atol=1e-15)
    # antialiasing draws offsets in (0,1): the ray stays inside its pixel
    o, d = O.generate_ray(cam, 3, 1, 4, 2, True, 7, 5, 9)
    p = d / -d[2]
    assert 1.0 <= p[0] <= 2.0 and 0.0 <= p[1] <= 1.0
    # file row 0 is the TOP row (rowIdx = H-1-j, RenderWorker.hs:66): a gradient background is brighter at the top
    spec = _spec([("sphere", [0, 0, 50], 1.0, None)], bg={"tag": "Gradient", "color1": [1, 1, 1], "color2": [0, 0, 0]})
    img = O.OracleScene(spec).render(n_threads=1)["radiance"]
    assert img[0, 0, 0] > img[1, 0, 0]


def test_background_gradient_and_sky_addressing():
    spec = _spec([("sphere", [0, 0, 50], 1.0, None)], bg={"tag": "Gradient", "color1": [1, 0, 0], "color2": [0, 0, 1]})
    s = O.OracleScene(spec)
    assert np.array_equal(s.background([0, 1, 0]), [1, 0, 0])       # color1 at the zenith (lerp t c2 c1, t = 1)
    assert np.array_equal(s.background([0, -3, 0]), [0, 0, 1])      # direction is normalised first
    assert np.allclose(s.background([1, 0, 0]), [0.5, 0, 0.5])
    sky = np.zeros((4, 8, 3), np.float32)
    sky[..., 0] = np.arange(8)[None, :]
    sky[..., 1] = np.arange(4)[:, None]
    spec.sky = ("rgbf", sky)
    s = O.OracleScene(spec)
    assert np.array_equal(s.background([1, 0, 0])[:2], [4, 2])      # u = v = 0.5 -> texel (4, 2)
    assert np.array_equal(s.background([0, 1, 0])[1:2], [0])        # straight up -> top row
    assert np.array_equal(s.background([0, -1, 0])[1:2], [3])       # straight down -> clamped to the last row
    assert s.background([-1, 0, -1e-12])[0] == 0                    # atan2 -> -pi: u = 0
    assert s.background([-1, 0, 1e-12])[0] == 7                     # atan2 -> +pi: u = 1 -> clamped to w-1
    spec.sky = ("rgb8", (sky * 10).astype(np.uint8))
    assert O.OracleScene(spec).background([1, 0, 0])[0] == pytest.approx(40 / 255)


# ---------------------------------------------------------------- Trace.hs:22-76, Light.hs:16-49
def test_trace_is_additive_with_nested_clamp_and_ray0_at_depth_zero():
    mats = [("m", {"diffuseColor": [0.5, 0.5, 0.5], "emissionColor": [0.1, 0.2, 0.3]})]
    sph = ("sphere", [0, 0, -5], 1.0, "m")
    o, d = [0, 0, 0], [0, 0, -1]
    s = O.OracleScene(_spec([sph], mats, bg={"tag": "SolidColor", "color": [2, 2, 2]}))
    assert np.allclose(s.trace_path(o, d, 1), [1.1, 1.2, 1.3], atol=1e-15)   # e + 0 + clamp(0.5 * bg)
    assert np.array_equal(s.trace_path(o, d, 0), [2, 2, 2])                   # depth 0: background, attenuation 1
    assert np.array_equal(s.trace_path([0, 0, 0], [0, 1, 0], 3), [2, 2, 2])   # primary miss: background, un-clamped
    s = O.OracleScene(_spec([sph], mats, bg={"tag": "SolidColor", "color": [30, 30, 30]}))
    assert np.allclose(s.trace_path(o, d, 1), [10.1, 10.2, 10.3], atol=1e-15)  # the bounce term is clamped to [0,10] ...
    assert np.array_equal(s.trace_path([0, 0, 0], [0, 1, 0], 3), [30, 30, 30])  # ... the primary miss is not
    # depth exhausted -> background along ray0 (the PRIMARY ray), not along the bounce ray (Trace.hs:22)
    s = O.OracleScene(_spec([sph], mats, bg={"tag": "Gradient", "color1": [1, 1, 1], "color2": [0, 0, 0]}))
    r = s.trace_path(o, d, 1, seed=3, pixel=11, sample=2)
    assert np.allclose(r, np.array([0.1, 0.2, 0.3]) + 0.5 * 0.5, atol=1e-15)    # t = 0.5 for d = (0,0,-1), whatever the bounce
    # glass keeps the attenuation (Trace.hs:69-71)
    glass = [("g", {"diffuseColor": [0.5, 0.5, 0.5], "ior": 1.5, "transmission": 1.0})]
    s = O.OracleScene(_spec([("sphere", [0, 0, -5], 1.0, "g")], glass, bg={"tag": "SolidColor", "color": [2, 2, 2]}))
    assert np.array_equal(s.trace_path(o, d, 1), [2, 2, 2])
    # unknown material id -> defaultMaterial: white diffuse (Trace.hs:28)
    s = O.OracleScene(_spec([("sphere", [0, 0, -5], 1.0, "nope")], bg={"tag": "SolidColor", "color": [2, 2, 2]}))
    assert np.array_equal(s.trace_path(o, d, 1), [2, 2, 2])


def test_direct_lighting_and_shadow_rays():
    floor = [("triangle", [-10, 0, -10], [-10, 0, 10], [10, 0, 10], None), ("triangle", [-10, 0, -10], [10, 0, 10], [10, 0, -10], None)]
    bg = {"tag": "SolidColor", "color": [0, 0, 0]}
    pl = {"tag": "PointLight", "position": [0, 2, 0], "intensity": [4, 4, 4]}
    o, d = [0, 1, 0], [0, -1, 0]
    s = O.OracleScene(_spec(floor, lights=[pl], bg=bg))
    assert np.allclose(s.trace_path(o, d, 1), [1, 1, 1], atol=1e-15)            # max(0, n.l)/d^2 * I = 1/4 * 4
    dl = {"tag": "DirectionalLight", "direction": [0, -1, 0], "intensity": [0.25, 0.5, 1]}
    s = O.OracleScene(_spec(floor, lights=[pl, dl], bg=bg))
    assert np.allclose(s.trace_path(o, d, 1), [1.25, 1.5, 2], atol=1e-15)       # lights add (Light.hs:21)
    blocker = ("sphere", [0, 1.0, 0], 0.25, None)                               # between the hit point and the point light
    s = O.OracleScene(_spec(floor + [blocker], lights=[pl], bg=bg))
    assert np.array_equal(s.trace_path([3, 1, 0], [-3, -1, 0], 1), [0, 0, 0])   # shadowed
    # shadow interval ends 0.01 before the light (Light.hs:30): a blocker just behind the light does not shadow
    behind = ("sphere", [0, 2.6, 0], 0.5, None)
    s = O.OracleScene(_spec(floor + [behind], lights=[pl], bg=bg))
    assert np.allclose(s.trace_path(o, [1e-3, -1, 0], 1), 4 / 4, atol=1e-5)
    # lighting is applied to the material's diffuse colour, never to the per-object colour (Trace.hs:30,36)
    s = O.OracleScene(_spec(floor, [("red", {"diffuseColor": [1, 0, 0]})], lights=[pl], bg=bg))
    assert np.allclose(s.trace_path(o, d, 1), [1, 1, 1], atol=1e-15)            # floor has no material name -> default white


def test_quantisation_truncates():
    spec = _spec([("sphere", [0, 0, 50], 1.0, None)], bg={"tag": "SolidColor", "color": [1.0, 0.5, 0.0]})
    res = O.OracleScene(spec).render(n_threads=1)
    # toneMap(1): c/(c+1) -> .5, 1/3, 0 ; gamma 1 ; truncate(c*255.999) (RenderWorker.hs:108-109)
    assert res["rgb8"][0, 0].tolist() == [127, 85, 0]
    assert np.allclose(res["color"][0, 0], [0.5, 1 / 3, 0.0])


def test_plane_seen_through_the_second_childs_interval():
    """sphere c=(0,-9,-6) r=10 (first child, hit at 6.0517) + plane p=(0,-0.5,0), leafThreshold 1.  The plane's
    leaf box (the constant slab, Plane.hs:20) is left at 4.4744, but as the SECOND child it receives [enter, t firstHit]
    (BVH.hs:137-138), so the reference returns the plane at t = 5.5902 (= 2.5 / (0.5/sqrt(1.25)))."""
    from plane_cases import two_object_scene
    o = O.OracleScene(two_object_scene())
    d = np.array([0.0, -0.5, -1.0]); d /= np.linalg.norm(d)
    pid, out = o.hit([0.0, 2.0, 5.0], d)
    assert pid == 1 and abs(out[0] - 2.5 * np.sqrt(1.25) / 0.5) < 1e-12
    # a ray that misses the sphere: nothing narrows the interval past the slab's exit, the plane stays invisible
    d2 = np.array([0.9, -0.5, 0.3]); d2 /= np.linalg.norm(d2)
    pid2, _ = o.hit([0.0, 2.0, 5.0], d2)
    assert pid2 == -1
    o.close()
