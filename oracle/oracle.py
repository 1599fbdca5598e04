"""ctypes wrapper of oracle/_build/liboracle.so — TEST INFRASTRUCTURE ONLY.

May be imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg, never by
raytracer_amd/.  It reuses raytracer_amd's ctypes struct definitions and SceneSpec (data classes of
the public ABI) and nothing else: the scene -> struct marshalling below is the checker's OWN reading
of the reference's set-up code, written separately from raytracer_amd/api.py, so that a slip in the
product's marshalling shows up as a GPU-vs-oracle difference instead of cancelling out.
"""
from __future__ import annotations

import time
import ctypes as C
import os

import numpy as np

from raytracer_amd import _ffi as F          # only the ctypes struct classes of the public ABI (layout pinned by tests/test_haskell_offsets.py)
from raytracer_amd.scenes import SceneSpec

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


class orc_counters(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in
                ("rays_closest", "rays_shadow", "nodes_literal", "nodes_nonempty", "tri_tests", "sphere_tests",
                 "plane_tests", "tri_tests_nonempty", "sphere_tests_nonempty", "plane_tests_nonempty", "shaded_hits")]

    def as_dict(self):
        return {n: getattr(self, n) for n, _ in self._fields_}


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "_build", "liboracle.so")
        if not os.path.exists(path):
            raise OSError(f"{path} missing: run `make oracle`")
        L = C.CDLL(path)
        dp, ip = F.c_double_p, F.c_int32_p
        L.orc_scene_create.restype = C.c_void_p
        L.orc_scene_create.argtypes = [C.c_int32, ip, dp, ip, C.c_int32, C.c_int32, C.c_int32, C.POINTER(F.rt_material),
                                       C.c_int32, C.POINTER(F.rt_light), C.c_int32, dp, dp, C.c_int32, C.c_int32, C.c_int32, C.c_void_p]
        L.orc_scene_destroy.argtypes = [C.c_void_p]
        L.orc_bvh_info.argtypes = [C.c_void_p, ip, ip, ip, C.POINTER(C.c_int64)]
        L.orc_bvh_export.argtypes = [C.c_void_p, dp, ip, ip, dp, ip, ip]
        L.orc_default_camera.argtypes = [dp, dp, dp, C.c_double, C.c_double, C.c_double, C.c_double, C.POINTER(F.rt_camera)]
        L.orc_render.restype = C.c_int
        L.orc_render.argtypes = [C.c_void_p, C.POINTER(F.rt_camera), C.POINTER(F.rt_params), C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                                 dp, dp, F.c_uint8_p, ip, C.POINTER(orc_counters)]
        L.orc_hit_aabb.restype = C.c_int
        L.orc_hit_aabb.argtypes = [dp, dp, dp, dp, C.c_double, C.c_double, dp]
        L.orc_hit_prim.restype = C.c_int
        L.orc_hit_prim.argtypes = [C.c_int32, dp, dp, dp, C.c_double, C.c_double, dp]
        L.orc_prim_box.argtypes = [C.c_int32, dp, dp]
        L.orc_hit_scene.restype = C.c_int
        L.orc_hit_scene.argtypes = [C.c_void_p, dp, dp, C.c_double, C.c_double, dp]
        L.orc_schlick.restype = C.c_double
        L.orc_schlick.argtypes = [C.c_double, C.c_double]
        L.orc_reflect.argtypes = [dp, dp, dp]
        L.orc_refract.argtypes = [dp, dp, C.c_double, dp]
        L.orc_tonemap_gamma.argtypes = [C.c_double, C.c_double, dp, dp]
        L.orc_background.argtypes = [C.c_void_p, dp, dp]
        L.orc_generate_ray.argtypes = [C.POINTER(F.rt_camera), C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                                       C.c_uint64, C.c_uint32, C.c_uint32, dp]
        L.orc_philox.argtypes = [C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
        L.orc_trace_path.argtypes = [C.c_void_p, dp, dp, C.c_int32, C.c_uint64, C.c_uint32, C.c_uint32, dp]
        _LIB = L
    return _LIB


def _d(a):
    return np.ascontiguousarray(a, np.float64)


def _dp(a):
    return a.ctypes.data_as(F.c_double_p)


def _ip(a):
    return a.ctypes.data_as(F.c_int32_p)


# ---- the checker's own scene -> struct marshalling (deliberately not shared with raytracer_amd/api.py)
def _set3(dst, src):
    dst[0], dst[1], dst[2] = float(src[0]), float(src[1]), float(src[2])


def oracle_materials(spec: SceneSpec):
    """`materials :: Maybe [(String, Material)]` -> ids 1.. in list order (assignMaterialIds, Parser/Material.hs:52-57);
    absent optional fields stay `Nothing` = no flag bit (Rendering/Material.hs:37-46: only diffuseColor has a default)."""
    table = (F.rt_material * max(1, len(spec.materials)))()
    for ident, (_name, fields) in enumerate(spec.materials, start=1):
        m = table[ident - 1]
        m.id, m.flags = ident, 0
        _set3(m.diffuse, fields["diffuseColor"] if "diffuseColor" in fields else (1.0, 1.0, 1.0))
        for key, bit, setter in (("emissionColor", F.RT_MAT_HAS_EMISSION, lambda v: _set3(m.emission, v)),
                                 ("shininess", F.RT_MAT_HAS_SHININESS, lambda v: setattr(m, "shininess", float(v))),
                                 ("ior", F.RT_MAT_HAS_IOR, lambda v: setattr(m, "ior", float(v))),
                                 ("transmission", F.RT_MAT_HAS_TRANSMISSION, lambda v: setattr(m, "transmission", float(v)))):
            if fields.get(key) is not None:
                m.flags |= bit
                setter(fields[key])
    return table, len(spec.materials)


def oracle_lights(spec: SceneSpec):
    """convertLight (Rendering/ImageGenerator.hs:137-139)."""
    table = (F.rt_light * max(1, len(spec.lights)))()
    for slot, light in zip(table, spec.lights):
        point = light["tag"] == "PointLight"
        if not point and light["tag"] != "DirectionalLight":
            raise ValueError(f"unknown light {light['tag']}")
        slot.kind = F.RT_LIGHT_POINT if point else F.RT_LIGHT_DIRECTIONAL
        _set3(slot.vec, light["position"] if point else light["direction"])
        _set3(slot.intensity, light["intensity"])
    return table, len(spec.lights)


def oracle_background(spec: SceneSpec):
    """BackgroundSettings (Config.hs:40-43): Gradient color1 color2 | SolidColor color."""
    tag = spec.background["tag"]
    if tag == "SolidColor":
        return F.RT_BG_SOLID, spec.background["color"], spec.background["color"]
    if tag != "Gradient":
        raise ValueError(f"unknown background {tag}")
    return F.RT_BG_GRADIENT, spec.background["color1"], spec.background["color2"]


def oracle_params(spec: SceneSpec, **over) -> F.rt_params:
    """ImageSettings + RaytracerSettings (Config.hs:29-37, 58-64) + the RNG key."""
    p = F.rt_params(width=spec.width, height=spec.height, samples_per_pixel=spec.spp, antialiasing=int(bool(spec.antialiasing)),
                    max_bounces=spec.max_bounces, rr_start_bounce=0, exposure=spec.exposure, gamma=spec.gamma, seed=spec.seed,
                    collect_stats=0, box_precision=0)
    for name, value in over.items():
        if not hasattr(p, name):
            raise AttributeError(name)
        setattr(p, name, value)
    return p


def default_camera(cam: dict, aspect: float) -> F.rt_camera:
    out = F.rt_camera()
    lf, la, up = _d(cam["lookFrom"]), _d(cam["lookAt"]), _d(cam["vUp"])
    lib().orc_default_camera(_dp(lf), _dp(la), _dp(up), float(cam["vfov"]), float(aspect), float(cam["aperture"]),
                             float(cam["focusDist"]), C.byref(out))
    return out


class OracleScene:
    def __init__(self, spec: SceneSpec):
        self.spec = spec
        kind, data, mat = spec.all_objects()
        self.n_objs = len(kind)
        mats, nm = oracle_materials(spec)
        lights, nl = oracle_lights(spec)
        bgk, c1, c2 = oracle_background(spec)
        c1, c2 = _d(c1), _d(c2)
        sky_fmt, sw, sh, sp = F.RT_SKY_NONE, 0, 0, None
        if spec.sky is not None:
            fmt, img = spec.sky
            self._sky = np.ascontiguousarray(img)
            sky_fmt = F.RT_SKY_RGB8 if fmt == "rgb8" else F.RT_SKY_RGBF
            sh, sw = self._sky.shape[0], self._sky.shape[1]
            sp = self._sky.ctypes.data
        self._h = lib().orc_scene_create(self.n_objs, _ip(kind), _dp(data), _ip(mat), spec.leaf_threshold, spec.bvh_max_depth,
                                         nm, mats, nl, lights, bgk, _dp(c1), _dp(c2), sky_fmt, sw, sh, sp)
        if not self._h:
            raise ValueError("oracle: empty scene (Cannot compute bounding box of empty list)")
        self.camera = default_camera(spec.camera, spec.width / spec.height)

    def bvh_info(self):
        nn, nl, dp, npr = C.c_int32(), C.c_int32(), C.c_int32(), C.c_int64()
        lib().orc_bvh_info(self._h, C.byref(nn), C.byref(nl), C.byref(dp), C.byref(npr))
        return {"n_nodes": nn.value, "n_leaves": nl.value, "depth": dp.value, "n_prims": npr.value}

    def bvh_export(self):
        info = self.bvh_info()
        nn, npr = info["n_nodes"], info["n_prims"]
        a = {"node_bounds": np.zeros((nn, 6)), "node_child": np.zeros((nn, 2), np.int32),
             "prim_kind": np.zeros(npr, np.int32), "prim_data": np.zeros((npr, 9)),
             "prim_material": np.zeros(npr, np.int32), "prim_id": np.zeros(npr, np.int32)}
        lib().orc_bvh_export(self._h, _dp(a["node_bounds"]), _ip(a["node_child"]), _ip(a["prim_kind"]), _dp(a["prim_data"]),
                             _ip(a["prim_material"]), _ip(a["prim_id"]))
        return a

    def render(self, rows=None, n_threads=8, counters=False, camera=None, want_ids=True, row_stride=1, **param_over):
        """Renders file rows rows[0], rows[0]+row_stride, ... < rows[1] (default: all).  Returns dict of full-frame arrays
        (unrendered rows are NaN / -2) and, with counters=True, the traversal counters."""
        p = oracle_params(self.spec, **param_over)
        W, H = p.width, p.height
        r0, r1 = rows if rows is not None else (0, H)
        rad = np.full((H, W, 3), np.nan); col = np.full((H, W, 3), np.nan)
        rgb = np.zeros((H, W, 3), np.uint8); ids = np.full((H, W), -2, np.int32)
        cnt = orc_counters()
        cam = camera if camera is not None else self.camera
        t0 = time.perf_counter()
        rc = lib().orc_render(self._h, C.byref(cam), C.byref(p), r0, r1, row_stride, n_threads, _dp(rad), _dp(col),
                              rgb.ctypes.data_as(F.c_uint8_p), _ip(ids) if want_ids else None,
                              C.byref(cnt) if counters else None)
        self.last_render_s = time.perf_counter() - t0      # the orc_render call alone (bench.py cpu_baseline)
        if rc != 0:
            raise ValueError(f"orc_render failed: {rc}")
        res = {"radiance": rad, "color": col, "rgb8": rgb, "prim_id": ids}
        return (res, cnt.as_dict()) if counters else res

    def hit(self, origin, direction, tmin=0.001, tmax=10000.0):
        out = np.zeros(8)
        o, d = _d(origin), _d(direction)
        pid = lib().orc_hit_scene(self._h, _dp(o), _dp(d), tmin, tmax, _dp(out))
        return pid, out

    def trace_path(self, origin, direction, max_bounces, seed=0, pixel=0, sample=0):
        out = np.zeros(3)
        o, d = _d(origin), _d(direction)
        lib().orc_trace_path(self._h, _dp(o), _dp(d), max_bounces, seed, pixel, sample, _dp(out))
        return out

    def background(self, direction):
        out = np.zeros(3)
        d = _d(direction)
        lib().orc_background(self._h, _dp(d), _dp(out))
        return out

    def close(self):
        if self._h:
            lib().orc_scene_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


# ---- unit-level helpers for the known-answer tests
def hit_aabb(lo, hi, o, d, tmin, tmax):
    out = np.zeros(2)
    lo, hi, o, d = _d(lo), _d(hi), _d(o), _d(d)
    h = lib().orc_hit_aabb(_dp(lo), _dp(hi), _dp(o), _dp(d), tmin, tmax, _dp(out))
    return (out[0], out[1]) if h else None


def hit_prim(kind, data9, o, d, tmin=0.001, tmax=10000.0):
    out = np.zeros(8)
    data9, o, d = _d(data9), _d(o), _d(d)
    h = lib().orc_hit_prim(kind, _dp(data9), _dp(o), _dp(d), tmin, tmax, _dp(out))
    return {"t": out[0], "point": out[1:4].copy(), "normal": out[4:7].copy(), "front": bool(out[7])} if h else None


def prim_box(kind, data9):
    out = np.zeros(6)
    data9 = _d(data9)
    lib().orc_prim_box(kind, _dp(data9), _dp(out))
    return out[:3].copy(), out[3:].copy()


def schlick(c, r):
    return lib().orc_schlick(c, r)


def reflect(v, n):
    out = np.zeros(3); v, n = _d(v), _d(n)
    lib().orc_reflect(_dp(v), _dp(n), _dp(out))
    return out


def refract(v, n, eta):
    out = np.zeros(3); v, n = _d(v), _d(n)
    lib().orc_refract(_dp(v), _dp(n), eta, _dp(out))
    return out


def tonemap_gamma(exposure, gamma, c):
    out = np.zeros(3); c = _d(c)
    lib().orc_tonemap_gamma(exposure, gamma, _dp(c), _dp(out))
    return out


def generate_ray(cam, i, j, W, H, aa, seed, pixel, sample):
    out = np.zeros(6)
    lib().orc_generate_ray(C.byref(cam), i, j, W, H, int(aa), seed, pixel, sample, _dp(out))
    return out[:3].copy(), out[3:].copy()


def philox(ctr, key):
    c = (C.c_uint32 * 4)(*ctr); k = (C.c_uint32 * 2)(*key); o = (C.c_uint32 * 4)()
    lib().orc_philox(c, k, o)
    return [int(x) for x in o]
