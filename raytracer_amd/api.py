"""Python face of the render core: numpy in, numpy out, everything through the C ABIs.

`HostBvh`   = constructBVHWithLimit + flattening            (librthost, C++)
`GpuScene`  = rt_scene_create / rt_render / rt_render_tiles_device   (librtcore, HIP — needs an MI355X)
There is no fallback: without the HIP library or a device the calls raise `RtError`/`OSError`.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _ffi as F
from .scenes import SceneSpec


def _dp(a):
    return a.ctypes.data_as(F.c_double_p)


def _ip(a):
    return a.ctypes.data_as(F.c_int32_p)


def _vec3(dst, src):
    for k in range(3):
        dst[k] = float(src[k])


def default_camera(cam: dict, aspect: float) -> F.rt_camera:
    """defaultCamera (app/Rendering/Camera.hs:26-44) via librthost."""
    out = F.rt_camera()
    lf = (C.c_double * 3)(*map(float, cam["lookFrom"]))
    la = (C.c_double * 3)(*map(float, cam["lookAt"]))
    up = (C.c_double * 3)(*map(float, cam["vUp"]))
    F.check(F.host().rth_default_camera(lf, la, up, float(cam["vfov"]), float(aspect), float(cam["aperture"]),
                                        float(cam["focusDist"]), C.byref(out)), "host")
    return out


def make_params(spec: SceneSpec, **over) -> F.rt_params:
    p = F.rt_params()
    p.width, p.height, p.samples_per_pixel = spec.width, spec.height, spec.spp
    p.antialiasing = 1 if spec.antialiasing else 0
    p.max_bounces = spec.max_bounces
    p.rr_start_bounce = 0
    p.exposure, p.gamma = spec.exposure, spec.gamma
    p.seed = spec.seed
    p.collect_stats = 0
    p.box_precision = 0
    for k, v in over.items():
        setattr(p, k, v)
    return p


def make_materials(spec: SceneSpec):
    """JSON materials -> rt_material[] with the reference's ids (assignMaterialIds: from 1)."""
    n = len(spec.materials)
    arr = (F.rt_material * max(n, 1))()
    for i, (_, m) in enumerate(spec.materials):
        a = arr[i]
        a.id = i + 1
        a.flags = 0
        _vec3(a.diffuse, m.get("diffuseColor", [1.0, 1.0, 1.0]))     # Material.hs:40
        if m.get("emissionColor") is not None:
            a.flags |= F.RT_MAT_HAS_EMISSION; _vec3(a.emission, m["emissionColor"])
        if m.get("shininess") is not None:
            a.flags |= F.RT_MAT_HAS_SHININESS; a.shininess = float(m["shininess"])
        if m.get("ior") is not None:
            a.flags |= F.RT_MAT_HAS_IOR; a.ior = float(m["ior"])
        if m.get("transmission") is not None:
            a.flags |= F.RT_MAT_HAS_TRANSMISSION; a.transmission = float(m["transmission"])
    return arr, n


def make_lights(spec: SceneSpec):
    n = len(spec.lights)
    arr = (F.rt_light * max(n, 1))()
    for i, l in enumerate(spec.lights):
        if l["tag"] == "PointLight":
            arr[i].kind = F.RT_LIGHT_POINT; _vec3(arr[i].vec, l["position"])
        else:
            arr[i].kind = F.RT_LIGHT_DIRECTIONAL; _vec3(arr[i].vec, l["direction"])
        _vec3(arr[i].intensity, l["intensity"])
    return arr, n


def background_of(spec: SceneSpec):
    bg = spec.background
    if bg["tag"] == "Gradient":
        return F.RT_BG_GRADIENT, bg["color1"], bg["color2"]
    return F.RT_BG_SOLID, bg["color"], bg["color"]


class HostBvh:
    """The reference's BVH over `allObjects`, built and flattened by librthost."""

    def __init__(self, kind, data, material, leaf_threshold, max_depth, n_threads=1):
        self.kind = np.ascontiguousarray(kind, np.int32)
        self.data = np.ascontiguousarray(data, np.float64).reshape(-1, 9)
        self.material = np.ascontiguousarray(material, np.int32)
        self._h = C.c_void_p()
        F.check(F.host().rth_bvh_build(len(self.kind), _ip(self.kind), _dp(self.data), _ip(self.material),
                                       int(leaf_threshold), int(max_depth), int(n_threads), C.byref(self._h)), "host")
        nn, nl, dp, npr = C.c_int32(), C.c_int32(), C.c_int32(), C.c_int32()
        F.host().rth_bvh_info(self._h, C.byref(nn), C.byref(nl), C.byref(dp), C.byref(npr))
        self.n_nodes, self.n_leaves, self.depth, self.n_prims = nn.value, nl.value, dp.value, npr.value

    def fill(self, desc: F.rt_scene_desc):
        F.check(F.host().rth_bvh_fill_desc(self._h, C.byref(desc)), "host")

    def arrays(self):
        """Copies of the flattened arrays (for tests)."""
        d = F.rt_scene_desc()
        self.fill(d)
        nn, npr = d.n_nodes, d.n_prims
        g = lambda p, n, t: np.ctypeslib.as_array(p, shape=(n,)).astype(t, copy=True)
        return {"node_bounds": g(d.node_bounds, 6 * nn, np.float64).reshape(nn, 6),
                "node_child": g(d.node_child, 2 * nn, np.int32).reshape(nn, 2),
                "prim_kind": g(d.prim_kind, npr, np.int32), "prim_data": g(d.prim_data, 9 * npr, np.float64).reshape(npr, 9),
                "prim_material": g(d.prim_material, npr, np.int32), "prim_id": g(d.prim_id, npr, np.int32)}

    def close(self):
        if self._h:
            F.host().rth_bvh_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def fill_desc_common(desc: F.rt_scene_desc, spec: SceneSpec, keep: list):
    mats, nm = make_materials(spec)
    lights, nl = make_lights(spec)
    keep += [mats, lights]
    desc.n_materials, desc.materials = nm, mats
    desc.n_lights, desc.lights = nl, lights
    kind, c1, c2 = background_of(spec)
    desc.bg_kind = kind
    _vec3(desc.bg_color1, c1); _vec3(desc.bg_color2, c2)
    if spec.sky is not None:
        fmt, img = spec.sky
        img = np.ascontiguousarray(img)
        keep.append(img)
        desc.sky_format = F.RT_SKY_RGB8 if fmt == "rgb8" else F.RT_SKY_RGBF
        desc.sky_height, desc.sky_width = img.shape[0], img.shape[1]
        desc.sky_pixels = img.ctypes.data
    else:
        desc.sky_format = F.RT_SKY_NONE


class GpuScene:
    """A scene resident in HBM (rt_scene).  `desc_arrays` lets tests feed an externally built tree."""

    def __init__(self, spec: SceneSpec, bvh: HostBvh | None = None, desc_arrays: dict | None = None, n_threads=1, device=-1,
                 devices=None):
        """device: the one GPU this process renders on (rt_init).  devices: a list of device ids instead - a device group
        for the single-process fan-out of rt_render (rt_init_devices; ids may repeat: virtual devices for tests)."""
        self.spec = spec
        lib = F.core()
        if devices is not None:
            arr = np.ascontiguousarray(devices, np.int32)
            F.check(lib.rt_init_devices(len(arr), _ip(arr)))
        else:
            F.check(lib.rt_init(device))
        desc = F.rt_scene_desc()
        keep = []
        if desc_arrays is not None:
            a = {k: np.ascontiguousarray(v) for k, v in desc_arrays.items()}
            keep.append(a)
            desc.n_nodes = len(a["node_child"]); desc.node_bounds = _dp(a["node_bounds"]); desc.node_child = _ip(a["node_child"])
            desc.n_prims = len(a["prim_kind"]); desc.prim_kind = _ip(a["prim_kind"]); desc.prim_data = _dp(a["prim_data"])
            desc.prim_material = _ip(a["prim_material"]); desc.prim_id = _ip(a["prim_id"])
        else:
            if bvh is None:
                k, d, m = spec.all_objects()
                bvh = HostBvh(k, d, m, spec.leaf_threshold, spec.bvh_max_depth, n_threads)
            self.bvh = bvh
            bvh.fill(desc)
        fill_desc_common(desc, spec, keep)
        self._h = C.c_void_p()
        F.check(lib.rt_scene_create(C.byref(desc), C.byref(self._h)))
        self.camera = default_camera(spec.camera, spec.width / spec.height)

    def render(self, want=("color", "radiance", "rgb8", "prim_id"), camera=None, **param_over):
        """rt_render: one frame into host arrays (file order, row 0 = top).  Returns (dict, rt_stats)."""
        p = make_params(self.spec, **param_over)
        W, H = p.width, p.height
        out = F.rt_output()
        res = {}
        if "radiance" in want:
            res["radiance"] = np.empty((H, W, 3), np.float64); out.radiance = _dp(res["radiance"])
        if "color" in want:
            res["color"] = np.empty((H, W, 3), np.float64); out.color = _dp(res["color"])
        if "rgb8" in want:
            res["rgb8"] = np.empty((H, W, 3), np.uint8); out.rgb8 = res["rgb8"].ctypes.data_as(F.c_uint8_p)
        if "prim_id" in want:
            res["prim_id"] = np.empty((H, W), np.int32); out.prim_id = _ip(res["prim_id"])
        st = F.rt_stats()
        cam = camera if camera is not None else self.camera
        F.check(F.core().rt_render(self._h, C.byref(cam), C.byref(p), C.byref(out), C.byref(st)))
        return res, st

    def kernel_times(self):
        """rt_kernel_times: (sum of ms, number) of the render-kernel launches since the previous call (blocks)."""
        ms, n = C.c_double(0), C.c_int32(0)
        F.check(F.core().rt_kernel_times(self._h, C.byref(ms), C.byref(n)))
        return ms.value, n.value

    def render_tiles_device(self, rank, world, d_color=0, d_radiance=0, d_prim_id=0, stream=0, camera=None, noalloc=False,
                            d_rgb8=0, **param_over):
        """rt_render_tiles_device: this rank's tiles into device buffers (raw pointers), asynchronous.
        noalloc=True: rt_render_tiles_device_noalloc, the variant whose workspace is the library's static device array.
        d_rgb8 (alone): rt_render_tiles_rgb8_device, the packed 8-bit tiles quantised in the resolve pass."""
        p = make_params(self.spec, **param_over)
        st = F.rt_stats()
        cam = camera if camera is not None else self.camera
        if d_rgb8:
            if d_color or d_radiance or d_prim_id or noalloc:
                raise ValueError("d_rgb8 is its own entry point: no other output, no noalloc")
            F.check(F.core().rt_render_tiles_rgb8_device(self._h, C.byref(cam), C.byref(p), rank, world, C.c_void_p(d_rgb8),
                                                         C.c_void_p(stream or None), C.byref(st)))
            return st
        fn = F.core().rt_render_tiles_device_noalloc if noalloc else F.core().rt_render_tiles_device
        F.check(fn(self._h, C.byref(cam), C.byref(p), rank, world,
                   C.c_void_p(d_color or None), C.c_void_p(d_radiance or None),
                   C.c_void_p(d_prim_id or None), C.c_void_p(stream or None), C.byref(st)))
        return st

    def close(self):
        if self._h:
            F.core().rt_scene_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def prim_boxes(kind, data) -> np.ndarray:
    """The boxes rt_scene_create gives the render core's own tree: triangles padded by 1e-8 (Triangle.hs:24-28),
    spheres exact (Sphere.hs:20-22).  Planes get the box of their reference leaf there and are not handled here."""
    kind = np.asarray(kind); data = np.asarray(data, np.float64).reshape(-1, 9)
    out = np.empty((len(kind), 6))
    tri = kind == 0
    v = data[tri].reshape(-1, 3, 3)
    out[tri, :3] = v.min(axis=1) - 1e-8
    out[tri, 3:] = v.max(axis=1) + 1e-8
    sph = kind == 1
    out[sph, :3] = data[sph, :3] - np.abs(data[sph, 3:4])
    out[sph, 3:] = data[sph, :3] + np.abs(data[sph, 3:4])
    if (kind == 2).any():
        raise ValueError("planes are bounded by their reference leaf: build the scene instead")
    return out


def fast_bvh(boxes, max_leaf=4, n_threads=1):
    """rt_fast_bvh_build: the render core's own SAH tree over `boxes` (n, 6).  Returns a dict with the COMPACTED tree:
    node_bounds (m, 6), node_child (m, 2) (leaf: -1 - first_slot, count), order (n,), depth."""
    boxes = np.ascontiguousarray(boxes, np.float64).reshape(-1, 6)
    n = len(boxes)
    nb = np.zeros((2 * n, 6)); nc = np.zeros((2 * n, 2), np.int32); order = np.zeros(n, np.int32); depth = C.c_int32(0)
    F.check(F.core().rt_fast_bvh_build(n, _dp(boxes), int(max_leaf), int(n_threads), _dp(nb), _ip(nc), _ip(order), C.byref(depth)))
    return {"node_bounds": nb, "node_child": nc, "order": order, "depth": depth.value}


def tile_list(width, height, rank, world) -> np.ndarray:
    lib = F.core()
    n = F.check(lib.rt_tile_count(width, height, rank, world))
    t = np.empty(max(n, 1), np.int32)
    F.check(lib.rt_tile_list(width, height, rank, world, _ip(t)))
    return t[:n]


def untile_host(packed: np.ndarray, tiles: np.ndarray, width, height, n_comp) -> np.ndarray:
    """Host mirror of rt_untile_device (for the CPU-side gather tests): packed (ntiles*64, n_comp) -> (H, W, n_comp)
    contribution, NaN where this rank owns nothing."""
    tx = (width + 7) // 8
    img = np.full((height, width, n_comp), np.nan, packed.dtype)
    p = packed.reshape(-1, 64, n_comp)
    lane = np.arange(64)
    for k, t in enumerate(tiles):
        px = (t % tx) * 8 + (lane & 7)
        py = (t // tx) * 8 + (lane >> 3)
        ok = (px < width) & (py < height)
        img[py[ok], px[ok]] = p[k][ok]
    return img
