"""ctypes bindings of the two C ABIs (include/rtcore.h, include/rthost.h).

The libraries are built in-tree by `make core host` (or `__graft_entry__.build()`); a missing
library raises — there is no Python or CPU fallback for the render core.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_DIR = os.path.join(_HERE, "lib")

c_double_p = C.POINTER(C.c_double)
c_int32_p = C.POINTER(C.c_int32)
c_uint8_p = C.POINTER(C.c_uint8)

RT_OK = 0
RT_ERR_INVALID, RT_ERR_NO_DEVICE, RT_ERR_HIP, RT_ERR_OOM, RT_ERR_EMPTY_SCENE, RT_ERR_UNSUPPORTED = -1, -2, -3, -4, -5, -6
RT_PRIM_TRIANGLE, RT_PRIM_SPHERE, RT_PRIM_PLANE = 0, 1, 2
RT_LIGHT_POINT, RT_LIGHT_DIRECTIONAL = 0, 1
RT_BG_GRADIENT, RT_BG_SOLID = 0, 1
RT_SKY_NONE, RT_SKY_RGB8, RT_SKY_RGBF = 0, 1, 2
RT_MAT_HAS_EMISSION, RT_MAT_HAS_SHININESS, RT_MAT_HAS_IOR, RT_MAT_HAS_TRANSMISSION = 1, 2, 4, 8


class rt_material(C.Structure):
    _fields_ = [("id", C.c_int32), ("flags", C.c_uint32), ("diffuse", C.c_double * 3), ("emission", C.c_double * 3),
                ("shininess", C.c_double), ("ior", C.c_double), ("transmission", C.c_double)]


class rt_light(C.Structure):
    _fields_ = [("kind", C.c_int32), ("reserved", C.c_int32), ("vec", C.c_double * 3), ("intensity", C.c_double * 3)]


class rt_scene_desc(C.Structure):
    _fields_ = [("n_nodes", C.c_int32), ("node_bounds", c_double_p), ("node_child", c_int32_p),
                ("n_prims", C.c_int32), ("prim_kind", c_int32_p), ("prim_data", c_double_p),
                ("prim_material", c_int32_p), ("prim_id", c_int32_p),
                ("n_materials", C.c_int32), ("materials", C.POINTER(rt_material)),
                ("n_lights", C.c_int32), ("lights", C.POINTER(rt_light)),
                ("bg_kind", C.c_int32), ("bg_color1", C.c_double * 3), ("bg_color2", C.c_double * 3),
                ("sky_format", C.c_int32), ("sky_width", C.c_int32), ("sky_height", C.c_int32),
                ("sky_pixels", C.c_void_p)]


class rt_camera(C.Structure):
    _fields_ = [("origin", C.c_double * 3), ("lower_left", C.c_double * 3), ("horizontal", C.c_double * 3),
                ("vertical", C.c_double * 3), ("u", C.c_double * 3), ("v", C.c_double * 3), ("w", C.c_double * 3),
                ("aperture", C.c_double)]


class rt_params(C.Structure):
    _fields_ = [("width", C.c_int32), ("height", C.c_int32), ("samples_per_pixel", C.c_int32),
                ("antialiasing", C.c_int32), ("max_bounces", C.c_int32), ("rr_start_bounce", C.c_int32),
                ("exposure", C.c_double), ("gamma", C.c_double), ("seed", C.c_uint64),
                ("collect_stats", C.c_int32), ("box_precision", C.c_int32)]


class rt_output(C.Structure):
    _fields_ = [("radiance", c_double_p), ("color", c_double_p), ("rgb8", c_uint8_p), ("prim_id", c_int32_p)]


class rt_stats(C.Structure):
    _fields_ = [("ms_render", C.c_double), ("ms_total", C.c_double), ("paths", C.c_uint64),
                ("rays_closest", C.c_uint64), ("rays_shadow", C.c_uint64), ("nodes_visited", C.c_uint64),
                ("tri_tests", C.c_uint64), ("sphere_tests", C.c_uint64), ("plane_tests", C.c_uint64),
                ("shaded_hits", C.c_uint64),
                ("sched_node_steps", C.c_uint64), ("sched_prim_steps", C.c_uint64), ("sched_shade_steps", C.c_uint64),
                ("top_node_hits", C.c_uint64), ("sched_live_lanes", C.c_uint64),
                ("cyc_vote", C.c_uint64), ("cyc_node", C.c_uint64), ("cyc_prim", C.c_uint64), ("cyc_shade", C.c_uint64), ("cyc_refill", C.c_uint64),
                ("rays_shadow_culled", C.c_uint64), ("rays_literal", C.c_uint64),
                ("shade_lanes", C.c_uint64 * 5), ("shade_trips", C.c_uint64 * 5), ("cyc_shade_detail", C.c_uint64 * 7)]


class rth_config_view(C.Structure):
    _fields_ = [("params", rt_params), ("use_bvh", C.c_int32), ("bvh_max_depth", C.c_int32), ("leaf_threshold", C.c_int32),
                ("camera", rt_camera), ("n_objs", C.c_int32), ("obj_kind", c_int32_p), ("obj_data", c_double_p),
                ("obj_material", c_int32_p), ("n_triangles", C.c_int32),
                ("n_materials", C.c_int32), ("materials", C.POINTER(rt_material)),
                ("n_lights", C.c_int32), ("lights", C.POINTER(rt_light)),
                ("bg_kind", C.c_int32), ("bg_color1", C.c_double * 3), ("bg_color2", C.c_double * 3),
                ("sky_path", C.c_char_p)]


# name -> (restype, argtypes): every symbol include/rtcore.h declares
RTCORE_SYMBOLS = {
    "rt_init": (C.c_int, [C.c_int]),
    "rt_shutdown": (None, []),
    "rt_last_error": (C.c_char_p, []),
    "rt_abi_version": (C.c_int, []),
    "rt_build_id": (C.c_char_p, []),
    "rt_kernel_name": (C.c_int, [C.c_void_p, C.POINTER(rt_params), C.c_char_p, C.c_int32]),
    "rt_tile_skew": (C.c_int, []),
    "rt_scene_create": (C.c_int, [C.POINTER(rt_scene_desc), C.POINTER(C.c_void_p)]),
    "rt_scene_destroy": (None, [C.c_void_p]),
    "rt_render": (C.c_int, [C.c_void_p, C.POINTER(rt_camera), C.POINTER(rt_params), C.POINTER(rt_output), C.POINTER(rt_stats)]),
    "rt_tile_count": (C.c_int, [C.c_int32, C.c_int32, C.c_int32, C.c_int32]),
    "rt_tile_list": (C.c_int, [C.c_int32, C.c_int32, C.c_int32, C.c_int32, c_int32_p]),
    "rt_render_tiles_device": (C.c_int, [C.c_void_p, C.POINTER(rt_camera), C.POINTER(rt_params), C.c_int32, C.c_int32,
                                         C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(rt_stats)]),
    "rt_render_tiles_device_noalloc": (C.c_int, [C.c_void_p, C.POINTER(rt_camera), C.POINTER(rt_params), C.c_int32, C.c_int32,
                                                 C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(rt_stats)]),
    "rt_render_tiles_rgb8_device": (C.c_int, [C.c_void_p, C.POINTER(rt_camera), C.POINTER(rt_params), C.c_int32, C.c_int32,
                                              C.c_void_p, C.c_void_p, C.POINTER(rt_stats)]),
    "rt_untile_all_bytes_device": (C.c_int, [C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_int64, C.c_void_p, C.c_int32, C.c_void_p]),
    "rt_init_devices": (C.c_int, [C.c_int32, c_int32_p]),
    "rt_device_count": (C.c_int, []),
    "rt_kernel_times": (C.c_int, [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_int32)]),
    "rt_untile_device": (C.c_int, [C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p]),
    "rt_untile_bytes_device": (C.c_int, [C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p]),
    "rt_quantize_device": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]),
    "rt_fast_bvh_build": (C.c_int, [C.c_int32, c_double_p, C.c_int32, C.c_int32, c_double_p, c_int32_p, c_int32_p, c_int32_p]),
}

# every symbol include/rthost.h declares
RTHOST_SYMBOLS = {
    "rth_last_error": (C.c_char_p, []),
    "rth_bvh_build": (C.c_int, [C.c_int32, c_int32_p, c_double_p, c_int32_p, C.c_int32, C.c_int32, C.c_int32, C.POINTER(C.c_void_p)]),
    "rth_bvh_destroy": (None, [C.c_void_p]),
    "rth_bvh_info": (None, [C.c_void_p, c_int32_p, c_int32_p, c_int32_p, c_int32_p]),
    "rth_bvh_fill_desc": (C.c_int, [C.c_void_p, C.POINTER(rt_scene_desc)]),
    "rth_default_camera": (C.c_int, [c_double_p, c_double_p, c_double_p, C.c_double, C.c_double, C.c_double, C.c_double, C.POINTER(rt_camera)]),
    "rth_config_load": (C.c_int, [C.c_char_p, C.POINTER(C.c_void_p)]),
    "rth_config_destroy": (None, [C.c_void_p]),
    "rth_config_get": (C.c_int, [C.c_void_p, C.POINTER(rth_config_view)]),
    "rth_sky_load": (C.c_int, [C.c_char_p, c_int32_p, c_int32_p, c_int32_p, C.POINTER(C.c_void_p)]),
    "rth_sky_free": (None, [C.c_void_p]),
    "rth_write_ppm": (C.c_int, [C.c_char_p, C.c_int32, C.c_int32, c_uint8_p]),
    "rth_format_ppm": (C.c_int64, [C.c_int32, C.c_int32, c_uint8_p, C.c_char_p, C.c_int64]),
}


def _bind(lib, table):
    for name, (res, args) in table.items():
        fn = getattr(lib, name)   # AttributeError if the library does not export it
        fn.restype = res
        fn.argtypes = args
    return lib


_core = None
_host = None


def core():
    """librtcore.so (HIP).  Raises OSError when it has not been built."""
    global _core
    if _core is None:
        # RTCORE_LIB: an alternative build of the SAME library (A/B runs of kernel variants, tools/ab.sh)
        path = os.environ.get("RTCORE_LIB") or os.path.join(LIB_DIR, "librtcore.so")
        if not os.path.exists(path):
            raise OSError(f"{path} is missing: run `make core` (hipcc --offload-arch=gfx950); there is no fallback")
        _core = _bind(C.CDLL(path, mode=C.RTLD_GLOBAL), RTCORE_SYMBOLS)
    return _core


def host():
    """librthost.so (C++ host mirror)."""
    global _host
    if _host is None:
        path = os.path.join(LIB_DIR, "librthost.so")
        if not os.path.exists(path):
            raise OSError(f"{path} is missing: run `make host`")
        _host = _bind(C.CDLL(path), RTHOST_SYMBOLS)
    return _host


class RtError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"rtcore error {code}: {msg}")
        self.code = code


def check(code, which="core"):
    if code < 0:
        lib = core() if which == "core" else host()
        msg = (lib.rt_last_error() if which == "core" else lib.rth_last_error()) or b""
        raise RtError(code, msg.decode("utf-8", "replace"))
    return code
