#!/usr/bin/env python3
"""Renders a reference-schema config.json with the CPU oracle and writes the P3 frame (bench/run_reference.sh).
The config goes through the C++ host mirror (librthost: JSON / OBJ / MTL ingest with the reference's quirks), the frame
through oracle/liboracle.so.  TEST INFRASTRUCTURE: never used by the product path."""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import oracle as O                      # noqa: E402
from raytracer_amd import _ffi as F                 # noqa: E402


def main():
    cfg_path, out_path = sys.argv[1], sys.argv[2]
    h = C.c_void_p()
    F.check(F.host().rth_config_load(cfg_path.encode(), C.byref(h)), "host")
    v = F.rth_config_view()
    F.check(F.host().rth_config_get(h, C.byref(v)), "host")
    n = v.n_objs
    kind = np.ctypeslib.as_array(v.obj_kind, shape=(n,)).copy()
    data = np.ctypeslib.as_array(v.obj_data, shape=(9 * n,)).copy()
    mat = np.ctypeslib.as_array(v.obj_material, shape=(n,)).copy()
    L = O.lib()
    sky_fmt, sw, sh, sky_p, keep = F.RT_SKY_NONE, 0, 0, None, None
    if v.sky_path:
        fmt, w_, h_, px = C.c_int32(), C.c_int32(), C.c_int32(), C.c_void_p()
        F.check(F.host().rth_sky_load(v.sky_path, C.byref(fmt), C.byref(w_), C.byref(h_), C.byref(px)), "host")
        sky_fmt, sw, sh, sky_p = fmt.value, w_.value, h_.value, px
    c1 = np.array(list(v.bg_color1)); c2 = np.array(list(v.bg_color2))
    scene = L.orc_scene_create(n, kind.ctypes.data_as(F.c_int32_p), data.ctypes.data_as(F.c_double_p), mat.ctypes.data_as(F.c_int32_p),
                               v.leaf_threshold, v.bvh_max_depth, v.n_materials, v.materials, v.n_lights, v.lights, v.bg_kind,
                               c1.ctypes.data_as(F.c_double_p), c2.ctypes.data_as(F.c_double_p), sky_fmt, sw, sh, sky_p)
    if not scene:
        raise SystemExit("oracle: empty scene")
    p = v.params
    W, H = p.width, p.height
    rad = np.zeros((H, W, 3)); col = np.zeros((H, W, 3)); rgb = np.zeros((H, W, 3), np.uint8)
    rc = L.orc_render(scene, C.byref(v.camera), C.byref(p), 0, H, 1, os.cpu_count() or 1, rad.ctypes.data_as(F.c_double_p),
                      col.ctypes.data_as(F.c_double_p), rgb.ctypes.data_as(F.c_uint8_p), None, None)
    if rc != 0:
        raise SystemExit(f"orc_render failed: {rc}")
    F.check(F.host().rth_write_ppm(out_path.encode(), W, H, rgb.ctypes.data_as(F.c_uint8_p)), "host")
    L.orc_scene_destroy(scene)
    F.host().rth_config_destroy(h)


if __name__ == "__main__":
    main()
