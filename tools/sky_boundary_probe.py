#!/usr/bin/env python3
"""Sky texel selection AT texel boundaries (SkySphere.hs:35-41): frames whose primary rays sweep through a column or row boundary
of the equirect sky in steps of 1e-18 ... 1e-13, GPU against the CPU oracle.  The sky has one colour per texel, so a differing
texel shows as a differing pixel; every difference is printed with its direction and the u w / v h the host's libm gives.
    python tools/sky_boundary_probe.py [sky width] [sky height]       exit status 1 on any difference"""
import math
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import oracle as O                     # noqa: E402
from oracle.pyref import hs_atan2                  # noqa: E402
from raytracer_amd import _ffi as F                # noqa: E402
from raytracer_amd import api, scenes              # noqa: E402

W, H = 256, 16                                     # frame: W steps across the boundary, H values of the other coordinate


def sweep_camera(c, t, s, step):
    """Rays (before normalisation) c + (px + 0.5 - W/2) step t + ((H-1-py) + 0.5 - H/2) / H * 0.3 s from the origin."""
    cam = F.rt_camera()
    c, t, s = np.asarray(c, float), np.asarray(t, float), np.asarray(s, float)
    hor, ver = t * (W * step), s * 0.3
    ll = c - hor / 2 - ver / 2
    for dst, src in ((cam.origin, (0, 0, 0)), (cam.lower_left, ll), (cam.horizontal, hor), (cam.vertical, ver),
                     (cam.u, (1, 0, 0)), (cam.v, (0, 1, 0)), (cam.w, (0, 0, 1))):
        for k in range(3):
            dst[k] = float(src[k])
    cam.aperture = 0.0
    return cam


def directions(cam):
    """The directions the sky lookup sees for a frame's primary rays (Camera.hs:54-60 without jitter or lens, then SkySphere.hs:34)."""
    px, py = np.meshgrid(np.arange(W), np.arange(H))
    ru, rv = (px + 0.5) / W, ((H - 1 - py) + 0.5) / H
    ll, hor, ver = (np.array(list(v)) for v in (cam.lower_left, cam.horizontal, cam.vertical))
    d = (ll[None, None, :] + ru[..., None] * hor) + rv[..., None] * ver
    for _ in range(2):                                                  # Camera.hs:60 normalises the ray, SkySphere.hs:34 normalises it again
        dot = d[..., 0] * d[..., 0] + d[..., 1] * d[..., 1] + d[..., 2] * d[..., 2]
        d = (1.0 / np.sqrt(dot))[..., None] * d                         # Vec3.hs:78-79: scale (1 / vLength v) v
    return d


def probe(sw, sh, steps=(1e-18, 1e-17, 1e-16, 1e-15, 1e-14, 1e-13), verbose=True):
    """(differing pixels, pixels compared) over all boundary cases of a sw x sh sky"""
    rng = np.random.default_rng(5)
    img = rng.uniform(0.1, 1.0, size=(sh, sw, 3)).astype(np.float32)
    spec = scenes.c2_spheres(width=W, height=H, spp=1, max_bounces=0)
    spec.config_objects = [("sphere", [0.0, -1e6, 0.0], 1.0, None)]          # out of every probe's way
    spec.obj_meshes = []
    spec.antialiasing = False
    spec.sky = ("rgbf", img)
    o, g = O.OracleScene(spec), api.GpuScene(spec)
    cases = []
    for k in sorted(set([0, sw // 8, sw // 4, 3 * sw // 8, sw // 2, 5 * sw // 8, 3 * sw // 4, 7 * sw // 8, 1, 5, sw // 2 + 3, sw - 1])):
        phi = 2 * math.pi * (k / sw - 0.5)
        for y0 in (0.0, 0.4, -2.5, 30.0):
            cases.append((f"col {k} y {y0}", (math.cos(phi), y0, math.sin(phi)), (-math.sin(phi), 0.0, math.cos(phi)), (0.0, 1.0, 0.0)))
    for axis in ((1, 0, 0), (-1, 0, 0), (0, 0, 1), (0, 0, -1)):                # exact axes: boundary directions with exact zeros
        tan = (-axis[2], 0, axis[0])
        cases.append((f"axis {axis}", axis, tan, (0.0, 1.0, 0.0)))
    for k in sorted(set([1, 2, sh // 4, sh // 2, 3 * sh // 4, sh - 1, sh // 2 + 1])):
        th = math.pi * (0.5 - k / sh)                                          # v = k / sh  <=>  asin(y) = th
        for phi in (0.3, 1.7, -2.2):
            c = (math.cos(th) * math.cos(phi), math.sin(th), math.cos(th) * math.sin(phi))
            t = (-math.sin(th) * math.cos(phi), math.cos(th), -math.sin(th) * math.sin(phi))
            cases.append((f"row {k} phi {phi}", c, t, (-math.sin(phi), 0.0, math.cos(phi))))
    bad = total = 0
    for name, c, t, s in cases:
        for step in steps:
            cam = sweep_camera(c, t, s, step)
            ref = o.render(camera=cam, n_threads=8)["radiance"]
            res, _ = g.render(camera=cam, want=("radiance",))
            diff = np.argwhere((res["radiance"] != ref).any(-1))
            total += W * H
            bad += len(diff)
            if len(diff) and verbose:
                d = directions(cam)
                print(f"{name} step {step:g}: {len(diff)} of {W * H} pixels differ")
                for py, px in diff[:6]:
                    x, y, z = d[py, px]
                    u = 0.5 + hs_atan2(z, x) / (2 * math.pi)
                    v = 0.5 - math.asin(max(-1.0, min(1.0, y))) / math.pi
                    print(f"    px {px} py {py} d ({x!r}, {y!r}, {z!r})  u w {u * sw!r}  v h {v * sh!r}")
    o.close(); g.close()
    return bad, total


def main():
    sw = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    sh = int(sys.argv[2]) if len(sys.argv) > 2 else 32
    bad, total = probe(sw, sh)
    print(f"sky boundary probe {sw}x{sh}: {bad} of {total} pixels differ")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
