#!/usr/bin/env python3
"""Parity hunt: random scenes against the CPU ORACLE (not only mode 0 against mode 1): triangle soups with room-sized
triangles, spheres (a few with negative or zero radius), planes of random point and normal, duplicated config triangles,
glass / metal / emissive materials, point and directional lights, gradient / solid / LDR / HDR sky, thin lens, random
leafThreshold / bvhMaxDepth / maxBounces, Russian roulette on some seeds; all three box_precision modes.
Prints one line per seed; exit status 1 if any pixel's id differs or any radiance differs by more than 1e-12 (relative to
max(1, |radiance|)).      python tools/fuzz_oracle.py <first seed> <last seed + 1>   |   --seeds 180,712   [--big] [--extreme] [--axis] [--nan] [--degenerate]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import oracle as O                     # noqa: E402
from raytracer_amd import api, scenes              # noqa: E402


def scene(seed, big=False, extreme=False):
    """big: 3000 / 40000 triangles, frames around 300 x 200, up to 30 bounces (deep trees, many refills per wave)."""
    rng = np.random.default_rng(seed)
    n_tris = int(rng.choice([3000, 40000] if big else [30, 300, 3000]))
    ctr = rng.uniform([-4, -0.5, -4], [4, 4, 4], size=(n_tris, 3))
    size = np.where(rng.random(n_tris) < 0.02, 6.0, rng.uniform(0.05, 0.7, n_tris))
    tri = ctr[:, None, :] + rng.uniform(-1, 1, size=(n_tris, 3, 3)) * size[:, None, None]
    names = ["matte", "metal", "glass", "lamp", None]
    fm = [names[i] for i in rng.integers(0, len(names), n_tris)]
    mats = [("matte", {"diffuseColor": list(rng.uniform(0.2, 0.9, 3))}),
            ("metal", {"diffuseColor": [0.9, 0.9, 0.9], "shininess": float(rng.choice([50.0, 300.0]))}),
            ("glass", {"diffuseColor": [1.0, 1.0, 1.0], "ior": float(rng.uniform(1.1, 1.8)), "transmission": 1.0}),
            ("lamp", {"diffuseColor": [0.8, 0.8, 0.8], "emissionColor": list(rng.uniform(0.0, 3.0, 3))})]
    cfg = []
    for _ in range(int(rng.integers(0, 7))):
        r = float(rng.uniform(0.2, 1.3))
        if rng.random() < 0.15:
            r = -r
        if rng.random() < 0.05:
            r = 0.0
        cfg.append(("sphere", list(rng.uniform(-3, 3, 3) + [0, 1.5, 0]), r, names[int(rng.integers(0, 5))]))
    for _ in range(int(rng.integers(0, 4))):
        n = rng.normal(size=3); n /= np.linalg.norm(n)
        cfg.append(("plane", list(rng.uniform([-2, -1.5, -2], [2, 1.0, 2])), list(n), names[int(rng.integers(0, 5))]))
    for _ in range(int(rng.integers(0, 3))):
        v = rng.uniform(-5, 5, size=(3, 3))
        cfg.append(("triangle", list(v[0]), list(v[1]), list(v[2]), names[int(rng.integers(0, 5))]))
    lights = []
    for _ in range(int(rng.integers(0, 4))):
        if rng.random() < 0.6:
            lights.append({"tag": "PointLight", "position": list(rng.uniform(-8, 8, 3) + [0, 6, 0]), "intensity": list(rng.uniform(5, 60, 3))})
        else:
            lights.append({"tag": "DirectionalLight", "direction": list(rng.normal(size=3)), "intensity": list(rng.uniform(0.1, 0.8, 3))})
    bg = ({"tag": "Gradient", "color1": [1.0, 1.0, 1.0], "color2": [0.5, 0.7, 1.0]} if rng.random() < 0.5
          else {"tag": "SolidColor", "color": list(rng.uniform(0, 1.5, 3))})
    spec = scenes.SceneSpec(
        name=f"fz{seed}", width=int(rng.integers(200, 400) if big else rng.integers(40, 160)),
        height=int(rng.integers(120, 260) if big else rng.integers(24, 90)), spp=int(rng.integers(1, 9 if big else 5)),
        antialiasing=bool(rng.random() < 0.8), gamma=float(rng.choice([1.0, 0.8, 2.2])), exposure=float(rng.uniform(0.5, 2.0)),
        background=bg,
        camera={"lookFrom": list(rng.uniform(-9, 9, 3) + [0, 3, 0]), "lookAt": [0.0, 1.0, 0.0], "vUp": [0.0, 1.0, 0.0],
                "vfov": float(rng.uniform(30, 90)), "aperture": float(rng.choice([0.0, 0.0, 0.3])), "focusDist": 3.0},
        max_bounces=int(rng.choice([1, 5, 10, 30]) if big else rng.integers(0, 8)),
        bvh_max_depth=int(rng.choice([12, 20, 32, 50]) if big else rng.choice([0, 2, 6, 12, 32])), leaf_threshold=int(rng.integers(1, 9)),
        materials=mats, lights=lights, config_objects=cfg,
        obj_meshes=[scenes.ObjMesh(tri.reshape(-1, 3), np.arange(3 * n_tris).reshape(n_tris, 3), fm, tuple(rng.uniform(-1, 1, 3)))],
        seed=int(rng.integers(0, 2 ** 40)))
    k = rng.random()
    if k < 0.25:
        spec.sky = scenes.sky_gradient_texture(fmt="rgbf")
    elif k < 0.4:
        spec.sky = scenes.sky_gradient_texture(fmt="rgb8")
    elif k < 0.5:
        spec.sky = scenes.sunset_sky(int(rng.integers(16, 300)), int(rng.integers(8, 150)), seed=seed)
    rr = int(rng.choice([0, 0, 0, 1, 2]))
    if extreme:
        # the FP32 box test's error terms at their largest: scene scaled by 0.01 ... 300 and moved up to 1e6 away from the origin
        # (point lights keep their effect: intensity x scale^2), a long lens from far away, or a view exactly along an axis
        sc = float(rng.choice([0.01, 0.1, 30.0, 300.0]))
        off = rng.uniform(-1, 1, 3) * float(rng.choice([0.0, 1e3, 1e6]))
        m = spec.obj_meshes[0]
        m.verts = m.verts * sc
        m.position = tuple(np.asarray(m.position) * sc + off)
        cfg2 = []
        for c in spec.config_objects:
            if c[0] == "sphere":
                cfg2.append(("sphere", list(np.asarray(c[1]) * sc + off), c[2] * sc, c[3]))
            elif c[0] == "plane":
                cfg2.append(("plane", list(np.asarray(c[1]) * sc + off), c[2], c[3]))
            else:
                cfg2.append(("triangle", list(np.asarray(c[1]) * sc + off), list(np.asarray(c[2]) * sc + off), list(np.asarray(c[3]) * sc + off), c[4]))
        spec.config_objects = cfg2
        for L in spec.lights:
            if L["tag"] == "PointLight":
                L["position"] = list(np.asarray(L["position"]) * sc + off)
                L["intensity"] = list(np.asarray(L["intensity"]) * sc * sc)
        cam = spec.camera
        lf, la = np.asarray(cam["lookFrom"]) * sc + off, np.asarray(cam["lookAt"]) * sc + off
        k = rng.random()
        if k < 0.3:                                        # long lens
            far = float(rng.choice([10.0, 100.0]))
            lf = la + (lf - la) * far
            cam["vfov"] = cam["vfov"] / far
        elif k < 0.5:                                      # exactly along an axis
            ax = int(rng.integers(0, 3))
            d = np.zeros(3); d[ax] = float(rng.choice([-1.0, 1.0])) * np.linalg.norm(lf - la)
            lf = la - d
            if ax == 1:
                cam["vUp"] = [0.0, 0.0, 1.0]
        cam["lookFrom"], cam["lookAt"] = list(lf), list(la)
        cam["focusDist"] = cam["focusDist"] * sc
        cam["aperture"] = cam["aperture"] * sc
    return spec, rr


def scene_axis(seed):
    """Spheres hit where they TOUCH their own box.  Sphere.hs:20-22 boxes a sphere exactly (c -+ r, no pad), so
    a ray along an axis through the centre has t1 == the box's entry distance in exact arithmetic, and in FP64 t1 falls one ulp
    below it about one time in three: `contains interval t1` (Sphere.hs:26) then fails and the reference sees through the sphere
    whenever its leaf is entered through an internal node.  Camera exactly on an axis through a sphere's centre, odd frame
    dimensions (the centre pixel's ray has rayu = rayv = 0.5 exactly); every third seed the camera is moved sideways by the
    radius, so that the centre ray is TANGENT to the sphere at an equator point and runs inside a face of its box."""
    rng = np.random.default_rng(1_000_003 * 7 + seed)
    names = ["matte", "metal", "glass", "lamp", None]
    mats = [("matte", {"diffuseColor": list(rng.uniform(0.2, 0.9, 3))}),
            ("metal", {"diffuseColor": [0.9, 0.9, 0.9], "shininess": 300.0}),
            ("glass", {"diffuseColor": [1.0, 1.0, 1.0], "ior": float(rng.uniform(1.1, 1.8)), "transmission": 1.0}),
            ("lamp", {"diffuseColor": [0.8, 0.8, 0.8], "emissionColor": list(rng.uniform(0.0, 3.0, 3))})]
    n_sph = int(rng.integers(2, 9))
    cfg, spheres = [], []
    for _ in range(n_sph):
        c = rng.uniform(-4, 4, 3)
        r = float(rng.uniform(0.05, 1.2))
        if rng.random() < 0.3:                                   # round numbers: exactly representable box faces
            c, r = np.round(c * 4) / 4, float(np.round(r * 8) / 8 + 0.125)
        spheres.append((c, r))
        cfg.append(("sphere", [float(x) for x in c], r, names[int(rng.integers(0, 5))]))
    if rng.random() < 0.5:                                       # a few triangles, so that leaves mix kinds
        for _ in range(int(rng.integers(1, 4))):
            v = rng.uniform(-5, 5, size=(3, 3))
            cfg.append(("triangle", list(v[0]), list(v[1]), list(v[2]), names[int(rng.integers(0, 5))]))
    c, r = spheres[int(rng.integers(0, n_sph))]
    ax = int(rng.integers(0, 3))
    e = np.zeros(3); e[ax] = float(rng.choice([-1.0, 1.0]))
    dist = float(rng.uniform(2.0, 30.0))
    side = np.zeros(3)
    if seed % 3 == 2:                                            # grazing: the centre ray touches the sphere at an equator point
        side[(ax + 1 + int(rng.integers(0, 2))) % 3] = float(rng.choice([-1.0, 1.0])) * r
    lf, la = c + side - dist * e, c + side
    lights = []
    for _ in range(int(rng.integers(0, 3))):                     # lights on an axis through another sphere: shadow rays through touch points
        c2, r2 = spheres[int(rng.integers(0, n_sph))]
        e2 = np.zeros(3); e2[int(rng.integers(0, 3))] = float(rng.choice([-1.0, 1.0]))
        lights.append({"tag": "PointLight", "position": list(c2 + e2 * float(rng.uniform(3, 12))), "intensity": list(rng.uniform(5, 60, 3))})
    spec = scenes.SceneSpec(
        name=f"axis{seed}", width=2 * int(rng.integers(8, 40)) + 1, height=2 * int(rng.integers(5, 24)) + 1, spp=int(rng.integers(1, 3)),
        antialiasing=bool(rng.random() < 0.25), gamma=1.0, exposure=1.0,
        background={"tag": "SolidColor", "color": list(rng.uniform(0, 1.5, 3))},
        camera={"lookFrom": [float(x) for x in lf], "lookAt": [float(x) for x in la], "vUp": [0.0, 0.0, 1.0] if ax == 1 else [0.0, 1.0, 0.0],
                "vfov": float(rng.uniform(5, 70)), "aperture": 0.0, "focusDist": 3.0},
        max_bounces=int(rng.integers(1, 5)), bvh_max_depth=int(rng.choice([2, 6, 32])), leaf_threshold=int(rng.integers(1, 5)),
        materials=mats, lights=lights, config_objects=cfg, seed=int(rng.integers(0, 2 ** 40)))
    return spec, 0


def scene_nan(seed):
    """The standard family with what makes NaN and infinite radiance: an RGBF sky with infinite and NaN texels and a black
    (attenuation 0) material on a third of the triangles - 0 x inf bounce terms, which the reference's clamp (GHC's max / min)
    turns into 0, and non-finite primary misses, which it returns unclamped."""
    spec, rr = scene(seed)
    rng = np.random.default_rng(77_000_000 + seed)
    img = scenes.sky_gradient_texture(w=48, h=24, fmt="rgbf")[1].copy()
    m = rng.random(img.shape[:2])
    img[m < 0.35] = np.inf
    img[m > 0.9] = np.nan
    spec.sky = ("rgbf", img)
    spec.materials[0] = ("matte", {"diffuseColor": [0.0, 0.0, 0.0], "emissionColor": list(rng.uniform(0.0, 0.2, 3))})
    return spec, rr


def scene_degenerate(seed):
    """The standard family with inputs a config file can hold and a careful user would not write: a directional light of
    direction 0 (normalize gives NaN), lights ON a surface point, at a sphere's centre or at the camera, zero / negative / huge
    intensities, triangles of zero area and of one point, a plane of normal 0, glass of ior 1 and ior 0, shininess 0, a lens with
    focus distance 0, a 1e-4 degree field of view, exposure 0, and (one seed in eight) vUp along the view direction: a NaN camera
    basis, every ray NaN.  Nothing here is physically meaningful; the reference still computes SOMETHING for each, by the
    comparison and min / max rules of its operators, and the kernel must compute the same."""
    spec, rr = scene(seed)
    rng = np.random.default_rng(55_000_000 + seed)
    m = spec.obj_meshes[0]
    tri0 = (m.verts[m.faces[0]] + np.asarray(m.position)) if len(m.faces) else np.zeros((3, 3))
    picks = set(int(x) for x in rng.choice(14, size=int(rng.integers(1, 5)), replace=False))
    if 0 in picks:
        spec.lights.append({"tag": "DirectionalLight", "direction": [0.0, 0.0, 0.0], "intensity": [0.5, 0.4, 0.3]})
    if 1 in picks:                                               # a light exactly at a triangle vertex / inside the triangle's plane
        spec.lights.append({"tag": "PointLight", "position": [float(x) for x in tri0[0]], "intensity": list(rng.uniform(5, 60, 3))})
        spec.lights.append({"tag": "PointLight", "position": [float(x) for x in tri0.mean(0)], "intensity": list(rng.uniform(5, 60, 3))})
    if 2 in picks:
        for c in spec.config_objects:
            if c[0] == "sphere":
                spec.lights.append({"tag": "PointLight", "position": list(c[1]), "intensity": [20.0, 20.0, 20.0]}); break
    if 3 in picks:
        spec.lights.append({"tag": "PointLight", "position": list(spec.camera["lookFrom"]), "intensity": [30.0, 0.0, -5.0]})
    if 4 in picks:
        spec.lights.append({"tag": "PointLight", "position": list(rng.uniform(-5, 5, 3)), "intensity": [1e300, 1e-300, 0.0]})
    if 5 in picks:                                               # zero-area triangles
        v = rng.uniform(-3, 3, size=(2, 3))
        spec.config_objects.append(("triangle", list(v[0]), list(v[0]), list(v[1]), "matte"))
        spec.config_objects.append(("triangle", list(v[1]), list(v[1]), list(v[1]), "metal"))
        spec.config_objects.append(("triangle", list(v[0]), list((v[0] + v[1]) / 2), list(v[1]), "glass"))
    if 6 in picks:
        spec.config_objects.append(("plane", list(rng.uniform(-1, 1, 3)), [0.0, 0.0, 0.0], "matte"))
    if 7 in picks:
        spec.materials[2] = ("glass", {"diffuseColor": [1.0, 1.0, 1.0], "ior": float(rng.choice([1.0, 0.0, -1.5, 1e-9])), "transmission": float(rng.choice([1.0, 0.5, 0.0]))})
    if 8 in picks:
        spec.materials[1] = ("metal", {"diffuseColor": [0.9, -0.2, 1.7], "shininess": float(rng.choice([0.0, -3.0, 1e9]))})
    if 9 in picks:
        spec.camera["aperture"] = 0.4; spec.camera["focusDist"] = 0.0
    if 10 in picks:
        spec.camera["vfov"] = float(rng.choice([1e-4, 179.999, 180.0]))
    if 11 in picks:
        spec.exposure = 0.0
    if 12 in picks:
        spec.gamma = float(rng.choice([1e-3, 1e3]))
    if 13 in picks and seed % 2 == 0:                            # vUp along the view direction: cross product 0, normalize NaN
        lf, la = np.asarray(spec.camera["lookFrom"]), np.asarray(spec.camera["lookAt"])
        spec.camera["vUp"] = [float(x) for x in (la - lf)]
    return spec, rr


def main():
    big, extreme, axis, nan, degen = "--big" in sys.argv, "--extreme" in sys.argv, "--axis" in sys.argv, "--nan" in sys.argv, "--degenerate" in sys.argv
    for f in ("--big", "--extreme", "--axis", "--nan", "--degenerate"):
        if f in sys.argv:
            sys.argv.remove(f)
    if len(sys.argv) > 2 and sys.argv[1] == "--seeds":
        seeds = [int(a) for a in sys.argv[2].split(",")]
    else:
        seeds = range(*((int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (0, 40)))
    bad = 0
    for seed in seeds:
        spec, rr = scene_axis(seed) if axis else (scene_nan(seed) if nan else (scene_degenerate(seed) if degen else scene(seed, big, extreme)))
        o = O.OracleScene(spec)
        ref = o.render(rr_start_bounce=rr, n_threads=min(16, os.cpu_count() or 8))
        o.close()
        g = api.GpuScene(spec)
        msg = []
        for mode in (0, 1, 2):
            res, st = g.render(box_precision=mode, rr_start_bounce=rr, collect_stats=1)
            ids = int((res["prim_id"] != ref["prim_id"]).sum())
            with np.errstate(invalid="ignore"):
                rel = np.abs(res["radiance"] - ref["radiance"]) / np.maximum(1.0, np.abs(ref["radiance"]))
            nan_mismatch = int((np.isnan(res["radiance"]) != np.isnan(ref["radiance"])).sum())
            nan_mismatch += int((np.isposinf(res["radiance"]) != np.isposinf(ref["radiance"])).sum() + (np.isneginf(res["radiance"]) != np.isneginf(ref["radiance"])).sum())
            rel = np.where(np.isfinite(ref["radiance"]) & np.isfinite(res["radiance"]), rel, 0.0)
            worst = float(np.nanmax(rel)) if np.isfinite(rel).any() else 0.0
            q = int((res["rgb8"] != ref["rgb8"]).sum())
            ok = ids == 0 and worst <= 1e-12 and nan_mismatch == 0 and q == 0
            bad += not ok
            msg.append(f"m{mode}: ids {ids} rel {worst:.1e} nan {nan_mismatch} q {q} lit {st.rays_literal} {'OK' if ok else 'MISMATCH'}")
        g.close()
        kinds = [c[0] for c in spec.config_objects]
        print(seed, f"{spec.width}x{spec.height}x{spec.spp} mb{spec.max_bounces} lt{spec.leaf_threshold} d{spec.bvh_max_depth} rr{rr}",
              f"sph{kinds.count('sphere')} pl{kinds.count('plane')} L{len(spec.lights)} sky{spec.sky[0] if spec.sky else '-'}", "|", " | ".join(msg), flush=True)
    print("mismatching (seed, mode) pairs:", bad)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
