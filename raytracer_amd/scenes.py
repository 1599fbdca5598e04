"""Synthetic scenes for the BASELINE.json configs.

The reference ships no assets (.gitignore:28-33 of the reference drops *.obj/*.mtl/scenes), so every
benchmark/parity scene is synthesised here with the triangle counts BASELINE.json names.  A scene is
kept as a `SceneSpec`: the reference's `Config` (app/Config.hs:99-105) with its OBJ files already
turned into triangles the way app/Parser/Object.hs does it (3-gons emitted as (v0, v2, v1), offset
added), in the reference's `allObjects` order (app/Rendering/SceneParser.hs:67-73:
config triangles ++ obj triangles ++ all config objects — config triangles appear twice).

`write_reference_files` emits the same scene as JSON + OBJ in the reference's schema, so the real
Haskell binary (or rtcli) can consume it.
"""
from __future__ import annotations

import json
import os
from dataclasses import dataclass, field

import numpy as np

TRI, SPHERE, PLANE = 0, 1, 2


@dataclass
class ObjMesh:
    """An OBJ file as the reference reads it: `v` lines and 3-index `f` lines, one `usemtl` per face group."""
    verts: np.ndarray            # (nv, 3) float64
    faces: np.ndarray            # (nf, 3) int64, 0-based
    face_material: list          # per face: material NAME or None
    position: tuple = (0.0, 0.0, 0.0)   # objposition

    def triangles(self):
        """parseFace + offsetTriangle (app/Parser/Object.hs:49-57, 85-86): (v0, v2, v1) + offset."""
        off = np.asarray(self.position, dtype=np.float64)
        v = self.verts
        f = self.faces
        t = np.stack([v[f[:, 0]] + off, v[f[:, 2]] + off, v[f[:, 1]] + off], axis=1)   # (nf, 3, 3)
        return t.reshape(-1, 9)


@dataclass
class SceneSpec:
    name: str
    width: int
    height: int
    spp: int
    antialiasing: bool
    gamma: float
    exposure: float
    background: dict             # {"tag": "Gradient", "color1", "color2"} | {"tag": "SolidColor", "color"}
    camera: dict                 # lookFrom, lookAt, vUp, vfov, aperture, focusDist
    max_bounces: int
    bvh_max_depth: int
    leaf_threshold: int
    materials: list = field(default_factory=list)       # [(name, {diffuseColor, emissionColor, shininess, ior, transmission})]
    lights: list = field(default_factory=list)          # [{"tag": "PointLight", "position", "intensity"} | {"tag": "DirectionalLight", ...}]
    config_objects: list = field(default_factory=list)  # [("sphere", center, radius, name) | ("plane", point, normal, name) | ("triangle", v0, v1, v2, name)]
    obj_meshes: list = field(default_factory=list)      # [ObjMesh]
    sky: tuple | None = None     # (format "rgb8"|"rgbf", ndarray (h, w, 3))
    seed: int = 0

    # ---- the reference's scene assembly (SceneParser.hs:26-80, Parser/Material.hs:52-57)
    def material_ids(self):
        """assignMaterialIds: JSON materials get ids from 1 in list order."""
        return {name: i + 1 for i, (name, _) in enumerate(self.materials)}

    def all_objects(self):
        """Returns (kind int32[n], data float64[n,9], material int32[n]) in `allObjects` order."""
        ids = self.material_ids()
        mid = lambda name: ids.get(name, 0) if name is not None else 0
        kinds, datas, mats = [], [], []

        def cfg(o):
            d = np.zeros(9)
            if o[0] == "sphere":
                d[0:3] = o[1]; d[3] = o[2]
                return SPHERE, d, mid(o[3])
            if o[0] == "plane":
                d[0:3] = o[1]; d[3:6] = o[2]
                return PLANE, d, mid(o[3])
            d[0:3] = o[1]; d[3:6] = o[2]; d[6:9] = o[3]
            return TRI, d, mid(o[4])

        cfg_all = [cfg(o) for o in self.config_objects]
        for k, d, m in cfg_all:                       # configTriangles
            if k == TRI:
                kinds.append(np.array([k], np.int32)); datas.append(d[None]); mats.append(np.array([m], np.int32))
        for mesh in self.obj_meshes:                  # allObjTriangles
            t = mesh.triangles()
            kinds.append(np.zeros(len(t), np.int32)); datas.append(t)
            mats.append(np.array([mid(n) for n in mesh.face_material], np.int32))
        for k, d, m in cfg_all:                       # hittables (all config objects, triangles again)
            kinds.append(np.array([k], np.int32)); datas.append(d[None]); mats.append(np.array([m], np.int32))
        if not kinds:
            return np.zeros(0, np.int32), np.zeros((0, 9)), np.zeros(0, np.int32)
        return (np.ascontiguousarray(np.concatenate(kinds)), np.ascontiguousarray(np.concatenate(datas)),
                np.ascontiguousarray(np.concatenate(mats)))

    def n_triangles(self):
        return sum(len(m.faces) for m in self.obj_meshes) + sum(1 for o in self.config_objects if o[0] == "triangle")

    def to_config_json(self, obj_paths):
        objs = []
        for o in self.config_objects:
            if o[0] == "sphere":
                objs.append({"tag": "SphereObj", "contents": [list(map(float, o[1])), float(o[2]), [1.0, 1.0, 1.0], o[3]]})
            elif o[0] == "plane":
                objs.append({"tag": "PlaneObj", "contents": [list(map(float, o[1])), list(map(float, o[2])), [1.0, 1.0, 1.0], o[3]]})
            else:
                objs.append({"tag": "TriangleObj", "contents": [list(map(float, o[1])), list(map(float, o[2])), list(map(float, o[3])), [1.0, 1.0, 1.0], o[4]]})
        scene = {"materials": [[n, m] for n, m in self.materials], "lights": self.lights}
        if objs:
            scene["objects"] = objs
        if self.obj_meshes:
            scene["objFiles"] = [{"path": p, "objposition": list(map(float, m.position))} for p, m in zip(obj_paths, self.obj_meshes)]
        return {
            "image": {"width": self.width, "height": self.height, "samplesPerPixel": self.spp,
                      "antialiasing": self.antialiasing, "gamma": self.gamma, "exposure": self.exposure},
            "background": self.background,
            "camera": self.camera,
            "raytracer": {"maxBounces": self.max_bounces, "useBVH": True, "bvhMaxDepth": self.bvh_max_depth,
                          "leafThreshold": self.leaf_threshold},
            "scene": scene,
        }


def write_reference_files(spec: SceneSpec, directory: str):
    """Writes <name>.json + <name>_<k>.obj in the reference's schema; returns the JSON path.
    Coordinates are written with repr() (17 significant digits) so they parse back to identical doubles."""
    os.makedirs(directory, exist_ok=True)
    obj_paths = []
    for k, mesh in enumerate(spec.obj_meshes):
        p = os.path.abspath(os.path.join(directory, f"{spec.name}_{k}.obj"))
        with open(p, "w") as f:
            f.write("# synthetic mesh (raytracer_amd.scenes)\n")
            for v in mesh.verts:
                f.write(f"v {float(v[0])!r} {float(v[1])!r} {float(v[2])!r}\n")
            cur = object()
            for face, mat in zip(mesh.faces, mesh.face_material):
                if mat != cur:
                    if mat is not None:
                        f.write(f"usemtl {mat}\n")
                    cur = mat
                f.write(f"f {face[0] + 1} {face[1] + 1} {face[2] + 1}\n")
        obj_paths.append(p)
    jp = os.path.join(directory, f"{spec.name}.json")
    with open(jp, "w") as f:
        json.dump(spec.to_config_json(obj_paths), f, indent=1)
    return jp


# ------------------------------------------------------------------------------------------ meshes
def uv_sphere(segments: int, rings: int, radius=1.0, center=(0.0, 0.0, 0.0), displacement=None):
    """UV sphere: 2*segments*(rings-1) triangles.  displacement(theta, phi) -> relative radial offset."""
    verts = []
    c = np.asarray(center, dtype=np.float64)

    def point(theta, phi):
        r = radius * (1.0 + (displacement(theta, phi) if displacement is not None else 0.0))
        return c + r * np.array([np.sin(theta) * np.cos(phi), np.cos(theta), np.sin(theta) * np.sin(phi)])

    verts.append(point(0.0, 0.0))
    for i in range(1, rings):
        theta = np.pi * i / rings
        for j in range(segments):
            verts.append(point(theta, 2 * np.pi * j / segments))
    verts.append(point(np.pi, 0.0))
    verts = np.array(verts, dtype=np.float64)
    top, bottom = 0, len(verts) - 1
    ring = lambda i, j: 1 + (i - 1) * segments + (j % segments)
    faces = []
    for j in range(segments):
        faces.append((top, ring(1, j + 1), ring(1, j)))
    for i in range(1, rings - 1):
        for j in range(segments):
            a, b, c_, d = ring(i, j), ring(i, j + 1), ring(i + 1, j), ring(i + 1, j + 1)
            faces.append((a, b, c_))
            faces.append((b, d, c_))
    for j in range(segments):
        faces.append((bottom, ring(rings - 1, j), ring(rings - 1, j + 1)))
    return verts, np.array(faces, dtype=np.int64)


def blob_displacement(seed=1, terms=6):
    """Seeded low-frequency radial displacement (the 'blob' of C3/C4, a stand-in for Blender's monkey)."""
    rng = np.random.default_rng(seed)
    amp = rng.uniform(0.03, 0.09, terms)
    kt = rng.integers(1, 5, terms)
    kp = rng.integers(1, 5, terms)
    ph = rng.uniform(0, 2 * np.pi, terms)

    def disp(theta, phi):
        return float(np.sum(amp * np.sin(kt * theta + ph) * np.cos(kp * phi + 0.5 * ph)) * np.sin(theta))
    return disp


def quad(p0, p1, p2, p3):
    """Two triangles (p0,p1,p2), (p0,p2,p3) as vertex/face arrays."""
    return np.array([p0, p1, p2, p3], dtype=np.float64), np.array([(0, 1, 2), (0, 2, 3)], dtype=np.int64)


def merge_meshes(parts):
    """parts: [(verts, faces, material_name)] -> ObjMesh pieces concatenated."""
    vs, fs, ms = [], [], []
    base = 0
    for v, f, m in parts:
        vs.append(v); fs.append(f + base); ms += [m] * len(f)
        base += len(v)
    return np.concatenate(vs), np.concatenate(fs), ms


def room_and_pedestal(cx, cz, top_y):
    """10 room triangles (floor, +x wall, -z wall, a floor step, a low curb) + 10 pedestal triangles
    (a box without bottom under the blob): the '+20' of the 988 / 63,432 triangle counts."""
    parts = []
    parts.append((*quad((cx - 8, 0, cz - 4), (cx - 8, 0, cz + 8), (cx + 5, 0, cz + 8), (cx + 5, 0, cz - 4)), "floor"))
    parts.append((*quad((cx + 5, 0, cz - 4), (cx + 5, 0, cz + 3), (cx + 5, 4, cz + 3), (cx + 5, 4, cz - 4)), "wall"))
    parts.append((*quad((cx - 8, 0, cz - 4), (cx + 5, 0, cz - 4), (cx + 5, 4, cz - 4), (cx - 8, 4, cz - 4)), "wall"))
    parts.append((*quad((cx + 2, 0.25, cz - 4), (cx + 2, 0.25, cz + 3), (cx + 5, 0.25, cz + 3), (cx + 5, 0.25, cz - 4)), "floor"))
    parts.append((*quad((cx + 2, 0, cz - 4), (cx + 2, 0, cz + 3), (cx + 2, 0.25, cz + 3), (cx + 2, 0.25, cz - 4)), "floor"))
    h, w = top_y, 0.55
    x0, x1, z0, z1 = cx - w, cx + w, cz - w, cz + w
    parts.append((*quad((x0, h, z0), (x0, h, z1), (x1, h, z1), (x1, h, z0)), "pedestal"))
    parts.append((*quad((x0, 0, z0), (x0, 0, z1), (x0, h, z1), (x0, h, z0)), "pedestal"))
    parts.append((*quad((x1, 0, z0), (x1, h, z0), (x1, h, z1), (x1, 0, z1)), "pedestal"))
    parts.append((*quad((x0, 0, z0), (x0, h, z0), (x1, h, z0), (x1, 0, z0)), "pedestal"))
    parts.append((*quad((x0, 0, z1), (x1, 0, z1), (x1, h, z1), (x0, h, z1)), "pedestal"))
    return parts


# ------------------------------------------------------------------------------------------ configs
_OBLI_CAMERA = {"lookFrom": [-4.0, 1.3, 5.5], "lookAt": [0.0, 0.8, 4.0], "vUp": [0.0, 1.0, 0.0],
                "vfov": 65.0, "aperture": 0.0, "focusDist": 3.0}            # configurations/obli.json:18-37
_OBLI_LIGHT = {"tag": "PointLight", "position": [20.0, 10.0, 20.0], "intensity": [170.0, 170.0, 150.0]}   # obli.json:46-60
_BLOB_MATERIALS = [
    ("floor", {"diffuseColor": [0.8, 0.8, 0.8]}),
    ("wall", {"diffuseColor": [0.75, 0.6, 0.5]}),
    ("pedestal", {"diffuseColor": [0.3, 0.3, 0.35]}),
    ("blob", {"diffuseColor": [0.7, 0.45, 0.2]}),
]


def blob_scene(name, segments, rings, width, height, spp, max_bounces=5, bvh_max_depth=32, leaf_threshold=4,
               antialiasing=True, copies=1):
    """'Blob on a pedestal in an open box' (C3: 22x23 -> 988 triangles; C4: 166x192 -> 63,432).
    copies > 1 lays further blobs out on an 8-wide grid of objpositions (C5's only form of instancing,
    cf. reference docs/17_config_files.md:68-77); the room is kept once."""
    v, f = uv_sphere(segments, rings, radius=0.75, center=(0.0, 0.0, 0.0), displacement=blob_displacement(1))
    meshes = []
    rv, rf, rm = merge_meshes(room_and_pedestal(0.0, 4.0, 0.45))
    bv = v + np.array([0.0, 1.25, 4.0])
    mv, mf, mm = merge_meshes([(bv, f, "blob"), (rv, rf, None)])
    mm = ["blob"] * len(f) + rm
    meshes.append(ObjMesh(mv, mf, mm, (0.0, 0.0, 0.0)))
    for k in range(1, copies):
        gx, gz = k % 8, k // 8
        meshes.append(ObjMesh(bv.copy(), f.copy(), ["blob"] * len(f), (1.9 * gx, 0.0, -1.9 * gz)))
    return SceneSpec(
        name=name, width=width, height=height, spp=spp, antialiasing=antialiasing, gamma=1.0, exposure=0.85,
        background={"tag": "SolidColor", "color": [1.0, 1.0, 1.0]},                       # obli.json:10-17
        camera=dict(_OBLI_CAMERA), max_bounces=max_bounces, bvh_max_depth=bvh_max_depth, leaf_threshold=leaf_threshold,
        materials=list(_BLOB_MATERIALS), lights=[dict(_OBLI_LIGHT)], obj_meshes=meshes)


def c1_testing(width=640, height=360, spp=1, antialiasing=False, max_bounces=1):
    """C1: configurations/testing.json camera / light / gradient (testing.json:2-65), a 32x16 UV sphere
    ('1ball', 960 triangles) at objposition [0,2,0] + 3 config spheres.  antialiasing=false and
    maxBounces<=1 make the reference deterministic: the exact-parity configuration."""
    v, f = uv_sphere(32, 16, radius=1.0)
    return SceneSpec(
        name="c1_testing", width=width, height=height, spp=spp, antialiasing=antialiasing, gamma=0.8, exposure=1.5,
        background={"tag": "Gradient", "color1": [1.0, 1.0, 1.0], "color2": [0.5, 0.7, 1.0]},
        camera={"lookFrom": [-5.0, 2.0, -5.0], "lookAt": [0.0, 2.0, 0.0], "vUp": [0.0, 1.0, 0.0], "vfov": 60.0,
                "aperture": 0.0, "focusDist": 3.0},
        max_bounces=max_bounces, bvh_max_depth=32, leaf_threshold=4,
        materials=[("red", {"diffuseColor": [0.8, 0.2, 0.2]}), ("green", {"diffuseColor": [0.2, 0.7, 0.3]}),
                   ("lamp", {"diffuseColor": [0.9, 0.9, 0.9], "emissionColor": [0.6, 0.5, 0.3]})],
        lights=[{"tag": "PointLight", "position": [20.0, 3.0, 20.0], "intensity": [0.5, 0.5, 0.5]},
                {"tag": "PointLight", "position": [-6.0, 8.0, -2.0], "intensity": [40.0, 40.0, 40.0]}],
        config_objects=[("sphere", [1.8, 1.2, -1.5], 0.7, "red"), ("sphere", [-1.6, 2.6, 1.2], 0.5, "green"),
                        ("sphere", [0.0, -100.0, 0.0], 100.0, None), ("sphere", [0.3, 3.6, -0.4], 0.35, "lamp")],
        obj_meshes=[ObjMesh(v, f, [None] * len(f), (0.0, 2.0, 0.0))])


def c2_spheres(width=1920, height=1080, spp=50, max_bounces=7):
    """C2: comp.json-style (comp.json:2-47): matte, metal (shininess 200) and glass (transmission 1,
    ior 1.5) config spheres on a 2-triangle ground quad, 2 point lights + 1 directional, gradient."""
    return SceneSpec(
        name="c2_spheres", width=width, height=height, spp=spp, antialiasing=True, gamma=1.0, exposure=1.25,
        background={"tag": "Gradient", "color1": [1.0, 1.0, 1.0], "color2": [0.5, 0.7, 1.0]},
        camera={"lookFrom": [-8.0, 2.5, 0.0], "lookAt": [0.0, 1.0, 0.0], "vUp": [0.0, 1.0, 0.0], "vfov": 40.0,
                "aperture": 0.0, "focusDist": 3.0},
        max_bounces=max_bounces, bvh_max_depth=24, leaf_threshold=5,
        materials=[("ground", {"diffuseColor": [0.6, 0.6, 0.55]}), ("red", {"diffuseColor": [0.8, 0.25, 0.2]}),
                   ("blue", {"diffuseColor": [0.2, 0.3, 0.8]}), ("yellow", {"diffuseColor": [0.85, 0.75, 0.2]}),
                   ("metal", {"diffuseColor": [0.9, 0.9, 0.9], "shininess": 200.0}),
                   ("glass", {"diffuseColor": [1.0, 1.0, 1.0], "ior": 1.5, "transmission": 1.0})],
        lights=[{"tag": "PointLight", "position": [-4.0, 8.0, 4.0], "intensity": [30.0, 28.0, 25.0]},
                {"tag": "PointLight", "position": [2.0, 6.0, -6.0], "intensity": [20.0, 22.0, 28.0]},
                {"tag": "DirectionalLight", "direction": [0.3, -1.0, 0.2], "intensity": [0.35, 0.35, 0.3]}],
        config_objects=[
            ("triangle", [-20.0, 0.0, -20.0], [-20.0, 0.0, 20.0], [20.0, 0.0, 20.0], "ground"),
            ("triangle", [-20.0, 0.0, -20.0], [20.0, 0.0, 20.0], [20.0, 0.0, -20.0], "ground"),
            ("sphere", [0.0, 1.0, 0.0], 1.0, "glass"), ("sphere", [2.2, 1.0, -2.2], 1.0, "metal"),
            ("sphere", [1.5, 0.7, 2.4], 0.7, "red"), ("sphere", [-1.5, 0.5, -1.6], 0.5, "blue"),
            ("sphere", [-2.0, 0.4, 1.3], 0.4, "yellow")])


def c3_blob988(width=1920, height=1080, spp=50, **kw):
    """C3: 988 triangles (22 segments x 23 rings = 968 + 20), obli.json camera/light, solid white."""
    return blob_scene("c3_blob988", 22, 23, width, height, spp, **kw)


def c4_blob63k(width=1920, height=1080, spp=100, **kw):
    """C4 (the metric config): 63,432 triangles (166 x 192 = 63,412 + 20), 1920x1080, 100 spp,
    maxBounces 5, leafThreshold 4 (obli.json:38-43; bvhMaxDepth 32 here, 10 as shipped is a variant)."""
    return blob_scene("c4_blob63k", 166, 192, width, height, spp, **kw)


def sunset_sky(w=2048, h=1024, seed=7):
    """A synthetic equirect HDR sky (RT_SKY_RGBF, row 0 = zenith) standing in for the .hdr obli.json names
    (configurations/obli.json `skyTexture`: the author's file is not in the reference, .gitignore:28-33): warm horizon,
    blue zenith, dark ground, a sun disk of radiance ~60 and seeded texel noise, so neighbouring texels differ and the
    nearest-texel addressing (SkySphere.hs:37-41) is exercised.  With a sky every terminating path runs sampleSkySphere
    (normalize, atan2, asin, one texel fetch) instead of returning the solid background (Trace.hs:78-83)."""
    rng = np.random.default_rng(seed)
    v = (np.arange(h, dtype=np.float64) + 0.5) / h                      # 0 = zenith, 1 = nadir
    u = (np.arange(w, dtype=np.float64) + 0.5) / w
    elev = (0.5 - v)[:, None] * np.pi                                   # +pi/2 .. -pi/2
    az = (u[None, :] - 0.5) * 2 * np.pi
    up = np.clip(np.sin(elev), 0.0, 1.0)
    horizon = np.exp(-np.abs(elev) * 6.0)
    img = np.empty((h, w, 3), np.float64)
    img[..., 0] = 0.25 + 0.15 * up + 0.9 * horizon + 0 * az
    img[..., 1] = 0.35 + 0.25 * up + 0.45 * horizon + 0 * az
    img[..., 2] = 0.55 + 0.45 * up + 0.15 * horizon + 0 * az
    img[elev[:, 0] < 0] *= 0.35                                          # ground half
    sun_el, sun_az = 0.18, 0.9
    ang = np.arccos(np.clip(np.sin(elev) * np.sin(sun_el) + np.cos(elev) * np.cos(sun_el) * np.cos(az - sun_az), -1, 1))
    img += (60.0 * (ang < 0.03) + 2.5 * np.exp(-ang * 9.0))[..., None] * np.array([1.0, 0.85, 0.6])
    img *= 1.0 + 0.04 * rng.standard_normal((h, w, 1))
    return "rgbf", np.ascontiguousarray(np.clip(img, 0.0, None).astype(np.float32))


def c4_shipped(width=1920, height=1080, spp=100, sky_size=(2048, 1024), **kw):
    """The metric config AS SHIPPED in configurations/obli.json: bvhMaxDepth 10 (obli.json:41: 1,024 leaves of ~62
    triangles in the reference's tree) and an HDR sky texture (obli.json `skyTexture`; synthetic here, see sunset_sky),
    which takes priority over the solid background (Trace.hs:78-83).  bench.py's default workload."""
    kw.setdefault("bvh_max_depth", 10)
    s = blob_scene("c4_blob63k_shipped", 166, 192, width, height, spp, **kw)
    s.sky = sunset_sky(*sky_size)
    return s


def c5_multi(width=3840, height=2160, spp=256, copies=32, **kw):
    """C5: 32 copies of the C4 blob as separate objFiles entries (2,029,184 + 20 triangles), 4K, 256 spp."""
    s = blob_scene("c5_multi", 166, 192, width, height, spp, copies=copies, **kw)
    s.camera = {"lookFrom": [-6.0, 4.5, 9.0], "lookAt": [6.0, 0.8, -1.0], "vUp": [0.0, 1.0, 0.0], "vfov": 65.0,
                "aperture": 0.0, "focusDist": 3.0}
    return s


def sky_gradient_texture(w=64, h=32, fmt="rgbf"):
    """A small deterministic equirect sky (no asset ships with the reference)."""
    y = np.linspace(0, 1, h)[:, None, None]
    x = np.linspace(0, 1, w)[None, :, None]
    img = np.concatenate([0.4 + 0.6 * y + 0 * x, 0.5 + 0.3 * x + 0 * y, 1.0 - 0.5 * y + 0 * x], axis=2)
    if fmt == "rgb8":
        return "rgb8", np.ascontiguousarray((img * 255).astype(np.uint8))
    img[2, 5] = [12.0, 11.0, 9.0]   # an HDR 'sun' texel
    return "rgbf", np.ascontiguousarray(img.astype(np.float32))
