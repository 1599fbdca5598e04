"""Host mirror of loadConfig / parseSceneObjects / parseObj / parseMaterial (librthost) on files written in the
reference's schema: the assembled `allObjects`, material ids and lights must equal the in-memory SceneSpec, and
the reference's parsing quirks must be reproduced."""
import ctypes as C
import json
import os

import numpy as np
import pytest

from raytracer_amd import _ffi as F
from raytracer_amd import api, scenes


def load(path):
    h = C.c_void_p()
    F.check(F.host().rth_config_load(str(path).encode(), C.byref(h)), "host")
    v = F.rth_config_view()
    F.check(F.host().rth_config_get(h, C.byref(v)), "host")
    n = v.n_objs
    res = {"view": v, "handle": h,
           "kind": np.ctypeslib.as_array(v.obj_kind, shape=(n,)).copy() if n else np.zeros(0, np.int32),
           "data": np.ctypeslib.as_array(v.obj_data, shape=(n * 9,)).reshape(n, 9).copy() if n else np.zeros((0, 9)),
           "material": np.ctypeslib.as_array(v.obj_material, shape=(n,)).copy() if n else np.zeros(0, np.int32)}
    return res


@pytest.mark.parametrize("mk", [scenes.c1_testing, scenes.c2_spheres, scenes.c3_blob988])
def test_roundtrip_equals_scenespec(tmp_path, mk):
    spec = mk()
    jp = scenes.write_reference_files(spec, str(tmp_path))
    r = load(jp)
    k, d, m = spec.all_objects()
    assert np.array_equal(r["kind"], k) and np.array_equal(r["material"], m)
    assert np.array_equal(r["data"], d)                     # repr() round-trips doubles exactly
    v = r["view"]
    assert (v.params.width, v.params.height, v.params.samples_per_pixel, v.params.max_bounces) == (spec.width, spec.height, spec.spp, spec.max_bounces)
    assert v.params.antialiasing == int(spec.antialiasing) and v.params.gamma == spec.gamma and v.params.exposure == spec.exposure
    assert (v.bvh_max_depth, v.leaf_threshold, v.n_triangles) == (spec.bvh_max_depth, spec.leaf_threshold, spec.n_triangles())
    assert bytes(v.camera) == bytes(api.default_camera(spec.camera, spec.width / spec.height))
    mats, nm = api.make_materials(spec)
    assert v.n_materials == nm
    for i in range(nm):
        assert bytes(v.materials[i]) == bytes(mats[i])
    lights, nl = api.make_lights(spec)
    assert v.n_lights == nl and all(bytes(v.lights[i]) == bytes(lights[i]) for i in range(nl))
    F.host().rth_config_destroy(r["handle"])


def base_config(scene):
    return {"image": {"width": 8, "height": 4, "samplesPerPixel": 1, "antialiasing": False, "gamma": 1.0, "exposure": 1.0},
            "background": {"tag": "SolidColor", "color": [1, 1, 1]},
            "camera": {"lookFrom": [0, 0, 5], "lookAt": [0, 0, 0], "vUp": [0, 1, 0], "vfov": 60.0, "aperture": 0.0, "focusDist": 1.0},
            "raytracer": {"maxBounces": 2, "useBVH": True, "bvhMaxDepth": 8, "leafThreshold": 2},
            "scene": scene}


def test_obj_and_mtl_quirks(tmp_path):
    (tmp_path / "m.mtl").write_text(
        "newmtl zeta\nKd 0.1 0.2 0.3\nNs 250\n"
        "newmtl alpha\nKd 1 0 0\nKe 2 2 2\nNi 1.45\nillum 7\n"
        "newmtl beta\nNi 1.0\nillum 9\n")
    (tmp_path / "a.obj").write_text(
        "mtllib m.mtl\n"
        "v 0 0 0\nv 1 0 0\nv 1 1 0\nv 0 1 0\nvn 0 0 1\nvt 0 0\n"
        "usemtl alpha\nf 1/1/1 2/2/1 3/3/1\n"        # 3-gon: emitted as (v0, v2, v1)
        "usemtl zeta\nf 1 2 3 4\n"                    # n-gon: fan (v0, vi, vi+1), un-flipped
        "usemtl unknown\nf 1 3 4\n")                  # unknown material name -> id 0
    cfg = base_config({"materials": [["json", {"diffuseColor": [0.5, 0.5, 0.5]}]],
                       "objFiles": [{"path": str(tmp_path / "a.obj"), "objposition": [10, 0, 0]}],
                       "objects": [{"tag": "TriangleObj", "contents": [[0, 0, 1], [1, 0, 1], [0, 1, 1], [1, 1, 1], "json"]},
                                   {"tag": "SphereObj", "contents": [[0, 0, 0], 1.5, [1, 1, 1], None]},
                                   {"tag": "PlaneObj", "contents": [[0, -1, 0], [0, 1, 0], [1, 1, 1], "beta"]}],
                       "lights": [{"tag": "DirectionalLight", "direction": [0, -1, 0], "intensity": [1, 1, 1]}]})
    jp = tmp_path / "c.json"
    jp.write_text(json.dumps(cfg))
    r = load(jp)
    # allObjects = configTriangles ++ objTriangles ++ ALL config objects (the config triangle twice)
    assert r["kind"].tolist() == [0, 0, 0, 0, 0, 0, 1, 2]
    d = r["data"]
    assert d[0].tolist() == [0, 0, 1, 1, 0, 1, 0, 1, 1]
    assert d[1].tolist() == [10, 0, 0, 11, 1, 0, 11, 0, 0]          # (v0, v2, v1) + offset
    assert d[2].tolist() == [10, 0, 0, 11, 0, 0, 11, 1, 0]          # fan, un-flipped
    assert d[3].tolist() == [10, 0, 0, 11, 1, 0, 10, 1, 0]
    assert d[6, :4].tolist() == [0, 0, 0, 1.5] and d[7, :6].tolist() == [0, -1, 0, 0, 1, 0]
    # ids: JSON from 1; MTL from 0 in NAME order (alpha 0, beta 1, zeta 2); JSON shadows MTL on id clashes
    assert r["material"].tolist() == [1, 0, 2, 2, 0, 1, 0, 1]
    v = r["view"]
    mats = {v.materials[i].id: v.materials[i] for i in range(v.n_materials)}
    assert sorted(mats) == [0, 1, 2]
    assert list(mats[1].diffuse) == [0.5, 0.5, 0.5]                 # id 1 is the JSON material, MTL "beta" is shadowed
    a = mats[0]                                                     # alpha: illum 7 and Ni > 1 -> transmission 1
    assert a.flags == (F.RT_MAT_HAS_EMISSION | F.RT_MAT_HAS_IOR | F.RT_MAT_HAS_TRANSMISSION) and a.transmission == 1.0 and a.ior == 1.45
    z = mats[2]
    assert z.flags == F.RT_MAT_HAS_SHININESS and z.shininess == 250 and list(z.diffuse) == [0.1, 0.2, 0.3]
    assert v.n_triangles == 5 and v.n_lights == 1 and v.lights[0].kind == F.RT_LIGHT_DIRECTIONAL
    F.host().rth_config_destroy(r["handle"])


def test_errors_mirror_the_reference(tmp_path):
    def fails(cfg_text, needle):
        p = tmp_path / "bad.json"
        p.write_text(cfg_text)
        h = C.c_void_p()
        rc = F.host().rth_config_load(str(p).encode(), C.byref(h))
        assert rc == F.RT_ERR_INVALID
        assert needle in F.host().rth_last_error().decode(), F.host().rth_last_error()

    h = C.c_void_p()
    assert F.host().rth_config_load(str(tmp_path / "missing.json").encode(), C.byref(h)) == F.RT_ERR_INVALID
    assert "Config file not found" in F.host().rth_last_error().decode()          # Config.hs:132
    fails("{", "Error parsing config file")
    cfg = base_config({})
    del cfg["image"]["gamma"]                                                      # every ImageSettings field is required
    fails(json.dumps(cfg), "gamma")
    cfg = base_config({"objects": [{"tag": "SphereObj", "contents": [[0, 0], 1.0, [1, 1, 1], None]}]})
    fails(json.dumps(cfg), "Vec3 must contain exactly three valid numbers")        # Vec3.hs:44
    (tmp_path / "bad.obj").write_text("v 0 0\nf 1 2 3\n")
    cfg = base_config({"objFiles": [{"path": str(tmp_path / "bad.obj"), "objposition": [0, 0, 0]}]})
    fails(json.dumps(cfg), "Invalid vertex format in .obj file")                   # Object.hs:40
    (tmp_path / "bad2.obj").write_text("v 0 0 0\nv 1 0 0\nf 1 2\n")
    cfg = base_config({"objFiles": [{"path": str(tmp_path / "bad2.obj"), "objposition": [0, 0, 0]}]})
    fails(json.dumps(cfg), "Invalid face format in .obj file")                     # Object.hs:67
    # optional scene fields may all be absent (Maybe): an empty scene parses, the BVH build then refuses it
    p = tmp_path / "empty.json"
    p.write_text(json.dumps(base_config({})))
    r = load(p)
    assert r["view"].n_objs == 0
    F.host().rth_config_destroy(r["handle"])


def test_ppm_body_is_byte_exact(tmp_path):
    rgb = np.array([[[0, 5, 12], [100, 255, 99]], [[7, 70, 200], [1, 2, 3]]], np.uint8)
    p = tmp_path / "o.ppm"
    F.check(F.host().rth_write_ppm(str(p).encode(), 2, 2, rgb.ctypes.data_as(F.c_uint8_p)), "host")
    # "P3\n" ++ show w ++ " " ++ show h ++ "\n255\n" ++ unlines (map showPixel ...)   ImageGenerator.hs:100, RenderWorker.hs:72,108
    assert p.read_text() == "P3\n2 2\n255\n0 5 12\n100 255 99\n7 70 200\n1 2 3\n"
    need = F.host().rth_format_ppm(2, 2, rgb.ctypes.data_as(F.c_uint8_p), None, 0)
    assert need == len(p.read_text())


def test_sky_loader(tmp_path):
    # flat (non-RLE) Radiance file, 2x1: RGBE (128,64,32,129) -> m * 2^(e-136)  (JuicyPixels' toFloat as recalled: rthost_sky.cpp)
    hdr = b"#?RADIANCE\nFORMAT=32-bit_rle_rgbe\n\n-Y 1 +X 2\n" + bytes([128, 64, 32, 129, 0, 0, 0, 0])
    p = tmp_path / "s.hdr"
    p.write_bytes(hdr)
    fmt, w, h, px = C.c_int32(), C.c_int32(), C.c_int32(), C.c_void_p()
    F.check(F.host().rth_sky_load(str(p).encode(), C.byref(fmt), C.byref(w), C.byref(h), C.byref(px)), "host")
    a = np.ctypeslib.as_array(C.cast(px, C.POINTER(C.c_float)), shape=(6,)).copy()
    F.host().rth_sky_free(px)
    assert (fmt.value, w.value, h.value) == (F.RT_SKY_RGBF, 2, 1)
    assert a.tolist() == [128 / 128, 64 / 128, 32 / 128, 0, 0, 0]
    q = tmp_path / "s.ppm"
    q.write_bytes(b"P6\n2 1\n255\n" + bytes([1, 2, 3, 4, 5, 6]))
    F.check(F.host().rth_sky_load(str(q).encode(), C.byref(fmt), C.byref(w), C.byref(h), C.byref(px)), "host")
    b = np.ctypeslib.as_array(C.cast(px, C.POINTER(C.c_uint8)), shape=(6,)).copy()
    F.host().rth_sky_free(px)
    assert fmt.value == F.RT_SKY_RGB8 and b.tolist() == [1, 2, 3, 4, 5, 6]


def test_edge_inputs_follow_the_reference(tmp_path):
    """aeson's number grammar (no hex floats / inf / nan), lazy OBJ vertices (a malformed `v` line only fails when
    a face uses it, Object.hs:19,39-40,52-54), `read :: Int` for illum ("2.0" is no parse)."""
    def load(cfg_text):
        p = tmp_path / "c.json"
        p.write_text(cfg_text)
        h = C.c_void_p()
        rc = F.host().rth_config_load(str(p).encode(), C.byref(h))
        if rc == 0:
            F.host().rth_config_destroy(h)
        return rc
    import json
    good = json.dumps(base_config({}))
    assert load(good) == 0
    for bad_number in ("0x1p3", "-inf", "-nan", "+1", "1.", ".5", "01"):
        assert load(good.replace('"gamma": 1.0', f'"gamma": {bad_number}')) != 0, bad_number
    assert load(good.replace('"gamma": 1.0', '"gamma": -1.5e+2')) == 0
    obj = tmp_path / "m.obj"
    obj.write_text("v 0 0 0\nv 1 0 0\nv 0 1 0\nv 9 9 9 9\nv oops 1 2\nf 1 2 3\n")             # vertices 4 and 5 are malformed but unused
    cfg = base_config({"objFiles": [{"path": str(obj), "objposition": [0, 0, 0]}]})
    assert load(json.dumps(cfg)) == 0
    obj.write_text("v 0 0 0\nv 1 0 0\nv 0 1 0\nv 9 9 9 9\nf 1 2 4\n")                          # ... and used
    assert load(json.dumps(cfg)) != 0
    (tmp_path / "m.mtl").write_text("newmtl a\nKd 1 0 0\nNi 1.5\nillum 2.0\n")
    obj.write_text("mtllib m.mtl\nv 0 0 0\nv 1 0 0\nv 0 1 0\nusemtl a\nf 1 2 3\n")
    assert load(json.dumps(cfg)) != 0
    (tmp_path / "m.mtl").write_text("newmtl a\nKd 1 0 0\nNi 1.5\nillum 7\n")
    assert load(json.dumps(cfg)) == 0


def _load_ldr(path):
    fmt, w, h, px = C.c_int32(), C.c_int32(), C.c_int32(), C.c_void_p()
    F.check(F.host().rth_sky_load(str(path).encode(), C.byref(fmt), C.byref(w), C.byref(h), C.byref(px)), "host")
    a = np.ctypeslib.as_array(C.cast(px, C.POINTER(C.c_uint8)), shape=(h.value, w.value, 3)).copy()
    F.host().rth_sky_free(px)
    assert fmt.value == F.RT_SKY_RGB8
    return a


def _png_bytes(width, height, depth, ctype, rows, palette=None, interlace=False, filters=None):
    """A PNG written by hand (zlib + CRCs): `rows` = per-scanline raw sample bytes of the WHOLE image; for Adam7 the seven
    reduced images are cut out of it here (whole-byte pixels only); `filters` picks a filter type per scanline."""
    import struct
    import zlib

    def chunk(t, d):
        return struct.pack(">I", len(d)) + t + d + struct.pack(">I", zlib.crc32(t + d) & 0xFFFFFFFF)

    nch = {0: 1, 2: 3, 3: 1, 4: 2, 6: 4}[ctype]
    bpp = max(1, nch * depth // 8)

    def filt(ft, cur, prev):
        out = bytearray(len(cur))
        for i, v in enumerate(cur):
            a = cur[i - bpp] if i >= bpp else 0
            b = prev[i]
            c = prev[i - bpp] if i >= bpp else 0
            if ft == 4:
                p = a + b - c
                pa, pb, pc = abs(p - a), abs(p - b), abs(p - c)
                pred = a if (pa <= pb and pa <= pc) else (b if pb <= pc else c)
            else:
                pred = [0, a, b, (a + b) >> 1][ft]
            out[i] = (v - pred) & 255
        return bytes([ft]) + bytes(out)

    def encode(img_rows):
        raw, prev = b"", bytes(len(img_rows[0])) if img_rows else b""
        for y, r in enumerate(img_rows):
            ft = (filters[y % len(filters)] if filters else 0)
            raw += filt(ft, r, prev)
            prev = r
        return raw

    if not interlace:
        raw = encode(rows)
    else:
        raw = b""
        x0, y0, dx, dy = [0, 4, 0, 2, 0, 1, 0], [0, 0, 4, 0, 2, 0, 1], [8, 8, 4, 4, 2, 2, 1], [8, 8, 8, 4, 4, 2, 2]
        for p in range(7):
            sub = [b"".join(rows[y][x * bpp:(x + 1) * bpp] for x in range(x0[p], width, dx[p])) for y in range(y0[p], height, dy[p])]
            sub = [r for r in sub if r]
            if sub:
                raw += encode(sub)
    out = b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", width, height, depth, ctype, 0, 0, 1 if interlace else 0))
    if palette is not None:
        out += chunk(b"PLTE", bytes(palette))
    half = len(zlib.compress(raw)) // 2
    z = zlib.compress(raw)
    return out + chunk(b"IDAT", z[:half]) + chunk(b"IDAT", z[half:]) + chunk(b"IEND", b"")      # two IDAT chunks: they concatenate


def test_sky_loader_ldr_formats(tmp_path):
    """SkySphere.hs:28-29: every non-.hdr sky goes through JuicyPixels' readImage + convertRGB8.  PNG, BMP and TGA files written
    by Pillow (an independent encoder) and by a hand-made PNG writer (16-bit, Adam7, every filter type, low bit depths) must decode to
    what convertRGB8 documents: alpha dropped, grey replicated, the HIGH byte of 16-bit samples, palettes expanded, 1/2/4-bit grey
    scaled as i * 255 div (2^depth - 1).  (PARITY UNPINNED against JuicyPixels itself, like the .hdr rule.)"""
    Image = pytest.importorskip("PIL.Image")
    rng = np.random.default_rng(5)
    W, H = 19, 11
    rgb = rng.integers(0, 256, (H, W, 3), dtype=np.uint8)
    rgba = np.concatenate([rgb, rng.integers(0, 256, (H, W, 1), dtype=np.uint8)], axis=2)
    grey = rng.integers(0, 256, (H, W), dtype=np.uint8)
    # ---- Pillow-written files
    Image.fromarray(rgb, "RGB").save(tmp_path / "a.png")
    assert np.array_equal(_load_ldr(tmp_path / "a.png"), rgb)
    Image.fromarray(rgba, "RGBA").save(tmp_path / "b.png")
    assert np.array_equal(_load_ldr(tmp_path / "b.png"), rgb)                                  # alpha dropped, not blended
    Image.fromarray(grey, "L").save(tmp_path / "c.png")
    assert np.array_equal(_load_ldr(tmp_path / "c.png"), np.repeat(grey[:, :, None], 3, axis=2))
    pal = Image.fromarray(rgb, "RGB").quantize(colors=16)                                      # 4-bit palette image
    pal.save(tmp_path / "d.png")
    assert np.array_equal(_load_ldr(tmp_path / "d.png"), np.asarray(pal.convert("RGB")))
    one = Image.fromarray((grey > 127).astype(np.uint8) * 255, "L").convert("1")               # 1-bit grey
    one.save(tmp_path / "e.png")
    assert np.array_equal(_load_ldr(tmp_path / "e.png"), np.repeat(((grey > 127) * 255).astype(np.uint8)[:, :, None], 3, axis=2))
    Image.fromarray(rgb, "RGB").save(tmp_path / "f.bmp")
    assert np.array_equal(_load_ldr(tmp_path / "f.bmp"), rgb)
    Image.fromarray(rgba, "RGBA").save(tmp_path / "g.bmp")
    assert np.array_equal(_load_ldr(tmp_path / "g.bmp"), rgb)
    Image.fromarray(rgb, "RGB").save(tmp_path / "h.tga")
    assert np.array_equal(_load_ldr(tmp_path / "h.tga"), rgb)
    Image.fromarray(rgba, "RGBA").save(tmp_path / "i.tga", compression="tga_rle")
    assert np.array_equal(_load_ldr(tmp_path / "i.tga"), rgb)
    Image.fromarray(grey, "L").save(tmp_path / "j.tga")
    assert np.array_equal(_load_ldr(tmp_path / "j.tga"), np.repeat(grey[:, :, None], 3, axis=2))
    # ---- hand-made PNGs: 16-bit RGB with every filter type; Adam7; 16-bit grey + alpha; 2-bit grey
    v16 = rng.integers(0, 65536, (H, W, 3), dtype=np.uint16)
    rows = [v16[y].astype(">u2").tobytes() for y in range(H)]
    (tmp_path / "k.png").write_bytes(_png_bytes(W, H, 16, 2, rows, filters=[0, 1, 2, 3, 4]))
    assert np.array_equal(_load_ldr(tmp_path / "k.png"), (v16 >> 8).astype(np.uint8))         # from16to8: the high byte
    rows8 = [rgba[y].tobytes() for y in range(H)]
    (tmp_path / "l.png").write_bytes(_png_bytes(W, H, 8, 6, rows8, interlace=True, filters=[4, 3, 1]))
    assert np.array_equal(_load_ldr(tmp_path / "l.png"), rgb)
    ga16 = rng.integers(0, 65536, (H, W, 2), dtype=np.uint16)
    (tmp_path / "m.png").write_bytes(_png_bytes(W, H, 16, 4, [ga16[y].astype(">u2").tobytes() for y in range(H)], filters=[2, 4]))
    assert np.array_equal(_load_ldr(tmp_path / "m.png"), np.repeat((ga16[:, :, :1] >> 8).astype(np.uint8), 3, axis=2))
    g2 = rng.integers(0, 4, (H, W), dtype=np.uint8)
    packed = []
    for y in range(H):
        bits = 0
        row = bytearray((W * 2 + 7) // 8)
        for x in range(W):
            row[(2 * x) >> 3] |= int(g2[y, x]) << (6 - ((2 * x) & 7))
        packed.append(bytes(row))
    (tmp_path / "n.png").write_bytes(_png_bytes(W, H, 2, 0, packed, filters=[0, 2]))
    assert np.array_equal(_load_ldr(tmp_path / "n.png"), np.repeat((g2.astype(np.int64) * 255 // 3).astype(np.uint8)[:, :, None], 3, axis=2))
    # ---- what this host does not decode fails loudly
    Image.fromarray(rgb, "RGB").save(tmp_path / "o.jpg")
    fmt, w, h, px = C.c_int32(), C.c_int32(), C.c_int32(), C.c_void_p()
    assert F.host().rth_sky_load(str(tmp_path / "o.jpg").encode(), C.byref(fmt), C.byref(w), C.byref(h), C.byref(px)) == F.RT_ERR_UNSUPPORTED
    (tmp_path / "p.png").write_bytes((tmp_path / "a.png").read_bytes()[:60])
    assert F.host().rth_sky_load(str(tmp_path / "p.png").encode(), C.byref(fmt), C.byref(w), C.byref(h), C.byref(px)) == F.RT_ERR_INVALID
