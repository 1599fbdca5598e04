"""A SECOND, independent restatement of the reference's hot path — TEST INFRASTRUCTURE ONLY.

oracle/oracle.cpp is the checker the GPU is compared with; nothing in the reference pins it (Haskell, no ghc here,
no tests, no assets: "parity unpinned").  This file is a separate reading of the same Haskell
sources, written after the C++ one and without looking at it: recursive, list-shaped, function for function like
the reference (file:line cited per function, relative to /root/reference/app).  tests/test_oracle_pyref.py requires the
two restatements to agree bit for bit on tiny frames; a slip in either reading shows up as a difference.
Pure-Python loops: tiny frames only.  Shares with the C++ oracle only the documented RNG (Philox4x32-10 keyed by
(pixel, sample, stream)), which stands in for the reference's entropy-seeded generators.
"""
import math

INF = float("inf")


# ---------------------------------------------------------------- Core/Vec3.hs:53-111
def add(a, b): return (a[0] + b[0], a[1] + b[1], a[2] + b[2])
def sub(a, b): return (a[0] - b[0], a[1] - b[1], a[2] - b[2])
def dot(a, b): return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]
def cross(a, b): return (a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2], a[0] * b[1] - a[1] * b[0])
def scale(s, v): return (s * v[0], s * v[1], s * v[2])
def mul(a, b): return (a[0] * b[0], a[1] * b[1], a[2] * b[2])
def negate_v(v): return (-v[0], -v[1], -v[2])
def v_length(v): return math.sqrt(dot(v, v))
def fdiv(a, b):
    """IEEE division, as `/` on Double is under GHC: x / 0 is +-inf or NaN where Python raises ZeroDivisionError."""
    try:
        return a / b
    except ZeroDivisionError:
        if a != a or a == 0:
            return math.nan
        return math.copysign(math.inf, a) * math.copysign(1.0, b)


def normalize(v): return scale(fdiv(1, v_length(v)), v)                      # :78-79
def reflect(v, n): return sub(v, scale(2 * dot(v, n), n))              # :98-99


def refract(v, n, eta):                                                # :102-107
    cos_theta = -dot(v, n)
    perp = scale(eta, add(v, scale(cos_theta, n)))
    par = scale(-math.sqrt(abs(1.0 - dot(perp, perp))), n)
    return add(perp, par)


def at(ray, t): return add(ray[0], scale(t, ray[1]))                   # Core/Ray.hs:18-19


# ---------------------------------------------------------------- Hittable/Class.hs:22-26
def set_face_normal(ray, outward):
    front = dot(ray[1], outward) < 0
    return (outward if front else negate_v(outward)), front


def contains(iv, x): return x >= iv[0] and x <= iv[1]                  # Utils/Interval.hs:20-21


# ---------------------------------------------------------------- primitives: ("tri", p0, p1, p2, mat, id) | ("sph", c, r, mat, id) | ("pl", p, n, mat, id)
def bounding_box(o):
    if o[0] == "tri":                                                  # Triangle.hs:24-28
        a, b, c = o[1], o[2], o[3]
        mn = tuple(min(min(a[k], b[k]), c[k]) for k in range(3))
        mx = tuple(max(max(a[k], b[k]), c[k]) for k in range(3))
        pad = (1e-8, 1e-8, 1e-8)
        return sub(mn, pad), add(mx, pad)
    if o[0] == "sph":                                                  # Sphere.hs:20-22
        r = (o[2], o[2], o[2])
        return sub(o[1], r), add(o[1], r)
    return (-1e6, -1e-3, -1e6), (1e6, 1e-3, 1e6)                       # Plane.hs:20


def hit_prim(o, ray, iv):
    org, d = ray
    if o[0] == "tri":                                                  # Triangle.hs:30-64
        p0, p1, p2 = o[1], o[2], o[3]
        e1, e2 = sub(p1, p0), sub(p2, p0)
        h = cross(d, e2)
        a = dot(e1, h)
        if abs(a) < 1e-8:
            return None
        f = fdiv(1.0, a)
        s = sub(org, p0)
        u = f * dot(s, h)
        if u < 0.0 or u > 1.0:
            return None
        q = cross(s, e1)
        v = f * dot(d, q)
        if v < 0.0 or u + v > 1.0:
            return None
        t = f * dot(e2, q)
        nrm, front = set_face_normal(ray, normalize(cross(e1, e2)))
        if not contains(iv, t):
            return None
        return {"point": at(ray, t), "normal": nrm, "t": t, "front": front, "mat": o[4], "id": o[5]}
    if o[0] == "sph":                                                  # Sphere.hs:24-40
        c, r = o[1], o[2]
        oc = sub(org, c)
        a = dot(d, d)
        h = dot(oc, d)
        disc = h * h - a * (dot(oc, oc) - r * r)
        if disc < 0:
            return None
        t1 = fdiv(-h - math.sqrt(disc), a)                                # the near root only
        if not contains(iv, t1):
            return None
        p = at(ray, t1)
        nrm, front = set_face_normal(ray, scale(fdiv(1.0, r), sub(p, c)))
        return {"point": p, "normal": nrm, "t": t1, "front": front, "mat": o[3], "id": o[4]}
    p, n = o[1], o[2]                                                  # Plane.hs:21-32
    denom = dot(d, n)
    if not abs(denom) > 1e-6:
        return None
    t = fdiv(dot(sub(p, org), n), denom)
    if not contains(iv, t):
        return None
    nrm, front = set_face_normal(ray, n)
    return {"point": at(ray, t), "normal": nrm, "t": t, "front": front, "mat": o[3], "id": o[4]}


# ---------------------------------------------------------------- Hittable/BoundingBox.hs:14-37
def surrounding_box(a, b):
    return tuple(min(a[0][k], b[0][k]) for k in range(3)), tuple(max(a[1][k], b[1][k]) for k in range(3))


def hit_aabb_interval(box, ray, iv):
    (lo, hi), (org, d) = box, ray

    def check_axis(o, dd, mn, mx):
        if abs(dd) < 1e-8:
            return None if (o < mn or o > mx) else (-INF, INF)
        inv = fdiv(1.0, dd)
        t0, t1 = (mn - o) * inv, (mx - o) * inv
        return (min(t0, t1), max(t0, t1))

    ax = [check_axis(org[k], d[k], lo[k], hi[k]) for k in range(3)]
    if ax[0] is None or ax[1] is None or ax[2] is None:
        return None
    t_enter = max(ax[0][0], ax[1][0], ax[2][0], iv[0])
    t_exit = min(ax[0][1], ax[1][1], ax[2][1], iv[1])
    return (t_enter, t_exit) if t_enter <= t_exit else None


# ---------------------------------------------------------------- Hittable/BVH.hs:29-103 (build), :120-146 (hit)
def construct_bvh(leaf_threshold, max_depth, objects):
    cached = [(o, bounding_box(o)) for o in objects]

    def boxes_union(boxes):
        if not boxes:
            raise ValueError("Cannot compute bounding box of empty list")
        acc = boxes[0]
        for b in reversed(boxes[1:]):                                  # foldr; min / max are exact, the order cannot matter
            acc = surrounding_box(b, acc)
        return acc

    def build(objs, depth):
        n = len(objs)
        if n <= leaf_threshold or depth >= max_depth:
            return ("leaf", [o for o, _ in objs], boxes_union([b for _, b in objs]))
        mn = [INF, INF, INF]; mx = [-INF, -INF, -INF]
        for _, (lo, hi) in objs:
            for k in range(3):
                mn[k] = min(lo[k], mn[k]); mx[k] = max(hi[k], mx[k])
        ext = [mx[k] - mn[k] for k in range(3)]
        axis = 0 if (ext[0] > ext[1] and ext[0] > ext[2]) else (1 if ext[1] > ext[2] else 2)   # chooseSplitAxis :51-71
        srt = sorted(objs, key=lambda ob: (0.5 * (ob[1][0][axis] + ob[1][1][axis])))            # sortOn: stable, :74-80
        half = n // 2
        left, right = build(srt[:half], depth + 1), build(srt[half:], depth + 1)
        return ("node", left, right, surrounding_box(right[-1], left[-1]))

    return build(cached, 0)


def closest_hit_list(hits):                                            # Utils/HitHelpers.hs:11-16 (foldr)
    acc = None
    for x in reversed(hits):
        if x is None:
            continue
        acc = x if acc is None else (x if x["t"] < acc["t"] else acc)
    return acc


def hit_bvh(node, ray, iv):
    if node[0] == "leaf":
        return closest_hit_list([hit_prim(o, ray, iv) for o in node[1]])
    left, right = node[1], node[2]
    li, ri = hit_aabb_interval(left[-1], ray, iv), hit_aabb_interval(right[-1], ray, iv)
    if li is None and ri is None:
        return None
    if ri is None:
        return hit_bvh(left, ray, li)
    if li is None:
        return hit_bvh(right, ray, ri)
    if li[0] < ri[0]:
        first, i_first, second, i_second = left, li, right, ri
    else:
        first, i_first, second, i_second = right, ri, left, li
    first_hit = hit_bvh(first, ray, i_first)
    new_max = i_second[1] if first_hit is None else first_hit["t"]
    second_hit = hit_bvh(second, ray, (i_second[0], new_max))
    if second_hit is None:
        return first_hit
    if first_hit is None:
        return second_hit
    return first_hit if first_hit["t"] < second_hit["t"] else second_hit


# ---------------------------------------------------------------- RNG stand-in
def philox4x32_10(ctr, key):
    c0, c1, c2, c3 = ctr
    k0, k1 = key
    for _ in range(10):
        p0, p1 = 0xD2511F53 * c0, 0xCD9E8D57 * c2
        c0, c1, c2, c3 = ((p1 >> 32) ^ c1 ^ k0) & 0xFFFFFFFF, p1 & 0xFFFFFFFF, ((p0 >> 32) ^ c3 ^ k1) & 0xFFFFFFFF, p0 & 0xFFFFFFFF
        k0, k1 = (k0 + 0x9E3779B9) & 0xFFFFFFFF, (k1 + 0xBB67AE85) & 0xFFFFFFFF
    return c0, c1, c2, c3


def u01(w): return (w + 0.5) * (1.0 / 4294967296.0)
def um11(w): return (w + 0.5) * (1.0 / 2147483648.0) - 1.0


# ---------------------------------------------------------------- Rendering/Light.hs:16-49
def compute_lighting(rec, lights, world):
    hit_point, hit_normal = rec["point"], normalize(rec["normal"])

    def visible(light):
        if light["kind"] == "point":
            to_light = normalize(sub(light["vec"], hit_point))
            dist = v_length(sub(light["vec"], hit_point))
            if hit_bvh(world, (hit_point, to_light), (0.001, dist - 0.01)) is not None:
                return (0.0, 0.0, 0.0)
            power = max(0, dot(hit_normal, to_light))
            return scale(power * fdiv(1, dist * dist), light["intensity"])
        to_light = normalize(scale(-1, light["vec"]))
        if hit_bvh(world, (hit_point, to_light), (0.001, 10000)) is not None:
            return (0.0, 0.0, 0.0)
        return scale(max(0, dot(hit_normal, to_light)), light["intensity"])

    acc = (0.0, 0.0, 0.0)
    for c in reversed([visible(l) for l in lights]):                   # foldr V.add (Vec3 0 0 0)
        acc = add(c, acc)
    return acc


# ---------------------------------------------------------------- Rendering/Trace.hs:19-94
def schlick(cosine, ref_idx):
    r0 = fdiv(1 - ref_idx, 1 + ref_idx)
    r0sq = r0 * r0
    return r0sq + (1 - r0sq) * hs_pow(1 - cosine, 5.0)


def hs_floor_int(x):
    """floor :: Double -> Int (GHC's floorDoubleInt, a cvttsd2si on x86-64): a NaN or a value out of range gives minBound."""
    if x != x or abs(x) >= 9.223372036854775807e18:
        return -(1 << 63)
    return int(math.floor(x))


def hs_pow(x, y):
    """(**) on Double = libm's pow (GHC.Float powerDouble): NaN for a negative base with a fractional exponent, +-inf on overflow
    and for 0 ** negative - where Python's ** returns a complex number or raises."""
    try:
        return math.pow(x, y)
    except OverflowError:
        return math.inf if (x > 0 or float(y).is_integer() and int(y) % 2 == 0) else -math.inf
    except ValueError:
        return math.inf if x == 0 else math.nan


def hs_atan2(y, x):
    """atan2 on Double under GHC: the RealFloat class DEFAULT (GHC.Float; the Haskell 2010 report's Prelude definition) - instance
    RealFloat Double does not define atan2, so libm's atan2 is never called: division, atan and `pi + ...` each round."""
    neg_zero = lambda v: v == 0.0 and math.copysign(1.0, v) < 0
    if x > 0:
        return math.atan(fdiv(y, x))
    if x == 0 and y > 0:
        return math.pi / 2
    if x < 0 and y > 0:
        return math.pi + math.atan(fdiv(y, x))
    if (x <= 0 and y < 0) or (x < 0 and neg_zero(y)) or (neg_zero(x) and neg_zero(y)):
        return -hs_atan2(-y, x)
    if y == 0 and (x < 0 or neg_zero(x)):
        return math.pi
    if x == 0 and y == 0:
        return y
    return x + y


def hs_min(x, y): return x if x <= y else y          # GHC.Classes defaults of Ord Double: NOT Python's min / max on a NaN
def hs_max(x, y): return y if x <= y else x
def clamp3(v, lo, hi): return tuple(hs_max(lo, hs_min(hi, x)) for x in v)   # Trace.hs:91-94: cl v = max lo (min hi v)

DEFAULT_MATERIAL = {"diffuse": (1.0, 1.0, 1.0)}                         # Rendering/Material.hs:25-35


def trace_ray(scene, ray0, max_depth, draw):
    """draw(bounce) -> (X, Y, Z, D): the four draws of Trace.hs:38-41 for the bounce-th hit of this path."""
    world, materials, lights = scene["world"], scene["materials"], scene["lights"]

    def background_sample(att, ray):                                   # :78-83
        return mul(att, scene["background"](ray))

    def loop(ray, depth, att, bounce):
        if depth == 0:
            return background_sample(att, ray0)                        # ray0, not the current ray (:22)
        rec = hit_bvh(world, ray, (0.001, 10000))
        if rec is None:
            return background_sample(att, ray)
        mat = materials.get(rec["mat"], DEFAULT_MATERIAL)
        emitted = mat.get("emission") or (0.0, 0.0, 0.0)
        surface = mat["diffuse"]
        n = rec["normal"]
        unit_dir = normalize(ray[1])
        lit = mul(surface, compute_lighting(rec, lights, world))
        rx, ry, rz, rd_ = draw(bounce)
        rand_vec = normalize((rx, ry, rz))
        if mat.get("transmission") == 1.0 and mat.get("ior") is not None:
            ior = mat["ior"]
            eta = fdiv(1.0, ior) if rec["front"] else ior
            cos_theta = hs_min(-dot(unit_dir, n), 1.0)
            sin_theta = math.sqrt(1.0 - cos_theta * cos_theta)
            if eta * sin_theta > 1.0 or rd_ < schlick(cos_theta, eta):
                direction = reflect(unit_dir, n)
            else:
                direction = refract(unit_dir, n, eta)
            nxt = (rec["point"], direction)
        elif mat.get("shininess") is not None and mat["shininess"] > 100:
            nxt = (rec["point"], normalize(add(reflect(unit_dir, n), scale(0.05, rand_vec))))
        else:
            nxt = (rec["point"], normalize(add(n, rand_vec)))
        new_att = att if mat.get("transmission") == 1.0 else mul(att, surface)
        bounce_color = loop(nxt, depth - 1, new_att, bounce + 1)
        return add(add(emitted, lit), clamp3(bounce_color, 0, 10.0))

    return loop(ray0, max_depth, (1.0, 1.0, 1.0), 0)


# ---------------------------------------------------------------- Rendering/Camera.hs:26-60, RenderWorker.hs:76-114
def default_camera(look_from, look_at, v_up, vfov, aspect, aperture, focus_dist):
    theta = vfov * math.pi / 180.0
    h = math.tan(theta / 2)
    vh = 2.0 * h * focus_dist
    vw = vh * aspect
    w = normalize(sub(look_from, look_at))
    u = normalize(cross(v_up, w))
    v = cross(w, u)
    horizontal, vertical = scale(vw, u), scale(vh, v)
    llc = sub(sub(sub(look_from, scale(0.5, horizontal)), scale(0.5, vertical)), scale(focus_dist, w))
    return {"origin": look_from, "llc": llc, "horizontal": horizontal, "vertical": vertical, "aperture": aperture, "u": u, "v": v}


def generate_ray(cam, i, j, width, height, u_off, v_off, lens):
    lens_offset = scale(cam["aperture"] / 2, add(scale(lens[0], cam["u"]), scale(lens[1], cam["v"])))
    origin = add(cam["origin"], lens_offset)
    rayu = (float(i) + u_off) / float(width)
    rayv = (float(j) + v_off) / float(height)
    direction = sub(add(add(cam["llc"], scale(rayu, cam["horizontal"])), scale(rayv, cam["vertical"])), origin)
    return (origin, normalize(direction))


def render(scene, cam, width, height, spp, antialiasing, max_bounces, exposure, gamma, seed=0):
    """Returns radiance, color, rgb8, prim_id as nested lists in FILE order (row 0 = top)."""
    key = (seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF)
    rad, col, rgb, ids = [], [], [], []
    for py in range(height):                                           # file row py is the reference's j = H-1-py (RenderWorker.hs:66)
        rr, cc, gg, ii = [], [], [], []
        for i in range(width):
            pixel = py * width + i
            samples, first_id = [], -1
            for s in range(spp):
                u_off = v_off = 0.5                                    # RenderWorker.hs:98-99
                lens = (0.0, 0.0)
                if antialiasing or cam["aperture"] != 0.0:
                    w = philox4x32_10((pixel, s, 0, 0), key)
                    if antialiasing:
                        u_off, v_off = u01(w[0]), u01(w[1])
                    if cam["aperture"] != 0.0:                         # randomInUnitDisk (Core/Vec3.hs:126-133): rejection
                        lens, sub_ = (um11(w[2]), um11(w[3])), 1
                        while not (lens[0] * lens[0] + lens[1] * lens[1] < 1.0) and sub_ < 64:
                            w = philox4x32_10((pixel, s, 0, sub_), key)
                            lens = (um11(w[0]), um11(w[1]))
                            if not (lens[0] * lens[0] + lens[1] * lens[1] < 1.0):
                                lens = (um11(w[2]), um11(w[3]))
                            sub_ += 1
                        if not (lens[0] * lens[0] + lens[1] * lens[1] < 1.0):
                            lens = (0.0, 0.0)
                ray = generate_ray(cam, i, height - 1 - py, width, height, u_off, v_off, lens)

                def draw(bounce, pixel=pixel, s=s):
                    w4 = philox4x32_10((pixel, s, 1 + bounce, 0), key)
                    return um11(w4[0]), um11(w4[1]), um11(w4[2]), u01(w4[3])

                if s == 0 and max_bounces > 0:
                    h0 = hit_bvh(scene["world"], ray, (0.001, 10000))
                    first_id = -1 if h0 is None else h0["id"]
                samples.append(trace_ray(scene, ray, max_bounces, draw))
            acc = (0.0, 0.0, 0.0)
            for c in samples:                                          # foldl' add, RenderWorker.hs:111-114
                acc = add(acc, c)
            avg = scale(1.0 / float(len(samples)), acc)
            toned = tuple(fdiv(x * exposure, x * exposure + 1) for x in avg)          # Trace.hs:96-101
            gammaed = tuple(hs_pow(x, gamma) for x in toned)                            # Trace.hs:103-105
            rr.append(avg); cc.append(gammaed); ii.append(first_id)
            # RenderWorker.hs:108-109 truncate (c * 255.999); the 8-bit output convention of include/rtcore.h (rt_output.rgb8): clamped
            # to 0..255, NaN -> 0 (the reference would `show` an out-of-range integer there)
            gg.append(tuple(0 if x != x else min(255, max(0, int(math.trunc(min(1e9, max(-1e9, x * 255.999)))))) for x in gammaed))
        rad.append(rr); col.append(cc); rgb.append(gg); ids.append(ii)
    return rad, col, rgb, ids


# ---------------------------------------------------------------- scene assembly from a SceneSpec (tests only)
def scene_from_spec(spec):
    kind, data, mat = spec.all_objects()                                # allObjects order, SceneParser.hs:67-73
    objs = []
    for idx in range(len(kind)):
        d = [float(x) for x in data[idx]]
        if kind[idx] == 0:
            objs.append(("tri", tuple(d[0:3]), tuple(d[3:6]), tuple(d[6:9]), int(mat[idx]), idx))
        elif kind[idx] == 1:
            objs.append(("sph", tuple(d[0:3]), d[3], int(mat[idx]), idx))
        else:
            objs.append(("pl", tuple(d[0:3]), tuple(d[3:6]), int(mat[idx]), idx))
    materials = {}
    for k, (name, m) in enumerate(spec.materials):                      # assignMaterialIds: ids from 1 (Parser/Material.hs:52-57)
        e = {"diffuse": tuple(float(x) for x in m.get("diffuseColor", [1, 1, 1]))}
        if "emissionColor" in m: e["emission"] = tuple(float(x) for x in m["emissionColor"])
        for src, dst in (("shininess", "shininess"), ("ior", "ior"), ("transmission", "transmission")):
            if src in m: e[dst] = float(m[src])
        materials[k + 1] = e
    lights = [{"kind": "point" if l["tag"] == "PointLight" else "dir",
               "vec": tuple(float(x) for x in (l["position"] if l["tag"] == "PointLight" else l["direction"])),
               "intensity": tuple(float(x) for x in l["intensity"])} for l in spec.lights]
    bg = spec.background
    if spec.sky is not None:
        fmt, img = spec.sky
        sh, sw = img.shape[0], img.shape[1]

        def background(ray):                                           # sampleSkySphere, Rendering/SkySphere.hs:32-47
            dx, dy, dz = normalize(ray[1])
            u = 0.5 + hs_atan2(dz, dx) / (2 * math.pi)
            v = 0.5 - math.asin(dy) / math.pi
            i = max(0, min(hs_floor_int(u * float(sw)), sw - 1))
            j = max(0, min(hs_floor_int(v * float(sh)), sh - 1))
            p = img[j, i]
            return tuple(float(x) / 255 for x in p) if fmt == "rgb8" else tuple(float(x) for x in p)
    elif bg["tag"] == "SolidColor":
        c = tuple(float(x) for x in bg["color"])
        background = lambda ray: c                                      # ImageGenerator.hs:131
    else:
        c1, c2 = tuple(float(x) for x in bg["color1"]), tuple(float(x) for x in bg["color2"])

        def background(ray):                                           # ImageGenerator.hs:132-135, Color.hs:16-17
            t = 0.5 * (normalize(ray[1])[1] + 1.0)
            return add(scale(1.0 - t, c2), scale(t, c1))
    world = construct_bvh(spec.leaf_threshold, spec.bvh_max_depth, objs)
    cam = spec.camera
    camera = default_camera(tuple(map(float, cam["lookFrom"])), tuple(map(float, cam["lookAt"])), tuple(map(float, cam["vUp"])),
                            float(cam["vfov"]), spec.width / spec.height, float(cam["aperture"]), float(cam["focusDist"]))
    return {"world": world, "materials": materials, "lights": lights, "background": background}, camera
