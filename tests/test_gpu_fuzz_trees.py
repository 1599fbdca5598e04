"""GPU fuzz: the render core's own tree (box_precision 0) against the literal walk of the reference's tree
(box_precision 1) on random scenes - triangle soups with a few room-sized triangles, overlapping spheres, a plane,
glass / metal / emissive materials, two or three lights, random leafThreshold / bvhMaxDepth.  Both modes run the same
FP64 primitive tests, shading and RNG, so the frames must agree to the bit wherever the same primitive wins: ids equal
on every pixel, radiance within 1e-12.  One seed is also checked against the CPU oracle (the reference restated)."""
import numpy as np
import pytest

from raytracer_amd import api, scenes

pytestmark = pytest.mark.gpu


def random_scene(seed, n_tris=3000, width=160, height=96, spp=3):
    rng = np.random.default_rng(seed)
    ctr = rng.uniform([-4, 0.1, -4], [4, 4, 4], size=(n_tris, 3))
    size = np.where(rng.random(n_tris) < 0.01, 6.0, rng.uniform(0.05, 0.5, n_tris))        # 1 % room-sized triangles
    tri = ctr[:, None, :] + rng.uniform(-1, 1, size=(n_tris, 3, 3)) * size[:, None, None]
    verts = tri.reshape(-1, 3)
    faces = np.arange(3 * n_tris).reshape(n_tris, 3)
    names = ["matte", "metal", "glass", "lamp", None]
    fm = [names[i] for i in rng.integers(0, len(names), n_tris)]
    mats = [("matte", {"diffuseColor": [0.7, 0.5, 0.3]}),
            ("metal", {"diffuseColor": [0.9, 0.9, 0.9], "shininess": 300.0}),
            ("glass", {"diffuseColor": [1.0, 1.0, 1.0], "ior": 1.5, "transmission": 1.0}),
            ("lamp", {"diffuseColor": [0.8, 0.8, 0.8], "emissionColor": [0.4, 0.3, 0.2]})]
    cfg = [("sphere", list(rng.uniform(-3, 3, 3) + [0, 2, 0]), float(rng.uniform(0.3, 1.2)), names[int(rng.integers(0, 4))]) for _ in range(6)]
    cfg.append(("plane", [0.0, 0.0, 0.0], [0.0, 1.0, 0.0], "matte"))
    cfg.append(("triangle", [-9.0, 0.05, -9.0], [-9.0, 0.05, 9.0], [9.0, 0.05, 0.0], "lamp"))  # listed twice by the reference
    lights = [{"tag": "PointLight", "position": [5.0, 9.0, -4.0], "intensity": [40.0, 38.0, 35.0]},
              {"tag": "DirectionalLight", "direction": [-0.4, -1.0, 0.3], "intensity": [0.4, 0.4, 0.45]}]
    if seed % 2:
        lights.append({"tag": "PointLight", "position": [-6.0, 3.0, 6.0], "intensity": [15.0, 15.0, 20.0]})
    return scenes.SceneSpec(
        name=f"fuzz{seed}", width=width, height=height, spp=spp, antialiasing=True, gamma=1.0, exposure=1.0,
        background={"tag": "Gradient", "color1": [1.0, 1.0, 1.0], "color2": [0.5, 0.7, 1.0]},
        camera={"lookFrom": [-8.0, 4.0, -7.0], "lookAt": [0.0, 1.5, 0.0], "vUp": [0.0, 1.0, 0.0], "vfov": 55.0,
                "aperture": 0.0, "focusDist": 3.0},
        max_bounces=int(rng.integers(2, 7)), bvh_max_depth=int(rng.choice([6, 12, 32])), leaf_threshold=int(rng.integers(1, 7)),
        materials=mats, lights=lights, config_objects=cfg, obj_meshes=[scenes.ObjMesh(verts, faces, fm, (0.0, 0.0, 0.0))])


@pytest.mark.parametrize("seed", [11, 12, 13, 14, 15, 16])
def test_own_tree_equals_reference_tree(need_gpu, seed):
    spec = random_scene(seed)
    g = api.GpuScene(spec)
    a, sa = g.render(collect_stats=1, box_precision=0)
    b, sb = g.render(collect_stats=1, box_precision=1)
    c, _ = g.render(box_precision=2)
    g.close()
    assert (b["prim_id"] >= 0).mean() > 0.3
    for name, r in (("own tree", a), ("reference tree, FP32 boxes", c)):
        assert (r["prim_id"] == b["prim_id"]).all(), f"{name}: {(r['prim_id'] != b['prim_id']).sum()} ids differ"
        assert np.abs(r["radiance"] - b["radiance"]).max() <= 1e-12, name
        assert (r["rgb8"] == b["rgb8"]).all(), name
    assert sa.rays_closest == sb.rays_closest and sa.rays_shadow + sa.rays_shadow_culled == sb.rays_shadow


def test_fuzz_scene_against_the_oracle(need_gpu):
    from oracle import oracle as O
    spec = random_scene(21, n_tris=1200, width=128, height=72, spp=2)
    g = api.GpuScene(spec)
    a, _ = g.render()
    g.close()
    o = O.OracleScene(spec)
    ref = o.render()
    o.close()
    assert (a["prim_id"] == ref["prim_id"]).all()
    assert np.abs(a["radiance"] - ref["radiance"]).max() <= 1e-12
