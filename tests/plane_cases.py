"""Scenes whose planes sit OUTSIDE the constant slab box the reference gives every plane (Plane.hs:20).

Such a plane is visible only through the interval its leaf receives (Plane.hs:26): [enter, min(exit, closest so far)]
of the leaf's box, or - when the leaf is visited second and its sibling produced a hit - [enter, t firstHit]
(BVH.hs:137-138), which reaches past the box's exit.  Shared by the CPU tests (oracle vs the second restatement) and
the GPU tests (render core vs oracle).
"""
import numpy as np

from raytracer_amd import scenes


def two_object_scene(width=96, height=54, spp=1, max_bounces=1, antialiasing=False, leaf_threshold=1):
    """VERDICT r01 'What's weak' #1: sphere c=(0,-9,-6) r=10 + plane p=(0,-0.5,0) n=(0,1,0), leafThreshold 1, camera (0,2,5)->(0,0,0).
    The ray o=(0,2,5) d~(0,-0.5,-1) leaves the plane leaf's box at 4.4744, hits the sphere (first child) at 6.0517 and the
    reference returns the PLANE at t=5.5902."""
    return scenes.SceneSpec(
        name="plane_two_objects", width=width, height=height, spp=spp, antialiasing=antialiasing, gamma=1.0, exposure=1.0,
        background={"tag": "SolidColor", "color": [0.2, 0.3, 0.4]},
        camera={"lookFrom": [0.0, 2.0, 5.0], "lookAt": [0.0, 0.0, 0.0], "vUp": [0.0, 1.0, 0.0], "vfov": 60.0,
                "aperture": 0.0, "focusDist": 3.0},
        max_bounces=max_bounces, bvh_max_depth=32, leaf_threshold=leaf_threshold,
        materials=[("a", {"diffuseColor": [0.8, 0.2, 0.2]}), ("b", {"diffuseColor": [0.2, 0.8, 0.2]})],
        lights=[{"tag": "PointLight", "position": [3.0, 6.0, 4.0], "intensity": [30.0, 30.0, 30.0]}],
        config_objects=[("sphere", [0.0, -9.0, -6.0], 10.0, "a"), ("plane", [0.0, -0.5, 0.0], [0.0, 1.0, 0.0], "b")])


def random_plane_scene(seed, n_tris=400, n_planes=3, width=112, height=64, spp=2):
    """Triangle soup + spheres + 2-3 planes of random point and normal, random leafThreshold / bvhMaxDepth."""
    rng = np.random.default_rng(seed)
    ctr = rng.uniform([-4, -1.0, -4], [4, 4, 4], size=(n_tris, 3))
    size = np.where(rng.random(n_tris) < 0.02, 5.0, rng.uniform(0.1, 0.8, n_tris))
    tri = ctr[:, None, :] + rng.uniform(-1, 1, size=(n_tris, 3, 3)) * size[:, None, None]
    names = ["matte", "metal", "glass", "lamp", None]
    fm = [names[i] for i in rng.integers(0, len(names), n_tris)]
    mats = [("matte", {"diffuseColor": [0.7, 0.5, 0.3]}),
            ("metal", {"diffuseColor": [0.9, 0.9, 0.9], "shininess": 300.0}),
            ("glass", {"diffuseColor": [1.0, 1.0, 1.0], "ior": 1.5, "transmission": 1.0}),
            ("lamp", {"diffuseColor": [0.8, 0.8, 0.8], "emissionColor": [0.4, 0.3, 0.2]})]
    cfg = [("sphere", list(rng.uniform(-3, 3, 3) + [0, 1.5, 0]), float(rng.uniform(0.3, 1.5)), names[int(rng.integers(0, 4))])
           for _ in range(5)]
    for k in range(n_planes):
        n = rng.normal(size=3)
        if k == 0:
            n = np.array([0.05 * rng.normal(), 1.0, 0.05 * rng.normal()])     # one nearly horizontal plane below the soup
        n = n / np.linalg.norm(n)
        p = rng.uniform([-2, -1.5, -2], [2, 1.0, 2])
        cfg.append(("plane", [float(x) for x in p], [float(x) for x in n], names[int(rng.integers(0, 4))]))
    lights = [{"tag": "PointLight", "position": [5.0, 9.0, -4.0], "intensity": [40.0, 38.0, 35.0]},
              {"tag": "DirectionalLight", "direction": [-0.4, -1.0, 0.3], "intensity": [0.4, 0.4, 0.45]}]
    return scenes.SceneSpec(
        name=f"planes{seed}", width=width, height=height, spp=spp, antialiasing=True, gamma=1.0, exposure=1.0,
        background={"tag": "Gradient", "color1": [1.0, 1.0, 1.0], "color2": [0.5, 0.7, 1.0]},
        camera={"lookFrom": [-8.0, 4.0, -7.0], "lookAt": [0.0, 1.0, 0.0], "vUp": [0.0, 1.0, 0.0], "vfov": 55.0,
                "aperture": 0.0, "focusDist": 3.0},
        max_bounces=int(rng.integers(2, 6)), bvh_max_depth=int(rng.choice([3, 6, 32])), leaf_threshold=int(rng.integers(1, 5)),
        materials=mats, lights=lights, config_objects=cfg,
        obj_meshes=[scenes.ObjMesh(tri.reshape(-1, 3), np.arange(3 * n_tris).reshape(n_tris, 3), fm, (0.0, 0.0, 0.0))])
