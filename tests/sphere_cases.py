"""Scenes in which a sphere is hit where it TOUCHES its own box (Sphere.hs:20-27, BVH.hs:124-139).

The reference boxes a sphere exactly (c -+ r, no pad, unlike Triangle.hs:24-28) and tests `contains interval t1` with the
interval its leaf receives.  A ray along an axis through the centre has t1 == the box's entry distance in exact
arithmetic; in FP64 t1 comes out one ulp BELOW it about one time in three, and the reference then sees through the sphere
whenever the leaf is reached through an internal node (a single-leaf root hands down the raw interval and sees it).
Shared by the CPU tests (oracle vs the second restatement) and the GPU tests (render core vs oracle)."""
from raytracer_amd import scenes

# the judge's sphere: centre pixel of a 33 x 17 frame, camera on the +z axis through c
C = [3.3576510391986965, -0.6723293209494665, 2.6228008245794197]
R = 0.10610755471822102


def two_sphere_scene(leaf_threshold=1, dist=10.123075247571, width=33, height=17, lights=False, max_bounces=1):
    """leafThreshold 1: two leaves under an internal root - the centre ray's t1 is one ulp below the slab entry and the
    reference returns NO hit there (id -1); leafThreshold 4: one leaf = the root, raw interval, id 0."""
    return scenes.SceneSpec(
        name=f"sphere_touch_lt{leaf_threshold}", width=width, height=height, spp=1, antialiasing=False, gamma=1.0, exposure=1.0,
        background={"tag": "SolidColor", "color": [0.2, 0.3, 0.4]},
        camera={"lookFrom": [C[0], C[1], C[2] + dist], "lookAt": list(C), "vUp": [0.0, 1.0, 0.0], "vfov": 40.0,
                "aperture": 0.0, "focusDist": 3.0},
        max_bounces=max_bounces, bvh_max_depth=32, leaf_threshold=leaf_threshold,
        materials=[("a", {"diffuseColor": [0.8, 0.2, 0.2]})],
        lights=[{"tag": "PointLight", "position": [C[0], C[1] + 7.0, C[2]], "intensity": [30.0, 30.0, 30.0]}] if lights else [],
        config_objects=[("sphere", list(C), R, "a"), ("sphere", [C[0] + 50.0, C[1], C[2]], 1.0, "a")])
