#!/usr/bin/env python3
"""Where the scene set-up time goes (C5 by default): scene assembly (numpy), reference BVH build (librthost),
rt_scene_create (validation, re-layout, the core's own tree, upload)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raytracer_amd import api, scenes
which = sys.argv[1] if len(sys.argv) > 1 else "c5"
spec = {"c5": scenes.c5_multi, "c4": scenes.c4_blob63k}[which]()
t = time.time(); k, d, m = spec.all_objects(); t_obj = time.time() - t
for nt in (8, 16):
    t = time.time(); bvh = api.HostBvh(k, d, m, spec.leaf_threshold, spec.bvh_max_depth, nt); t_bvh = time.time() - t
    t = time.time(); g = api.GpuScene(spec, bvh=bvh); t_create = time.time() - t
    print(f"{which}: all_objects {t_obj:.3f} s, reference tree ({nt} threads) {t_bvh:.3f} s, rt_scene_create {t_create:.3f} s", flush=True)
    for env in ("1", "4"):
        os.environ["RT_BUILD_THREADS"] = env
        t = time.time(); g2 = api.GpuScene(spec, bvh=bvh); print(f"   rt_scene_create with RT_BUILD_THREADS={env}: {time.time() - t:.3f} s", flush=True); g2.close()
    del os.environ["RT_BUILD_THREADS"]
    g.close(); bvh.close()
