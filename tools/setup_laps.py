#!/usr/bin/env python3
"""Where the set-up time of one scene goes (GPU box): python tools/setup_laps.py [c4|c5]   (RT_DEBUG=1 prints rt_scene_create's laps)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["RT_DEBUG"] = "1"
t00 = time.time()
import torch
from raytracer_amd import api, scenes, _ffi as F
t_import = time.time() - t00
which = sys.argv[1] if len(sys.argv) > 1 else "c4"
spec = {"c4": scenes.c4_shipped, "c5": scenes.c5_multi}[which]()
for rep in range(2):
    t0 = time.time(); kind, data, mat = spec.all_objects(); t1 = time.time()
    bvh = api.HostBvh(kind, data, mat, spec.leaf_threshold, spec.bvh_max_depth, 8); t2 = time.time()
    if rep == 0:
        F.check(F.core().rt_init(0)); t3 = time.time()
    else:
        t3 = t2
    g = api.GpuScene(spec, bvh=bvh, n_threads=8, device=0); t4 = time.time()
    print(f"rep {rep}: import {t_import:.3f} scene assembly {t1 - t0:.3f} reference tree {t2 - t1:.3f} rt_init {t3 - t2:.3f} GpuScene (flatten + rt_scene_create) {t4 - t3:.3f}", flush=True)
    g.close()
