import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np
from test_gpu_fuzz_trees import random_scene
from raytracer_amd import api
bad = 0
import sys as _s
lo_, hi_ = (int(_s.argv[1]), int(_s.argv[2])) if len(_s.argv) > 2 else (100, 140)
for seed in range(lo_, hi_):
    spec = random_scene(seed, n_tris=[500, 5000, 40000][seed % 3], width=224, height=128, spp=2)
    g = api.GpuScene(spec)
    a, _ = g.render(box_precision=0)
    b, _ = g.render(box_precision=1)
    g.close()
    ids = (a["prim_id"] != b["prim_id"]).sum(); d = np.abs(a["radiance"] - b["radiance"]).max()
    ok = ids == 0 and d <= 1e-12
    bad += not ok
    print(seed, spec.leaf_threshold, spec.bvh_max_depth, spec.max_bounces, "ids differ", ids, "max|d|", d, "OK" if ok else "MISMATCH", flush=True)
print("mismatching scenes:", bad)
