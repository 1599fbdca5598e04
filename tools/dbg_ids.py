"""Debug aid: primitive-id mismatches and radiance error of the loaded librtcore (RTCORE_LIB) against the oracle."""
import sys, os, numpy as np
sys.path.insert(0, os.getcwd())
from raytracer_amd import api, scenes
from oracle import oracle as O
cases = [("c1", scenes.c1_testing(width=33, height=17, spp=1)), ("c1", scenes.c1_testing(width=96, height=54, spp=1)),
         ("c3", scenes.c3_blob988(width=96, height=56, spp=4)), ("c2", scenes.c2_spheres(width=80, height=48, spp=4))]
for name, spec in cases:
    g = api.GpuScene(spec); res, _ = g.render(); g.close()
    o = O.OracleScene(spec); ref = o.render(); o.close()
    bad = np.argwhere(res["prim_id"] != ref["prim_id"])
    err = np.abs(res["radiance"] - ref["radiance"])
    print(name, spec.width, spec.height, "id mismatches", len(bad), "max |drad|", err.max(), "n>1e-9", int((err > 1e-9).sum()), "n>1e-13", int((err > 1e-13).sum()), "of", err.size)
    for y, x in bad[:4]:
        print("  px", x, y, "gpu", res["prim_id"][y, x], "ref", ref["prim_id"][y, x])
