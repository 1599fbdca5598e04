#!/usr/bin/env python3
"""The checker checked (CPU only): the two independent restatements of the reference - oracle/oracle.cpp and oracle/pyref.py - on the
random scenes of tools/fuzz_oracle.py at postage-stamp size (pyref is pure Python): spheres of radius <= 0 (inside-out boxes),
planes anywhere, duplicated triangles, depth-limited and one-object leaves, all materials, both light kinds, sky textures, thin lens.
Ids equal, radiance bit-identical, 8-bit frame equal - the bar of tests/test_oracle_pyref.py.
    python tools/fuzz_pyref.py <first seed> <last seed + 1> [--extreme | --degenerate]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import fuzz_oracle as FZ                            # noqa: E402
from oracle import oracle as O                      # noqa: E402
from oracle import pyref                            # noqa: E402


def main():
    extreme, degen = "--extreme" in sys.argv, "--degenerate" in sys.argv
    for f in ("--extreme", "--degenerate"):
        if f in sys.argv:
            sys.argv.remove(f)
    lo, hi = int(sys.argv[1]), int(sys.argv[2])
    bad = 0
    for seed in range(lo, hi):
        spec, _ = FZ.scene_degenerate(seed) if degen else FZ.scene(seed, False, extreme)
        m = spec.obj_meshes[0]
        if len(m.faces) > 300:                      # keep pure Python in seconds: the first 300 triangles
            m.faces = m.faces[:300]
            m.face_material = m.face_material[:300]
        spec.width, spec.height, spec.spp = 8 + seed % 9, 5 + seed % 6, 1 + seed % 2
        scene, cam = pyref.scene_from_spec(spec)
        rad, col, rgb, ids = pyref.render(scene, cam, spec.width, spec.height, spec.spp, spec.antialiasing, spec.max_bounces,
                                          spec.exposure, spec.gamma, seed=spec.seed)
        o = O.OracleScene(spec)
        ref = o.render(n_threads=2)
        o.close()
        rad, col, rgb, ids = np.array(rad), np.array(col), np.array(rgb, dtype=np.uint8), np.array(ids, dtype=np.int32)
        ok = (ids == ref["prim_id"]).all() and np.array_equal(rad, ref["radiance"], equal_nan=True) and (rgb == ref["rgb8"]).all()
        with np.errstate(invalid="ignore"):
            ok = ok and not (np.abs(col - ref["color"]) > 1e-15).any()
        bad += not ok
        kinds = [c[0] for c in spec.config_objects]
        neg = sum(1 for c in spec.config_objects if c[0] == "sphere" and c[2] <= 0)
        print(seed, f"{spec.width}x{spec.height}x{spec.spp} mb{spec.max_bounces} lt{spec.leaf_threshold} d{spec.bvh_max_depth} sph{kinds.count('sphere')}(r<=0: {neg}) pl{kinds.count('plane')}",
              "OK" if ok else f"MISMATCH ids {(ids != ref['prim_id']).sum()} rad {np.nanmax(np.abs(rad - ref['radiance'])):.2e}", flush=True)
    print("mismatching seeds:", bad)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
