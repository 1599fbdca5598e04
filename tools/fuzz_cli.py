#!/usr/bin/env python3
"""The C++ host path on random scenes: the scenes of tools/fuzz_oracle.py written as reference-schema JSON + OBJ files, rendered by
`rtcli <config.json>` (JSON / OBJ parsing, scene assembly, the reference's BVH build, camera, P3 text: all in C++, rendered on the
GPU through the C ABI), and the written PPM compared VALUE FOR VALUE with the oracle's 8-bit frame of the same SceneSpec.
(No sky texture: the writer has no image formats.  Faces and objects without a material get "matte": an OBJ file cannot return to
"no material" after a usemtl line.)
    python tools/fuzz_cli.py <first seed> <last seed + 1> [--extreme]"""
import glob
import os
import shutil
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import fuzz_oracle as FZ                            # noqa: E402
from oracle import oracle as O                      # noqa: E402
from raytracer_amd import scenes                    # noqa: E402


def read_p3(path):
    tok = open(path).read().split()
    assert tok[0] == "P3" and int(tok[3]) == 255
    return np.array(tok[4:], dtype=np.int64).reshape(int(tok[2]), int(tok[1]), 3)


def main():
    extreme = "--extreme" in sys.argv
    if extreme:
        sys.argv.remove("--extreme")
    lo, hi = int(sys.argv[1]), int(sys.argv[2])
    exe = os.path.join(ROOT, "raytracer_amd", "bin", "rtcli")
    bad = 0
    for seed in range(lo, hi):
        spec, _ = FZ.scene(seed, False, extreme)
        spec.sky = None
        for m in spec.obj_meshes:
            m.face_material = [f if f is not None else "matte" for f in m.face_material]
        spec.config_objects = [tuple(c[:-1]) + (c[-1] if c[-1] is not None else "matte",) for c in spec.config_objects]
        spec.camera = {k: ([float(x) for x in v] if isinstance(v, (list, tuple, np.ndarray)) else float(v)) for k, v in spec.camera.items()}
        for L in spec.lights:
            for k in ("position", "direction", "intensity"):
                if k in L:
                    L[k] = [float(x) for x in L[k]]
        wd = tempfile.mkdtemp(prefix="fzcli")
        try:
            cfg = scenes.write_reference_files(spec, os.path.join(wd, "scene"))
            p = subprocess.run([exe, cfg], cwd=wd, env=dict(os.environ, RT_SEED=str(spec.seed)), capture_output=True, text=True, timeout=120)
            if p.returncode != 0:
                print(seed, "rtcli failed:", (p.stdout + p.stderr)[-300:], flush=True)
                bad += 1
                continue
            img = read_p3(glob.glob(os.path.join(wd, "out", "*.ppm"))[0])
        finally:
            shutil.rmtree(wd, ignore_errors=True)
        o = O.OracleScene(spec)
        ref = o.render(n_threads=min(16, os.cpu_count() or 8))
        o.close()
        d = int((img != ref["rgb8"].astype(np.int64)).sum())
        bad += d != 0
        print(seed, f"{spec.width}x{spec.height}x{spec.spp}", "OK" if d == 0 else f"MISMATCH ({d} values)", flush=True)
    print("mismatching seeds:", bad)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
