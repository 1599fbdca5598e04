"""bench/run_reference.sh is the hook that would pin the oracle against the real Haskell binary on a box with GHC
(SURVEY 8c "Consequence").  GHC is absent here, so the CPU suite checks its moving parts: a reference-schema config
written by scenes.write_reference_files goes through the C++ config loader into the oracle and comes out as the P3 file
the differ reads, and the differ's verdicts are what the script relies on."""
import os
import subprocess
import sys

import numpy as np

from oracle import oracle as O
from raytracer_amd import scenes

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import ppm_diff  # noqa: E402


def test_config_to_oracle_to_ppm_round_trip(tmp_path):
    spec = scenes.c3_blob988(width=40, height=24, spp=2, max_bounces=3)
    spec.name = "hook"
    cfg = scenes.write_reference_files(spec, str(tmp_path))
    out = str(tmp_path / "oracle.ppm")
    subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "oracle_ppm.py"), cfg, out])
    img, mx = ppm_diff.read_p3(out)
    o = O.OracleScene(spec)
    ref = o.render(n_threads=2)
    o.close()
    assert mx == 255 and np.array_equal(img, ref["rgb8"].astype(np.int64))     # JSON + OBJ ingest reproduces the in-memory scene


def test_differ_verdicts(tmp_path):
    a = np.full((4, 5, 3), 100, np.int64)
    r = ppm_diff.compare(a, a)
    assert r["max_abs"] == 0 and r["psnr_db"] == float("inf")
    b = a.copy(); b[0, 0, 0] = 140
    r = ppm_diff.compare(a, b)
    assert r["max_abs"] == 40 and abs(r["frac_equal"] - 59 / 60) < 1e-12 and 25 < r["psnr_db"] < 40


def test_script_names_the_documented_invocation():
    s = open(os.path.join(ROOT, "bench", "run_reference.sh")).read()
    assert "cabal run -v0 raytracer" in s and "+RTS -N -qg -A32M" in s and "Time taken" in s
