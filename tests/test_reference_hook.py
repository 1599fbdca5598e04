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


FAKE_CABAL = r"""#!/bin/bash
# stands in for `cabal run -v0 raytracer <config.json> +RTS ...` (the real one needs GHC): renders the config with the
# repository's oracle into out/<timestamp>.ppm like Main.hs:48-58 and prints the line the script greps for
cfg=""
for a in "$@"; do case "$a" in *.json) cfg="$a";; esac; done
mkdir -p out
python3 "$RT_REPO/tools/oracle_ppm.py" "$cfg" out/20250101-000000.ppm > /dev/null
if [ "${FAKE_PERTURB:-0}" != "0" ]; then
python3 - out/20250101-000000.ppm "$FAKE_PERTURB" <<'PY'
import sys
tok = open(sys.argv[1]).read().split()
d = int(sys.argv[2])
vals = [str(min(255, max(0, int(v) + (d if i % 2 else -d)))) for i, v in enumerate(tok[4:])]
open(sys.argv[1], "w").write("P3\n%s %s\n%s\n" % (tok[1], tok[2], tok[3]) + "\n".join(vals) + "\n")
PY
fi
echo "Time taken to create and write file: 1.23s"
"""


def run_hook(tmp_path, perturb):
    ref = tmp_path / "checkout"; ref.mkdir(exist_ok=True)
    bindir = tmp_path / "bin"; bindir.mkdir(exist_ok=True)
    cabal = bindir / "cabal"
    cabal.write_text(FAKE_CABAL)
    cabal.chmod(0o755)
    env = dict(os.environ, PATH=f"{bindir}:{os.environ['PATH']}", RT_REPO=ROOT, FAKE_PERTURB=str(perturb),
               RT_REF_CASES="t_det|c1_testing(width=64, height=36, spp=1)|det;t_mean|c2_spheres(width=32, height=18, spp=8)|mean")
    return subprocess.run(["bash", os.path.join(ROOT, "bench", "run_reference.sh"), str(ref), str(tmp_path / "work")],
                          env=env, capture_output=True, text=True, timeout=300)


def test_hook_passes_on_equal_frames_and_fails_on_a_two_level_shift(tmp_path):
    """the script has never met a real `cabal`; a fake one that produces the oracle's own frame must make
    it pass (exit 0, the reference's time line reported), and the same frame shifted by +-2 levels must fail the deterministic
    case (<= 1 level on >= 99.99 % of the values) while the mean-image case (PSNR >= 30 dB) still passes; +-40 levels fails both."""
    ok = run_hook(tmp_path, 0)
    assert ok.returncode == 0, ok.stdout[-1500:] + ok.stderr[-1500:]
    assert ok.stdout.count("PASS") == 2 and "Time taken to create and write file" in ok.stdout
    two = run_hook(tmp_path, 2)
    assert two.returncode == 1, two.stdout[-1500:]
    lines = [l for l in two.stdout.splitlines() if l in ("PASS", "FAIL")]
    assert lines == ["FAIL", "PASS"], two.stdout[-1500:]
    forty = run_hook(tmp_path, 40)
    assert forty.returncode == 1 and [l for l in forty.stdout.splitlines() if l in ("PASS", "FAIL")] == ["FAIL", "FAIL"]
