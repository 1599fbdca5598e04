"""The two restatements of the reference (oracle/oracle.cpp, oracle/pyref.py) on the random scenes of tools/fuzz_oracle.py at
postage-stamp size, CPU only: tools/fuzz_pyref.py, 60 + 30 seeds here (4,500 were run once without a mismatch)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run(*args):
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "fuzz_pyref.py"), *args], cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, (p.stdout + p.stderr)[-3000:]
    assert "mismatching seeds: 0" in p.stdout


def test_restatements_agree_on_random_scenes():
    run("0", "60")


def test_restatements_agree_on_extreme_scenes():
    run("0", "30", "--extreme")


def test_restatements_agree_on_degenerate_inputs():
    """zero-direction lights, lights on surfaces, zero-area triangles, a plane of normal 0, ior 0 / 1, NaN camera bases (fuzz_oracle --degenerate)"""
    run("0", "30", "--degenerate")
