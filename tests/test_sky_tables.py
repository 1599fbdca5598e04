"""The sky lookup's host tables (raytracer_amd/csrc/rt_sky_tables.h) against the reference's expressions (SkySphere.hs:35-40) as
oracle/pyref.py restates them, GHC's atan2 included: every row threshold is the last double of its row, and for directions within
1e-19 ... 1e-14 rad of a column boundary the device's quotient test gives the reference's column - exactly, no tolerance."""
import math
import os
import subprocess

import pytest

from oracle.pyref import hs_atan2

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def driver(tmp_path_factory):
    exe = str(tmp_path_factory.mktemp("skytab") / "driver")
    subprocess.run(["g++", "-O2", "-ffp-contract=off", "-I", os.path.join(ROOT, "raytracer_amd", "csrc"),
                    os.path.join(ROOT, "tests", "sky_tables_driver.cpp"), "-o", exe], check=True)
    return exe


@pytest.mark.parametrize("w,h", [(64, 32), (100, 50), (17, 9), (2, 1), (4, 2), (1, 1), (333, 200), (2048, 1024)])
def test_tables_reproduce_the_reference_texel(driver, w, h):
    out = subprocess.run([driver, str(w), str(h), "0" if w < 2 else "6000"], check=True, capture_output=True, text=True).stdout
    row = lambda y: math.floor((0.5 - math.asin(y) / math.pi) * h)
    rows = cols = 0
    for line in out.splitlines():
        f = line.split()
        if f[0] == "R":
            k, t = int(f[1]), float.fromhex(f[2])
            if k <= 0:
                assert t == math.inf
            elif k > h:
                assert t == -math.inf
            else:
                assert row(t) >= k and (t == 1.0 or row(math.nextafter(t, math.inf)) < k), (k, t)
                rows += 1
        else:
            k, x, z, side = int(f[1]), float.fromhex(f[2]), float.fromhex(f[3]), int(f[4])
            want = 1 if math.floor((0.5 + hs_atan2(z, x) / (2 * math.pi)) * w) >= k else 0
            assert side == want, (k, x, z)
            cols += 1
    assert rows == h and cols == (0 if w < 2 else 6000)


def test_ghc_atan2_is_not_libm_atan2():
    """The reason for the tables' column half: about a quarter of all directions get a different last bit."""
    import random
    rnd = random.Random(3)
    pts = [(rnd.uniform(-1, 1), rnd.uniform(-1, 1)) for _ in range(20000)]
    differ = sum(hs_atan2(z, x) != math.atan2(z, x) for x, z in pts)
    assert 0.1 < differ / len(pts) < 0.5
    assert all(abs(hs_atan2(z, x) - math.atan2(z, x)) <= 2 * math.ulp(math.pi) for x, z in pts)
    for y, x in [(0.0, -1.0), (-0.0, -1.0), (0.0, -0.0), (-0.0, -0.0), (0.0, 0.0), (-0.0, 0.0), (1.0, 0.0), (-1.0, 0.0), (1.0, -0.0), (-1.0, -0.0)]:
        a, b = hs_atan2(y, x), math.atan2(y, x)                       # the zero guards agree with libm, signs of zero included
        assert a == b and math.copysign(1.0, a) == math.copysign(1.0, b), (y, x)
