"""GPU, end to end through the drop-in's own executable: `rtcli <config.json>` (the mirror of `cabal run raytracer
<config.json>`, app/Main.hs) reads reference-schema JSON + OBJ files, builds the BVH on the host, renders on the GPU
through the C ABI and writes out/<timestamp>.ppm.  The file must equal the oracle's image of the same scene."""
import glob
import os
import subprocess

import numpy as np
import pytest

from oracle import oracle as O
from raytracer_amd import scenes

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def read_p3(path):
    tok = open(path).read().split()
    assert tok[0] == "P3"
    w, h, mx = int(tok[1]), int(tok[2]), int(tok[3])
    assert mx == 255
    return np.array(tok[4:], dtype=np.int64).reshape(h, w, 3)


@pytest.mark.parametrize("mk,kw", [(scenes.c1_testing, dict(width=160, height=90, spp=1)),
                                   (scenes.c3_blob988, dict(width=128, height=72, spp=8))])
def test_rtcli_ppm_equals_oracle(need_gpu, tmp_path, mk, kw):
    exe = os.path.join(ROOT, "raytracer_amd", "bin", "rtcli")
    if not os.path.exists(exe):
        pytest.fail("raytracer_amd/bin/rtcli is missing: run `make cli`")
    spec = mk(**kw)
    cfg = scenes.write_reference_files(spec, str(tmp_path / "scene"))
    env = dict(os.environ, RT_SEED=str(spec.seed))
    p = subprocess.run([exe, cfg], cwd=str(tmp_path), env=env, capture_output=True, text=True, timeout=100)
    assert p.returncode == 0, p.stdout + p.stderr
    assert "Time taken to create and write file" in p.stdout            # Main.hs:58
    files = glob.glob(str(tmp_path / "out" / "*.ppm"))
    assert len(files) == 1
    img = read_p3(files[0])
    o = O.OracleScene(spec)
    ref = o.render()
    o.close()
    d = np.abs(img - ref["rgb8"].astype(np.int64))
    assert d.max() == 0, f"max {d.max()}, equal {(d == 0).mean():.6f}"


def test_rtcli_device_group(need_gpu, tmp_path):
    """RT_DEVICES selects the device group of rtcli's single rt_render call, as the Haskell binding does with
    rt_init_devices: three virtual devices on this one GPU must write the same file as one device."""
    exe = os.path.join(ROOT, "raytracer_amd", "bin", "rtcli")
    spec = scenes.c3_blob988(width=128, height=72, spp=4)
    cfg = scenes.write_reference_files(spec, str(tmp_path / "scene"))
    frames = []
    for k, devs in enumerate((None, "0,0,0")):
        env = dict(os.environ, RT_SEED="5")
        if devs:
            env["RT_DEVICES"] = devs
        wd = tmp_path / f"run{k}"
        wd.mkdir()
        p = subprocess.run([exe, cfg], cwd=str(wd), env=env, capture_output=True, text=True, timeout=100)
        assert p.returncode == 0, p.stdout + p.stderr
        assert ("device group of 3" in p.stdout) == bool(devs)
        frames.append(read_p3(glob.glob(str(wd / "out" / "*.ppm"))[0]))
    assert np.array_equal(frames[0], frames[1])
