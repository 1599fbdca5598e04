"""The C-ABI libraries load (no GPU needed) and export every symbol include/*.h declares."""
import ctypes
import os
import re

import pytest

from raytracer_amd import _ffi as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions(header):
    text = open(os.path.join(ROOT, "include", header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return set(re.findall(r"\b(rt_[a-z_0-9]+|rth_[a-z_0-9]+)\s*\(", text))


def test_rtcore_exports_every_declared_symbol():
    names = declared_functions("rtcore.h")
    assert names == set(F.RTCORE_SYMBOLS), names ^ set(F.RTCORE_SYMBOLS)
    lib = ctypes.CDLL(os.path.join(F.LIB_DIR, "librtcore.so"))
    for n in names:
        getattr(lib, n)
    assert F.core().rt_abi_version() == 4
    assert len(F.core().rt_build_id()) == 16 and F.core().rt_tile_skew() == 1


def test_rthost_exports_every_declared_symbol():
    names = declared_functions("rthost.h")
    assert names == set(F.RTHOST_SYMBOLS), names ^ set(F.RTHOST_SYMBOLS)
    lib = ctypes.CDLL(os.path.join(F.LIB_DIR, "librthost.so"))
    for n in names:
        getattr(lib, n)


def test_core_fails_loudly_without_a_device():
    """No CPU fallback: without a HIP device rt_init / rt_scene_create return RT_ERR_NO_DEVICE."""
    lib = F.core()
    rc = lib.rt_init(-1)
    if rc == 0:
        pytest.skip("a GPU is visible here")
    assert rc == F.RT_ERR_NO_DEVICE
    assert b"no CPU fallback" in lib.rt_last_error()
    from raytracer_amd import api, scenes
    with pytest.raises(F.RtError) as e:
        api.GpuScene(scenes.c2_spheres(width=16, height=8, spp=1))
    assert e.value.code == F.RT_ERR_NO_DEVICE


def test_tile_helpers_need_no_gpu():
    from raytracer_amd import api, dist
    for (w, h, world) in [(1920, 1080, 8), (101, 67, 3), (8, 8, 2), (640, 360, 1)]:
        seen = []
        for r in range(world):
            t = api.tile_list(w, h, r, world)
            assert (t == dist.tile_list_host(w, h, r, world)).all()
            seen += t.tolist()
        assert sorted(seen) == list(range(((w + 7) // 8) * ((h + 7) // 8)))      # a partition of all tiles
    counts = [len(api.tile_list(1920, 1080, r, 8)) for r in range(8)]
    assert max(counts) - min(counts) <= 30                                       # interleaving balances the shards
    with pytest.raises(F.RtError):
        api.tile_list(64, 64, 2, 2)
