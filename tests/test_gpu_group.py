"""GPU: the single-process device group behind the C ABI (rt_init_devices + ONE rt_render call, what a Haskell host
binds).  The box has one GPU, so the group is made of VIRTUAL devices - device 0 listed n times: n replicas of the
scene, n host threads, n streams, peer copies to replica 0, un-tiling - and every output must equal the single-device
rt_render bit for bit, for every n."""
import numpy as np
import pytest

from raytracer_amd import _ffi as F
from raytracer_amd import api, scenes

pytestmark = pytest.mark.gpu

WANTS = [("color", "radiance", "rgb8", "prim_id"), ("rgb8",), ("color",), ("radiance", "prim_id")]


@pytest.mark.parametrize("n", [2, 3, 8])
def test_group_render_is_bit_identical(need_gpu, n):
    spec = scenes.c3_blob988(width=203, height=117, spp=3)
    g1 = api.GpuScene(spec)
    ref, st1 = g1.render(collect_stats=1)
    g1.close()
    gN = api.GpuScene(spec, devices=[0] * n)
    assert F.core().rt_device_count() == n
    for want in WANTS:
        res, st = gN.render(want=want, collect_stats=1)
        for k in want:
            assert np.array_equal(res[k], ref[k]), (n, want, k)
        assert st.rays_closest == st1.rays_closest and st.rays_shadow == st1.rays_shadow and st.paths == st1.paths
        assert st.ms_render > 0
    gN.close()
    F.check(F.core().rt_init(-1))
    assert F.core().rt_device_count() == 1


def test_group_with_more_devices_than_tiles_and_planes(need_gpu):
    from plane_cases import two_object_scene
    spec = two_object_scene(width=24, height=9, spp=2, max_bounces=3, antialiasing=True)     # 3 x 2 tiles, 8 devices: some get none
    g1 = api.GpuScene(spec)
    ref, _ = g1.render()
    g1.close()
    gN = api.GpuScene(spec, devices=[0] * 8)
    res, _ = gN.render()
    gN.close()
    F.check(F.core().rt_init(-1))
    for k in ref:
        assert np.array_equal(res[k], ref[k]), k


def test_init_devices_rejects_bad_ids(need_gpu):
    bad = np.array([0, 99], np.int32)
    assert F.core().rt_init_devices(2, bad.ctypes.data_as(F.c_int32_p)) == F.RT_ERR_INVALID
    F.check(F.core().rt_init(-1))
