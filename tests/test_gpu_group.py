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


def test_group_is_not_replaced_by_a_call_from_another_thread(need_gpu):
    """an entry point reached from a fresh OS thread (a GHC `safe` ccall on another capability) must not
    re-initialise the library and turn the group into [current device]."""
    import threading
    lib = F.core()
    ids = np.array([0, 0, 0], np.int32)
    F.check(lib.rt_init_devices(3, ids.ctypes.data_as(F.c_int32_p)))
    rcs = []

    def other():
        rcs.append(lib.rt_quantize_device(None, None, 0, None))       # runs ensure_init, then rejects the null buffers
        spec = scenes.c2_spheres(width=32, height=16, spp=1)
        g = api.GpuScene.__new__(api.GpuScene)                        # rt_scene_create without a preceding rt_init
        g.spec = spec
        import ctypes as C
        desc = F.rt_scene_desc()
        keep = []
        k, d, m = spec.all_objects()
        g.bvh = api.HostBvh(k, d, m, spec.leaf_threshold, spec.bvh_max_depth)
        g.bvh.fill(desc)
        api.fill_desc_common(desc, spec, keep)
        g._h = C.c_void_p()
        rcs.append(lib.rt_scene_create(C.byref(desc), C.byref(g._h)))
        rcs.append(lib.rt_device_count())
        g.close()

    t = threading.Thread(target=other)
    t.start()
    t.join()
    assert rcs == [F.RT_ERR_INVALID, F.RT_OK, 3], rcs
    assert lib.rt_device_count() == 3
    F.check(lib.rt_init(-1))


@pytest.mark.timeout(300)
def test_bench_single_process_group_line(need_gpu):
    """bench.py --single-process: the metric through rt_init_devices + ONE rt_render per frame (the path the Haskell host
    binds), here on two virtual devices of the one GPU."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--single-process", "--devices", "0,0",
                          "--workload", "c3", "--width", "320", "--height", "180", "--spp", "4", "--steps", "2", "--warmup", "1"],
                         capture_output=True, text=True, timeout=280, cwd=root)
    assert out.returncode == 0, out.stderr[-2000:]
    line = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["config"]["parallelism"].startswith("group2") and line["value"] > 0
