"""GPU: the sharded entry point (rt_render_tiles_device + rt_untile_device) reproduces rt_render bit-for-bit for
every world size — the property that makes 1/2/4/8-GPU frames identical (RNG keyed on the global pixel)."""
import ctypes as C

import numpy as np
import pytest

from raytracer_amd import _ffi as F
from raytracer_amd import api, scenes
from raytracer_amd import dist as rdist

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("world", [1, 2, 3, 8])
def test_sharded_render_is_bit_identical(need_gpu, world):
    import torch
    spec = scenes.c3_blob988(width=203, height=117, spp=3)
    g = api.GpuScene(spec)
    full, _ = g.render(want=("color", "radiance", "prim_id"))
    W, H = spec.width, spec.height
    dev = torch.device("cuda", 0)
    n_max = rdist.max_tiles(W, H, world)
    image = torch.full((H, W, 3), float("nan"), dtype=torch.float64, device=dev)
    rad = torch.full((H, W, 3), float("nan"), dtype=torch.float64, device=dev)
    stream = torch.cuda.current_stream().cuda_stream
    for rank in range(world):                      # one GPU plays every rank in turn
        packed = torch.zeros((n_max * 64, 3), dtype=torch.float64, device=dev)
        prad = torch.zeros((n_max * 64, 3), dtype=torch.float64, device=dev)
        g.render_tiles_device(rank, world, d_color=packed.data_ptr(), d_radiance=prad.data_ptr(), stream=stream)
        rdist.untile_into(image, packed, W, H, rank, world)
        rdist.untile_into(rad, prad, W, H, rank, world)
    torch.cuda.synchronize()
    g.close()
    assert np.array_equal(image.cpu().numpy(), full["color"])
    assert np.array_equal(rad.cpu().numpy(), full["radiance"])


@pytest.mark.parametrize("world", [1, 3, 8])
def test_eight_bit_frame_gathered_as_bytes(need_gpu, world):
    """bench.py's exchange: every rank quantises its packed tiles (rt_quantize_device) and the 8-bit tiles are un-tiled
    as bytes (rt_untile_bytes_device): the frame must equal rt_render's rgb8 for every world size."""
    import torch
    spec = scenes.c3_blob988(width=203, height=117, spp=2)
    g = api.GpuScene(spec)
    full, _ = g.render(want=("rgb8",))
    W, H = spec.width, spec.height
    n_max = rdist.max_tiles(W, H, world)
    rgb8 = torch.full((H, W, 3), 7, dtype=torch.uint8, device="cuda")
    stream = torch.cuda.current_stream().cuda_stream
    for rank in range(world):
        packed = torch.zeros((n_max * 64, 3), dtype=torch.float64, device="cuda")
        packed8 = torch.zeros((n_max * 64, 3), dtype=torch.uint8, device="cuda")
        g.render_tiles_device(rank, world, d_color=packed.data_ptr(), stream=stream)
        F.check(F.core().rt_quantize_device(C.c_void_p(packed.data_ptr()), C.c_void_p(packed8.data_ptr()), packed.numel(), C.c_void_p(stream)))
        rdist.untile_into(rgb8, packed8, W, H, rank, world)
    torch.cuda.synchronize()
    g.close()
    assert np.array_equal(rgb8.cpu().numpy(), full["rgb8"])


@pytest.mark.parametrize("world", [1, 2, 3, 8])
def test_fused_eight_bit_tiles_and_one_launch_untile(need_gpu, world):
    """bench.py's step: rt_render_tiles_rgb8_device writes the rank's packed 8-bit tiles in the resolve pass (no colour buffer,
    no quantise launch); the ranks' buffers, laid side by side as the gather leaves them (dist.make_gather_buffers), are
    un-tiled by ONE rt_untile_all_bytes_device launch.  The frame must equal rt_render's rgb8, and the same launch must
    reproduce the colour frame from packed doubles (24-byte slots)."""
    import torch
    spec = scenes.c3_blob988(width=203, height=117, spp=2)
    g = api.GpuScene(spec)
    full, _ = g.render(want=("rgb8", "color"))
    W, H = spec.width, spec.height
    n_max = rdist.max_tiles(W, H, world)
    stream = torch.cuda.current_stream().cuda_stream
    packed8 = torch.zeros((n_max * 64, 3), dtype=torch.uint8, device="cuda")
    bufs8 = rdist.make_gather_buffers(packed8, world)
    packed = torch.zeros((n_max * 64, 3), dtype=torch.float64, device="cuda")
    bufs = rdist.make_gather_buffers(packed, world)
    for rank in range(world):
        g.render_tiles_device(rank, world, d_rgb8=bufs8[rank].data_ptr(), stream=stream)
        g.render_tiles_device(rank, world, d_color=bufs[rank].data_ptr(), stream=stream)
    rgb8 = torch.full((H, W, 3), 7, dtype=torch.uint8, device="cuda")
    col = torch.full((H, W, 3), float("nan"), dtype=torch.float64, device="cuda")
    F.check(F.core().rt_untile_all_bytes_device(W, H, world, C.c_void_p(bufs8.big.data_ptr()), n_max * 64 * 3,
                                                C.c_void_p(rgb8.data_ptr()), 3, C.c_void_p(stream)))
    F.check(F.core().rt_untile_all_bytes_device(W, H, world, C.c_void_p(bufs.big.data_ptr()), n_max * 64 * 24,
                                                C.c_void_p(col.data_ptr()), 24, C.c_void_p(stream)))
    torch.cuda.synchronize()
    g.close()
    assert np.array_equal(rgb8.cpu().numpy(), full["rgb8"])
    assert np.array_equal(col.cpu().numpy(), full["color"])
    with pytest.raises(F.RtError):                           # a stride that cannot hold a rank's tiles is refused on the host
        F.check(F.core().rt_untile_all_bytes_device(W, H, world, C.c_void_p(bufs8.big.data_ptr()), 64, C.c_void_p(rgb8.data_ptr()), 3, None))


def test_quantize_device_matches_host_rule(need_gpu):
    import torch
    c = torch.tensor([0.0, 0.5, 0.999999, 1.0, 0.0039, 1.7, -0.2, float("nan")], dtype=torch.float64, device="cuda")
    q = torch.zeros(8, dtype=torch.uint8, device="cuda")
    F.check(F.core().rt_quantize_device(C.c_void_p(c.data_ptr()), C.c_void_p(q.data_ptr()), 8, None))
    torch.cuda.synchronize()
    assert q.cpu().tolist() == [0, 127, 255, 255, 0, 255, 0, 0]      # truncate(c*255.999), clamped to 0..255


def test_sample_buffer_passes_do_not_change_the_frame(need_gpu, monkeypatch):
    """A frame whose per-path radiance does not fit the sample buffer is rendered in several launches (tile ranges);
    k_resolve sums every pixel's samples in sample order, so the split is invisible.  1 MB holds 6 tiles at 112 spp."""
    spec = scenes.c3_blob988(width=130, height=75, spp=112)
    g = api.GpuScene(spec)
    one, _ = g.render(want=("color", "radiance", "prim_id"))
    g.close()
    monkeypatch.setenv("RT_SAMPLE_BUDGET_MB", "1")
    g = api.GpuScene(spec)                                   # a new scene: its sample buffer is sized by the 1-MB budget
    many, st = g.render(want=("color", "radiance", "prim_id"), collect_stats=1)
    g.close()
    for k in ("color", "radiance", "prim_id"):
        assert np.array_equal(one[k], many[k]), k
    assert st.paths == 130 * 75 * 112
