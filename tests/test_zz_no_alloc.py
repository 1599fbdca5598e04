"""GPU: rt_render_tiles_device_noalloc, the entry point that allocates nothing (its workspace is a static device array
inside librtcore), against rt_render_tiles_device through the C ABI: identical buffers, bit for bit, on two calls in a
row - the second with a smaller input, so it runs over what the first left in the shared array."""
import numpy as np
import pytest

from raytracer_amd import api, scenes
from raytracer_amd import dist as rdist

pytestmark = pytest.mark.gpu


def render_packed(g, rank, world, noalloc, **over):
    import torch
    W, H = over.get("width", g.spec.width), over.get("height", g.spec.height)
    n = rdist.max_tiles(W, H, world)
    col = torch.full((n * 64, 3), float("nan"), dtype=torch.float64, device="cuda")
    rad = torch.full((n * 64, 3), float("nan"), dtype=torch.float64, device="cuda")
    ids = torch.full((n * 64,), -7, dtype=torch.int32, device="cuda")
    stream = torch.cuda.current_stream().cuda_stream
    st = g.render_tiles_device(rank, world, d_color=col.data_ptr(), d_radiance=rad.data_ptr(), d_prim_id=ids.data_ptr(),
                               stream=stream, noalloc=noalloc, **over)
    torch.cuda.synchronize()
    return col.cpu().numpy(), rad.cpu().numpy(), ids.cpu().numpy(), st


def same(a, b):
    for x, y in zip(a[:3], b[:3]):
        assert np.array_equal(x, y, equal_nan=True)


def test_noalloc_equals_allocating_entry_point(need_gpu):
    big = scenes.c3_blob988(width=203, height=117, spp=24)        # 380 tiles x 64 x 24 spp x 24 B = 14 MB of samples: one launch
    g = api.GpuScene(big)
    ref = render_packed(g, 0, 1, noalloc=False, collect_stats=1)
    out = render_packed(g, 0, 1, noalloc=True, collect_stats=1)
    same(out, ref)
    assert out[3].rays_closest == ref[3].rays_closest and out[3].nodes_visited == ref[3].nodes_visited
    # second call, smaller input, another rank layout: runs over the first call's leftovers in the static array
    ref2 = render_packed(g, 1, 3, noalloc=False, width=96, height=56, samples_per_pixel=3)
    out2 = render_packed(g, 1, 3, noalloc=True, width=96, height=56, samples_per_pixel=3)
    same(out2, ref2)
    g.close()


def test_noalloc_splits_frames_larger_than_its_sample_area(need_gpu):
    spec = scenes.c3_blob988(width=320, height=180, spp=100)      # 920 tiles x 64 x 100 x 24 B = 141 MB > 64 MiB: three launches
    g = api.GpuScene(spec)
    same(render_packed(g, 0, 1, noalloc=True), render_packed(g, 0, 1, noalloc=False))
    same(render_packed(g, 0, 1, noalloc=True, width=64, height=40, samples_per_pixel=2),
         render_packed(g, 0, 1, noalloc=False, width=64, height=40, samples_per_pixel=2))
    g.close()


def test_noalloc_rejects_what_does_not_fit(need_gpu):
    from raytracer_amd import _ffi as F
    spec = scenes.c3_blob988(width=64, height=40, spp=2)
    g = api.GpuScene(spec)
    with pytest.raises(F.RtError) as e:
        render_packed(g, 0, 1, noalloc=True, samples_per_pixel=50000)          # one tile's samples: 76.8 MB > 64 MiB
    assert e.value.code == F.RT_ERR_UNSUPPORTED and "static scratch" in str(e.value)
    g.close()


def test_noalloc_call_allocates_nothing(need_gpu):
    """nothing asserted that the entry point allocates nothing.  Free device memory (hipMemGetInfo through
    torch) must not move across noalloc calls - while the allocating twin, on a fresh scene, visibly grows its workspace -
    and with collect_stats = 0 the call must be legal under HIP stream capture (no synchronisation, no blocking copy)."""
    import torch
    spec = scenes.c3_blob988(width=203, height=117, spp=24)
    g = api.GpuScene(spec)
    n = rdist.max_tiles(spec.width, spec.height, 1)
    col = torch.zeros((n * 64, 3), dtype=torch.float64, device="cuda")
    stream = torch.cuda.current_stream().cuda_stream
    torch.cuda.synchronize()
    free0 = torch.cuda.mem_get_info()[0]
    for _ in range(2):
        g.render_tiles_device(0, 1, d_color=col.data_ptr(), stream=stream, noalloc=True)
    torch.cuda.synchronize()
    assert torch.cuda.mem_get_info()[0] == free0
    g.render_tiles_device(0, 1, d_color=col.data_ptr(), stream=stream, noalloc=False)      # grows the scene's own workspace
    torch.cuda.synchronize()
    assert torch.cuda.mem_get_info()[0] < free0
    # under capture: a side stream, the call recorded into a graph, the graph replayed
    ref = col.clone()
    col.zero_()
    s = torch.cuda.Stream()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.stream(s):
        gr.capture_begin()
        g.render_tiles_device(0, 1, d_color=col.data_ptr(), stream=s.cuda_stream, noalloc=True)
        gr.capture_end()
    torch.cuda.synchronize()
    gr.replay()
    torch.cuda.synchronize()
    assert torch.equal(col, ref)
    g.close()
