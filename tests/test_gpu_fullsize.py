"""GPU parity at BASELINE.json's FULL sizes (C4: 63,432 triangles, 1920x1080, 100 spp; C5: 2,029,204 triangles,
3840x2160, 256 spp).  The oracle cannot render such a frame in seconds, but it can render ROWS of it: oracle and
kernel share the counter-based RNG keyed by (pixel, sample, bounce), so every pixel of the full-size GPU frame is
comparable path-for-path with the oracle's value for that pixel.  Plus the size-independent properties: the frame
is bit-identical for every tile sharding and for every split over sample-buffer launches.

Tolerance: the north_star bar is 1e-4 absolute on radiance; the library is built without FP contraction, so the
measured difference is ~1e-15 and the tests also pin that (<= 1e-9 on >= 99.9 % of the compared pixels)."""
import numpy as np
import pytest

from oracle import oracle as O
from raytracer_amd import api, scenes
from raytracer_amd import dist as rdist

pytestmark = pytest.mark.gpu


def compare_rows(spec, res, rows, n_threads=16):
    o = O.OracleScene(spec)
    worst, n_bad, n_px = 0.0, 0, 0
    for r in rows:
        ref = o.render(rows=(r, r + 1), n_threads=n_threads, want_ids=True)
        d = np.abs(res["radiance"][r] - ref["radiance"][r]).max(axis=1)
        assert np.isfinite(ref["radiance"][r]).all()
        worst = max(worst, float(d.max()))
        n_bad += int((d > 1e-9).sum())
        n_px += d.size
        assert (d <= 1e-4).mean() >= 0.999, f"row {r}: only {(d <= 1e-4).mean():.5f} within 1e-4"
        ids_ok = res["prim_id"][r] == ref["prim_id"][r]
        assert ids_ok.mean() >= 0.999, f"row {r}: primitive ids {ids_ok.mean():.5f}"
    o.close()
    assert n_bad <= 1e-3 * n_px, f"{n_bad} of {n_px} pixels differ by more than 1e-9 (worst {worst:.3e})"
    return worst


@pytest.mark.timeout(300)
def test_c4_full_size_rows_match_oracle(need_gpu):
    spec = scenes.c4_blob63k()                                 # 1920x1080, 100 spp, maxBounces 5: the bench workload
    assert (spec.width, spec.height, spec.spp) == (1920, 1080, 100)
    g = api.GpuScene(spec, n_threads=8)
    res, st = g.render(want=("radiance", "prim_id"), collect_stats=1)
    g.close()
    assert st.paths == 1920 * 1080 * 100
    assert np.isfinite(res["radiance"]).all()
    compare_rows(spec, res, rows=[3, 271, 540, 777, 1079])


def test_c4_full_size_sharding_is_bit_identical(need_gpu):
    import torch
    spec = scenes.c4_blob63k()
    W, H, world = spec.width, spec.height, 8
    g = api.GpuScene(spec, n_threads=8)
    full, _ = g.render(want=("color",))
    dev = torch.device("cuda", 0)
    image = torch.full((H, W, 3), float("nan"), dtype=torch.float64, device=dev)
    stream = torch.cuda.current_stream().cuda_stream
    for rank in range(world):                                  # one GPU plays the 8 ranks in turn
        packed = torch.zeros((rdist.max_tiles(W, H, world) * 64, 3), dtype=torch.float64, device=dev)
        g.render_tiles_device(rank, world, d_color=packed.data_ptr(), stream=stream)
        rdist.untile_into(image, packed, W, H, rank, world)
    torch.cuda.synchronize()
    g.close()
    assert np.array_equal(image.cpu().numpy(), full["color"])


@pytest.mark.timeout(600)
def test_c5_full_size_rows_match_oracle(need_gpu):
    spec = scenes.c5_multi()                                   # 2,029,204 triangles, 3840x2160, 256 spp
    assert (spec.width, spec.height, spec.spp) == (3840, 2160, 256)
    g = api.GpuScene(spec, n_threads=16)
    res, st = g.render(want=("radiance", "prim_id"))
    g.close()
    assert np.isfinite(res["radiance"]).all()
    compare_rows(spec, res, rows=[1080, 1700])
