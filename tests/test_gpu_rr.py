"""GPU: the opt-in Russian roulette extension (rt_params.rr_start_bounce > 0; default 0 = the reference) against the
oracle's statement of it, path for path, and - clamps inactive - against the roulette-free image in the mean."""
import numpy as np
import pytest

from oracle import oracle as O
from raytracer_amd import api, scenes

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("rr", [1, 2, 3])
def test_rr_matches_the_oracle_path_for_path(need_gpu, rr):
    for spec in (scenes.c3_blob988(width=96, height=56, spp=6), scenes.c2_spheres(width=96, height=54, spp=6),
                 scenes.c4_shipped(width=96, height=54, spp=4, sky_size=(128, 64))):
        g = api.GpuScene(spec)
        res, st = g.render(rr_start_bounce=rr, collect_stats=1)
        base, st0 = g.render(collect_stats=1)
        g.close()
        o = O.OracleScene(spec)
        ref, cnt = o.render(rr_start_bounce=rr, counters=True)
        o.close()
        assert (res["prim_id"] == ref["prim_id"]).all()
        rel = np.abs(res["radiance"] - ref["radiance"]) / np.maximum(1.0, np.abs(ref["radiance"]))
        assert rel.max() <= 1e-12, (spec.name, rr, rel.max())
        assert st.rays_closest == cnt["rays_closest"] < st0.rays_closest        # the same paths end early on both sides


def test_rr_is_unbiased_where_no_clamp_is_active(need_gpu):
    spec = scenes.c3_blob988(width=160, height=90, spp=2048, max_bounces=5)
    g = api.GpuScene(spec)
    off1, _ = g.render(want=("radiance",))
    off2, _ = g.render(want=("radiance",), seed=99)
    on, _ = g.render(want=("radiance",), rr_start_bounce=1)
    g.close()
    noise = np.sqrt(np.mean((off1["radiance"] - off2["radiance"]) ** 2))
    d = on["radiance"] - off1["radiance"]
    rms = np.sqrt(np.mean(d ** 2))
    assert rms <= 3.0 * noise
    assert abs(d.mean()) <= 4.0 * rms / np.sqrt(d.shape[0] * d.shape[1])


def test_terminated_paths_do_not_sample_the_sky(need_gpu):
    """a path the roulette terminates adds nothing deeper - in particular not 0 x background.  With an infinite
    texel in an RGBF sky that product would be NaN; the definition (the oracle) returns emitted + lit."""
    spec = scenes.c3_blob988(width=96, height=54, spp=8)
    sky = scenes.sky_gradient_texture(fmt="rgbf")
    img = sky[1].copy()
    img[: img.shape[0] // 2] = np.inf                       # the whole upper half of the sky
    spec.sky = ("rgbf", img)
    o = O.OracleScene(spec)
    ref = o.render(rr_start_bounce=1)
    o.close()
    g = api.GpuScene(spec)
    res, _ = g.render(rr_start_bounce=1)
    g.close()
    assert np.isfinite(ref["radiance"]).sum() > 0 and not np.isnan(ref["radiance"]).all()
    assert np.array_equal(np.isnan(res["radiance"]), np.isnan(ref["radiance"]))
    assert np.array_equal(np.isinf(res["radiance"]), np.isinf(ref["radiance"]))
    ok = np.isfinite(ref["radiance"])
    assert np.abs(res["radiance"][ok] - ref["radiance"][ok]).max() <= 1e-12 * max(1.0, np.abs(ref["radiance"][ok]).max())
