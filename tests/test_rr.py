"""Russian roulette (BASELINE.json config 4; NOT in the reference, Config.hs:57) is an opt-in extension of the render
core, default off.  Its definition lives in the oracle (oracle.cpp traceLoop, rrStart) and in DESIGN.md section 9; here
the CPU suite checks the definition itself: off = the reference's estimator bit for bit, on = the same image in the
mean (clamps inactive), with fewer rays traced."""
import numpy as np

from oracle import oracle as O
from raytracer_amd import scenes


def render(spec, **over):
    o = O.OracleScene(spec)
    res, cnt = o.render(n_threads=8, counters=True, **over)
    o.close()
    return res, cnt


def test_rr_off_is_the_reference_and_on_is_unbiased():
    spec = scenes.c3_blob988(width=40, height=24, spp=1024, max_bounces=5)       # solid white background, one weak light: no clamp is active
    off1, c_off = render(spec)
    assert np.array_equal(render(spec, rr_start_bounce=0)[0]["radiance"], off1["radiance"])
    off2, _ = render(spec, seed=77)
    on, c_on = render(spec, rr_start_bounce=1)
    noise = np.sqrt(np.mean((off1["radiance"] - off2["radiance"]) ** 2))        # what two independent renders differ by
    d = on["radiance"] - off1["radiance"]
    rms = np.sqrt(np.mean(d ** 2))
    assert rms <= 3.0 * noise                                                    # roulette adds variance (about 2x here) ...
    n_px = d.shape[0] * d.shape[1]                                               # (a pixel's three channels are correlated)
    assert abs(d.mean()) <= 4.0 * rms / np.sqrt(n_px)                            # ... not bias: the signed mean difference is noise
    assert c_on["rays_closest"] < 0.95 * c_off["rays_closest"]                   # and paths do end early
