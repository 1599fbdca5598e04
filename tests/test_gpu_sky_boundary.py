"""Sky texels AT texel boundaries (SkySphere.hs:35-41): frames whose primary rays sweep through a column or row boundary of the
equirect sky in steps of 1e-17 ... 1e-14 - where the texel is decided by how GHC's atan2, the division by 2 pi, the sum and the
product round.  Every pixel must equal the oracle's (tools/sky_boundary_probe.py; DESIGN.md section 8, sky texel)."""
import os
import sys

import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))


@pytest.mark.gpu
@pytest.mark.parametrize("sw,sh", [(64, 32), (100, 50)])
def test_boundary_directions_get_the_reference_texel(need_gpu, sw, sh):
    import sky_boundary_probe as P
    bad, total = P.probe(sw, sh, steps=(1e-17, 1e-16, 1e-15, 1e-14), verbose=False)
    assert total > 1_000_000 and bad == 0
