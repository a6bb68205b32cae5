"""GPU parity tests proper: HIP kernels through the C-ABI vs the CPU oracle, bit-exact
(tile bitmaps, reject mask, tile definitions, quantised indices, dst planes)."""
import numpy as np
import pytest

from tests.images import edge_image, synth_planes
from tests.parity import compare_encode

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def hip():
    from yaik_amd.encoder import HipTileEncoder
    e = HipTileEncoder(0)
    yield e
    e.close()


@pytest.mark.parametrize("size,npl", [(64, 3), (128, 3), (256, 3), (256, 4), (512, 4), (1024, 4)])
@pytest.mark.parametrize("m3", [False, True])
def test_synth_bit_exact(hip, oracle_built, size, npl, m3):
    bad = compare_encode(synth_planes(size, n_planes=npl), hip, m3)
    assert not bad, bad


@pytest.mark.parametrize("kind", ["flat", "noise", "ramp", "smooth", "mixed", "white", "dark", "twocolor"])
@pytest.mark.parametrize("wh,npl", [((64, 64), 3), ((128, 128), 4), ((72, 40), 3), ((200, 136), 3), ((256, 256), 4)])
def test_edge_images_bit_exact(hip, oracle_built, kind, wh, npl):
    bad = compare_encode(edge_image(wh[0], wh[1], kind, npl), hip, False)
    assert not bad, bad
