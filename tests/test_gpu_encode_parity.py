"""GPU parity tests proper: HIP kernels through the C-ABI vs the CPU oracle, bit-exact
(tile bitmaps, reject mask, tile definitions, quantised indices, dst planes)."""
import numpy as np
import pytest

from tests.images import edge_image, synth_planes
from tests.parity import compare_encode

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", params=[2, 1], ids=["kernel_v2", "kernel_v1"])
def hip(request):
    """Both implementations of the fused kernel (lane per 4x4 cell / lane per pixel row) must match the oracle."""
    from yaik_amd._lib import lib
    from yaik_amd.encoder import HipTileEncoder
    e = HipTileEncoder(0)
    assert lib().yk_set_kernel_version(e._h, request.param) == 0
    yield e
    e.close()


@pytest.mark.parametrize("size,npl", [(64, 3), (128, 3), (256, 3), (256, 4), (512, 4), (1024, 4)])
@pytest.mark.parametrize("m3", [False, True])
def test_synth_bit_exact(hip, oracle_built, size, npl, m3):
    bad = compare_encode(synth_planes(size, n_planes=npl), hip, m3)
    assert not bad, bad


@pytest.mark.parametrize("kind", ["flat", "noise", "ramp", "smooth", "mixed", "white", "dark", "twocolor", "photo"])
@pytest.mark.parametrize("wh,npl", [((64, 64), 3), ((128, 128), 4), ((72, 40), 3), ((200, 136), 3), ((256, 256), 4)])
def test_edge_images_bit_exact(hip, oracle_built, kind, wh, npl):
    bad = compare_encode(edge_image(wh[0], wh[1], kind, npl), hip, False)
    assert not bad, bad


@pytest.mark.parametrize("m3", [False, True])
def test_photo_like_1024_bit_exact(hip, oracle_built, m3):
    """Textured content: tiles of every span, i.e. every rangeDecode slab of the quantiser table, incl. exact ties between modes."""
    bad = compare_encode(edge_image(1024, 1024, "photo", 4, seed=3), hip, m3)
    assert not bad, bad


@pytest.mark.parametrize("case", ["synth256x4", "mixed128x4", "twocolor128", "noise64"])
def test_exact_resummation_path_bit_exact(oracle_built, case):
    """kernel v2 screens the mode-selection sums in tree order and falls back to the reference's sequential order only
    for ambiguous tiles; flag 16 forces that fallback for every tile, and the result must still be bit-exact."""
    from yaik_amd._lib import lib
    from yaik_amd.encoder import HipTileEncoder
    planes = {"synth256x4": lambda: synth_planes(256, n_planes=4), "mixed128x4": lambda: edge_image(128, 128, "mixed", 4),
              "twocolor128": lambda: edge_image(128, 128, "twocolor", 3), "noise64": lambda: edge_image(64, 64, "noise", 3)}[case]()
    e = HipTileEncoder(0)
    try:
        lib().yk_set_kernel_version(e._h, 2)
        lib().yk_set_ablation(e._h, 16)
        for m3 in (False, True):
            bad = compare_encode(planes, e, m3)
            assert not bad, bad
    finally:
        e.close()


@pytest.mark.parametrize("rf", [0, 1, 2, 5, 9, 40, 64])
def test_other_reject_factors_bit_exact(hip, oracle_built, rf):
    """FittingQuadSmooth's rejectFactor is an argument of the operator (the shipped encoder always passes 3): the range tests on
    D = S' - 256*cur and the second-difference early-out are parametrised by it.  64 is the largest factor the C-ABI admits (the packed
    16-bit tests of yk_encode2_kernel need 256*rf + 255 inside int16)."""
    from oracle.pyoracle import PASSES, OracleEncoder
    from tests.parity import cells_from_plane
    for planes in (synth_planes(256, n_planes=3), edge_image(128, 128, "smooth", 3), edge_image(128, 128, "mixed", 3)):
        ora = OracleEncoder(planes)
        hip.set_image(planes)
        hip.encode(rf, False, False)
        for i, (sx, sy) in enumerate(PASSES):
            cnt, bm, rgb = ora.fitting_quad_smooth(sx, sy, reject_factor=rf)
            assert np.array_equal(hip.gradient_bitmap(i), bm), (rf, i)
        assert np.array_equal(hip.coverage(), cells_from_plane(ora.state("smoothMap")))
        for p in range(3):
            defs, nib, nn, dst = ora.dynamic_tile_encode(p, False)
            d2, n2, nn2 = hip.range_streams(p)
            assert nn2 == nn and np.array_equal(d2, defs) and np.array_equal(n2, nib), (rf, p)
