"""Exhaustive on-device checks of the arithmetic shortcuts the kernels use, and corner-stream parity."""
import ctypes as C

import pytest

from tests.images import edge_image, synth_planes
from tests.parity import compare_encode

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def hip():
    from yaik_amd.encoder import HipTileEncoder
    e = HipTileEncoder(0, hooks=True)                        # yk_selftest lives in the test build (include/yaik_hip_test.h), same kernel sources
    yield e
    e.close()


def test_reciprocal_division_is_ieee_exact(hip):
    from yaik_amd._lib import test_lib as lib
    res = C.c_int(-1)
    assert lib().yk_selftest(hip._h, 0, C.byref(res)) == 0
    assert res.value == 0, f"{res.value} of 65536 (minDiff, value) pairs differ from __fdiv_rn"


def test_reciprocal_scale_division_is_exact(hip):
    from yaik_amd._lib import test_lib as lib
    res = C.c_int(-1)
    assert lib().yk_selftest(hip._h, 1, C.byref(res)) == 0
    assert res.value == 0, f"{res.value} (scale, diff) pairs differ from the integer division of DiffRangeEncode"


def test_reciprocal_model1_division_is_exact(hip):
    from yaik_amd._lib import test_lib as lib
    res = C.c_int(-1)
    assert lib().yk_selftest(hip._h, 2, C.byref(res)) == 0
    assert res.value == 0, f"{res.value} (n, delta) pairs differ from the integer division of GetValueModel1"


def test_model1_multiply_shift_is_exact(hip):
    """the 1-D range kernel codes a pixel as (v * A + B) >> 20 (yk_r1_magic): every delta, minCol and v against GetValueModel1's C expression"""
    from yaik_amd._lib import test_lib as lib
    res = C.c_int(-1)
    assert lib().yk_selftest(hip._h, 4, C.byref(res)) == 0
    assert res.value == 0, f"{res.value} (delta, minCol, v) triples differ from GetValueModel1"


def test_quantiser_table_matches_lut_scan(hip):
    """yk_encode2_kernel reads index / minDiff of a pixel from a table indexed by (rangeDecode, v - BN): for every (min, max) of a
    tile the LUTs built the reference's way must equal BN + K[rangeDecode], and every value in [min, max] must find in the table
    what the first-minimum scan of those LUTs finds."""
    from yaik_amd._lib import test_lib as lib
    res = C.c_int(-1)
    assert lib().yk_selftest(hip._h, 3, C.byref(res)) == 0
    assert res.value == 0, f"{res.value} table entries differ from the LUT scan"


CORNER_CASES = {
    "synth256x4": lambda: synth_planes(256, n_planes=4), "synth512x3": lambda: synth_planes(512, n_planes=3),
    "mixed128": lambda: edge_image(128, 128, "mixed"), "ramp200x136": lambda: edge_image(200, 136, "ramp"),
    "smooth256": lambda: edge_image(256, 256, "smooth"), "twocolor128": lambda: edge_image(128, 128, "twocolor"),
}


@pytest.mark.parametrize("case", sorted(CORNER_CASES))
def test_corner_streams_bit_exact(hip, oracle_built, case):
    bad = compare_encode(CORNER_CASES[case](), hip, False, want_dst=False, check_corners=True)
    assert not bad, bad


@pytest.mark.parametrize("case", ["synth256x4", "synth512x3", "mixed128", "ramp200x136", "twocolor128"])
def test_live_1d_range_path_bit_exact(hip, oracle_built, case):
    """a15: 3x DynamicTileCompressor streams (pixel bytes + color0/minCol/delta) vs the oracle."""
    import numpy as np
    from oracle.pyoracle import PASSES, OracleEncoder
    planes = CORNER_CASES[case]()
    ora = OracleEncoder(planes)
    if planes.shape[0] == 4:
        ora.mip_prefilter()
    for sx, sy in PASSES:
        ora.fitting_quad_smooth(sx, sy)
    for p in range(3):
        ora.dynamic_tile_compressor(p)
    opix, otyp = ora.streams_1d()
    hip.set_image(planes)
    if planes.shape[0] == 4:
        hip.mip_prefilter()
    hip.encode(3, False, False)
    pix, typ = hip.dynamic_tile_compressor()
    assert np.array_equal(typ, otyp)
    assert np.array_equal(pix, opix)
