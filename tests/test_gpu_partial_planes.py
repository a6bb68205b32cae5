"""GPU partial-plane gradient passes through the C-ABI: yk_gradient_partial_pass (FittingQuadSmooth with NULL planes) and
yk_decode_gradient_planes (DecompressGradient4x4 with planeBit 1..6) vs the CPU oracle and the reference fixtures (tests/golden/pp_*.npz)."""
import os

import numpy as np
import pytest

from oracle.pyoracle import PASSES, OracleDecoder, OracleEncoder, palette_remap
from tests.blobs import PP_MASKS
from tests.golden.make_golden import PARTIAL
from tests.images import edge_image, synth_planes

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture(scope="module")
def hip():
    from yaik_amd.encoder import HipTileEncoder
    e = HipTileEncoder(0)
    yield e
    e.close()


@pytest.fixture(scope="module")
def dec():
    from yaik_amd.decoder import HipTileDecoder
    d = HipTileDecoder(0)
    yield d
    d.close()


def cells(plane8: np.ndarray) -> np.ndarray:
    h, w = plane8.shape
    return plane8[: h // 4 * 4: 4, : w // 4 * 4: 4] != 0


ENC_CASES = {
    "planemix128": lambda: edge_image(128, 128, "planemix", 3),
    "planemix200x72": lambda: edge_image(200, 72, "planemix", 3),
    "planemix256_rgba": lambda: edge_image(256, 256, "planemix", 4),
    "mixed128_rgba": lambda: edge_image(128, 128, "mixed", 4),
    "photo256": lambda: edge_image(256, 256, "photo", 3),
    "synth512": lambda: synth_planes(512, n_planes=3),
    "planemix1024": lambda: edge_image(1024, 1024, "planemix", 3, seed=5),
}


@pytest.mark.parametrize("name", sorted(ENC_CASES))
def test_partial_passes_match_oracle(hip, oracle_built, name):
    planes = ENC_CASES[name]()
    n, h, w = planes.shape
    ora = OracleEncoder(planes)
    if n == 4:
        ora.mip_prefilter()
    for sx, sy in PASSES:
        ora.fitting_quad_smooth(sx, sy)
    hip.set_image(planes)
    if n == 4:
        hip.mip_prefilter()
    hip.encode(3, False, False)
    total = 0
    for m in PP_MASKS:
        cnt, bm, rgb = ora.fitting_quad_smooth(2, 2, plane_bit=m)
        gcnt, gbm, grgb = hip.fitting_quad_smooth_planes(m)
        assert gcnt == cnt, (m, gcnt, cnt)
        assert np.array_equal(gbm, bm), m
        assert np.array_equal(grgb, rgb), m
        total += cnt
    assert total > 0
    assert np.array_equal(hip.coverage(), cells(ora.state("smoothMap")))
    for p in range(3):
        assert np.array_equal(hip.coverage_plane(p), cells(ora.state("mapSmoothTile", p))), p
    for p in range(3):
        ora.dynamic_tile_compressor(p)
    pix, typ = ora.streams_1d()
    gpix, gtyp = hip.dynamic_tile_compressor()
    assert np.array_equal(gtyp, typ) and np.array_equal(gpix, pix)


@pytest.mark.parametrize("kind,w,h", [("planemix", 128, 128), ("mixed", 200, 136), ("photo", 256, 256)])
def test_preview_planes_match_oracle(hip, oracle_built, kind, w, h):
    """FittingQuadSmooth's testOutput planes (blendC6Exp of the accepted tiles), RGB passes and a plane-subset pass."""
    planes = edge_image(w, h, kind, 3)
    ora = OracleEncoder(planes)
    hip.set_image(planes)
    hip.encode(3, False, False)
    for sx, sy in PASSES:
        ora.fitting_quad_smooth(sx, sy)
    got = hip.gradient_preview(range(7))
    want = np.stack([ora.state("preview", p) for p in range(3)])
    assert np.array_equal(np.where(got == np.iinfo(np.int32).min, 0, got), want)
    assert (got != np.iinfo(np.int32).min).any()
    ora.fitting_quad_smooth(2, 2, plane_bit=5)
    hip.fitting_quad_smooth_planes(5)
    got = hip.gradient_preview([7])
    want = np.stack([ora.state("preview", p) for p in range(3)])
    assert np.array_equal(np.where(got == np.iinfo(np.int32).min, 0, got), want)


def test_partial_passes_other_shapes_and_state(hip, oracle_built):
    """Any tile shape the function takes, masks in another order, and the state rules: a new encode forgets the partial passes."""
    planes = edge_image(192, 128, "planemix", 3, seed=9)
    ora = OracleEncoder(planes)
    for sx, sy in PASSES:
        ora.fitting_quad_smooth(sx, sy)
    hip.set_image(planes)
    hip.encode(3, False, False)
    for m, sx, sy in ((1, 3, 3), (6, 3, 2), (2, 2, 3), (4, 2, 2), (3, 2, 2), (7, 2, 2)):
        cnt, bm, rgb = ora.fitting_quad_smooth(sx, sy, plane_bit=m)
        gcnt, gbm, grgb = hip.fitting_quad_smooth_planes(m, sx, sy)
        assert (gcnt, gbm.tobytes(), grgb.tobytes()) == (cnt, bm.tobytes(), rgb.tobytes()), (m, sx, sy)
    hip.encode(3, False, False)
    ora2 = OracleEncoder(planes)
    for sx, sy in PASSES:
        ora2.fitting_quad_smooth(sx, sy)
    for p in range(3):
        assert np.array_equal(hip.coverage_plane(p), cells(ora2.state("mapSmoothTile", p)))
        ora2.dynamic_tile_compressor(p)
    pix, typ = ora2.streams_1d()
    gpix, gtyp = hip.dynamic_tile_compressor()
    assert np.array_equal(gtyp, typ) and np.array_equal(gpix, pix)
    from yaik_amd._lib import YaikError as YaikHipError
    with pytest.raises(YaikHipError):
        hip.fitting_quad_smooth_planes(0)
    with pytest.raises(YaikHipError):
        hip.fitting_quad_smooth_planes(5, 1, 1)


@pytest.mark.parametrize("name", sorted(PARTIAL))
def test_encoder_partial_passes_match_reference_fixture(hip, name):
    """Bitmaps, counts, per-plane coverage and the 1-D streams against what the unmodified reference produced."""
    ref = dict(np.load(os.path.join(GOLD, name + ".npz")))
    planes = PARTIAL[name]()
    n, h, w = planes.shape
    hip.set_image(planes)
    if n == 4:
        hip.mip_prefilter()
    hip.encode(3, False, False)
    counts = np.frombuffer(ref["pp_counts"].tobytes(), np.int32)
    for i, m in enumerate(PP_MASKS):
        gcnt, gbm, _ = hip.fitting_quad_smooth_planes(m)
        assert gcnt == counts[i] and np.array_equal(gbm, ref[f"pp_bitmap_{i}"]), (i, m)
    assert np.array_equal(hip.coverage(), cells(ref["pp_smoothMap"].reshape(h, w)))
    for p in range(3):
        assert np.array_equal(hip.coverage_plane(p), cells(ref[f"pp_mapSmoothTile_{p}"].reshape(h, w)))
    gpix, gtyp = hip.dynamic_tile_compressor()
    assert np.array_equal(gpix, ref["d1_pix"]) and np.array_equal(gtyp, ref["d1_type"])


@pytest.mark.parametrize("name", sorted(PARTIAL))
def test_decoder_partial_planes_match_reference_fixture(dec, oracle_built, name):
    """DecompressGradient4x4(planeBit) on the reference's own bitmaps and dequantised colour streams, reference-exact marks."""
    ref = dict(np.load(os.path.join(GOLD, name + ".npz")))
    if "pp_dec_planes_grad" not in ref:
        pytest.skip("image is not a whole number of 16x16 tiles")
    planes = PARTIAL[name]()
    n, h, w = planes.shape
    ora = OracleEncoder(planes)
    if n == 4:
        ora.mip_prefilter()
    dec.begin(w, h)
    for sx, sy in PASSES:
        cnt, bm, rgb = ora.fitting_quad_smooth(sx, sy)
        if cnt:
            from oracle.pyoracle import palette_decompress
            dec.decompress_gradient(sx, sy, bm, palette_decompress(ora.palette_compress(rgb), rgb.size, 250))
    counts = np.frombuffer(ref["pp_counts"].tobytes(), np.int32)
    for i, m in enumerate(PP_MASKS):
        if counts[i]:
            dec.decompress_gradient_planes(m, ref[f"pp_bitmap_{i}"], ref[f"pp_rgbdq_{i}"], consistent_marks=False)
    assert np.array_equal(dec.planes().ravel(), ref["pp_dec_planes_grad"])
    assert np.array_equal(dec.tile4x4(True).ravel(), ref["pp_dec_tile4x4"])


@pytest.mark.parametrize("kind,w,h", [("planemix", 128, 128), ("planemix", 256, 192), ("mixed", 128, 128), ("photo", 256, 256)])
def test_partial_round_trip_consistent_marks(hip, dec, oracle_built, kind, w, h):
    """GPU encode (RGB passes, partial passes, 1-D path) -> GPU decode with consistent marks == the oracle's decode, all streams consumed."""
    planes = edge_image(w, h, kind, 3)
    hip.set_image(planes)
    hip.encode(3, False, False)
    od = OracleDecoder(w, h)
    dec.begin(w, h)
    for i, (sx, sy) in enumerate(PASSES):
        bm, rgb = hip.gradient_bitmap(i), hip.gradient_corners(i)
        if rgb.size:
            dq = palette_remap(rgb, 250)
            od.gradient(sx, sy, bm, dq)
            dec.decompress_gradient(sx, sy, bm, dq)
    od.split_masks()
    for m in PP_MASKS:
        cnt, bm, rgb = hip.fitting_quad_smooth_planes(m)
        if cnt:
            dq = palette_remap(rgb, 250)
            od.gradient_planes(m, bm, dq, consistent_marks=True)
            dec.decompress_gradient_planes(m, bm, dq, consistent_marks=True)
    assert np.array_equal(dec.planes(), od.planes())
    assert np.array_equal(dec.tile4x4(True).ravel(), od.tile4x4(True).ravel())
    pix, typ = hip.dynamic_tile_compressor()
    assert od.decode_1d(typ, pix) == (typ.size, pix.size)
    dec.decompress_1d(typ, pix)
    assert np.array_equal(dec.planes(), od.planes())
