"""CPU suite: FittingQuadSmooth with NULL planes + DecompressGradient4x4(planeBit 1..6), oracle vs the unmodified reference
(committed fixtures tests/golden/pp_*.npz from `ref_driver ... partial`; live where the reference build exists)."""
import os

import numpy as np
import pytest

from oracle.refrun import have_ref, run_reference
from tests.blobs import compare_partial, oracle_partial_blobs
from tests.golden.make_golden import PARTIAL
from tests.images import edge_image, synth_planes

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.mark.parametrize("name", sorted(PARTIAL))
def test_oracle_partial_passes_match_fixture(oracle_built, name):
    ref = {k: v.tobytes() for k, v in np.load(os.path.join(GOLD, name + ".npz")).items()}
    assert np.frombuffer(ref["pp_counts"], np.int32).sum() > 0
    bad = compare_partial(ref, oracle_partial_blobs(PARTIAL[name]()))
    assert not bad, bad


LIVE = {
    "planemix200x72_rgb": lambda: edge_image(200, 72, "planemix", 3),        # partial tiles on both edges, h % 16 != 0
    "planemix128_rgb_seed3": lambda: edge_image(128, 128, "planemix", 3, seed=3),
    "photo256_rgb": lambda: edge_image(256, 256, "photo", 3),
    "synth256_rgb": lambda: synth_planes(256, n_planes=3),
}


@pytest.mark.skipif(not have_ref(), reason="oracle/_ref/ref_driver not built (needs /root/reference)")
@pytest.mark.parametrize("name", sorted(LIVE))
def test_oracle_partial_passes_match_reference_live(oracle_built, name):
    planes = LIVE[name]()
    ours = oracle_partial_blobs(planes)
    ref = run_reference(planes, partial=True)
    if "pp_dec_planes_grad" not in ours:                                      # decode loops need whole 16x16 tiles
        ref = {k: v for k, v in ref.items() if not k.startswith("pp_dec_")}
    bad = compare_partial(ref, ours)
    assert not bad, bad


def test_consistent_marks_round_trip(oracle_built):
    """The reference's single- and two-plane 4x4 decoders leave tile4x4Mask wrong (R/G/B never mark it, GB/RB put the B marks at
    tile4x4Mask + size/2; decoder/YAIK_Gradient.cpp:1420-2732), so its own Decompress1D then desyncs.  consistentMarks = 1 marks the
    planes the pass covered: the 1-D streams the encoder wrote are then consumed exactly."""
    from oracle.pyoracle import PASSES, OracleDecoder, OracleEncoder, palette_remap
    from tests.blobs import PP_MASKS
    planes = edge_image(128, 128, "planemix", 3)
    n, h, w = planes.shape
    enc = OracleEncoder(planes)
    dec = OracleDecoder(w, h)
    for sx, sy in PASSES:
        cnt, bm, rgb = enc.fitting_quad_smooth(sx, sy)
        if cnt:
            dec.gradient(sx, sy, bm, palette_remap(rgb, 250))
    dec.split_masks()
    for m in PP_MASKS:
        cnt, bm, rgb = enc.fitting_quad_smooth(2, 2, plane_bit=m)
        if cnt:
            dec.gradient_planes(m, bm, palette_remap(rgb, 250), consistent_marks=True)
    for p in range(3):
        enc.dynamic_tile_compressor(p)
    pix, typ = enc.streams_1d()
    assert dec.decode_1d(typ, pix) == (typ.size, pix.size)
