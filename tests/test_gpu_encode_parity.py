"""GPU parity tests proper: HIP kernels through the C-ABI vs the CPU oracle, bit-exact
(tile bitmaps, reject mask, tile definitions, quantised indices, dst planes)."""
import numpy as np
import pytest

from tests.images import edge_image, lineart_image, natural_photo, synth_planes
from tests.parity import compare_encode

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", params=[2, 1], ids=["kernel_v2", "kernel_v1"])
def hip(request):
    """Both implementations of the fused kernel (lane per 4x4 cell / lane per pixel row) must match the oracle."""
    from tests.parity import encoder_for_kernel_version
    e = encoder_for_kernel_version(request.param)
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


@pytest.mark.parametrize("m3", [False, True])
def test_non_synthetic_images_bit_exact(hip, oracle_built, m3):
    """Not YAIK-synth: a procedurally drawn flat-colour illustration with anti-aliased edges and an alpha surround (the content
    YAIK targets, 1024x1024 RGBA) and a crop of a natural photograph (256x256 RGB), corner streams included."""
    for planes in (lineart_image(1024, 4), natural_photo()):
        bad = compare_encode(planes, hip, m3, check_corners=True)
        assert not bad, bad


@pytest.mark.parametrize("name", ["lineart1024_rgba", "photo_astronaut256_rgb", "synth1024_rgba"])
def test_gpu_outputs_match_reference_hashes(oracle_built, name):
    """No oracle in between: SHA-256 of the HIP path's outputs against the hashes tests/golden/make_golden.py took from the compiled
    reference's own blobs (tile bitmaps, tile definitions / nibble streams of both start modes, 1-D streams, alpha bounds)."""
    import hashlib, json, os
    from tests.golden.make_golden import HASHED
    from yaik_amd.encoder import HipTileEncoder
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "hashes.json")) as f:
        want = json.load(f)[name]
    planes = HASHED[name]()
    sha = lambda a: hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()
    e = HipTileEncoder(0)
    try:
        for m in (0, 1):
            e.set_image(planes)
            if planes.shape[0] == 4:
                mip = e.mip_prefilter()
                assert sha(np.array(list(mip["bounds"]) + [16, mip["remaining"]], dtype=np.int32)) == want["mip_bounds"]
            e.encode(3, bool(m), False)
            if m == 0:
                for i in range(7):
                    assert sha(e.gradient_bitmap(i)) == want[f"grad_bitmap_{i}"], i
                assert e.gradient_counts().tolist() == want["grad_counts_values"]
            for p in range(3):
                defs, nib, nn = e.range_streams(p)
                assert sha(defs) == want[f"plnt_defs_{m}_{p}"] and sha(nib) == want[f"plnt_idx_{m}_{p}"], (m, p)
        pix, typ = e.dynamic_tile_compressor()
        assert sha(pix) == want["d1_pix"] and sha(typ) == want["d1_type"]
        # the same streams when the 1-D path reads the fused kernel's pixel cache (4 B per uncovered pixel) instead of the planes (yk_set_pixel_cache)
        e.set_pixel_cache(True)
        e.encode(3, True, False)
        pix2, typ2 = e.dynamic_tile_compressor()
        assert sha(pix2) == want["d1_pix"] and sha(typ2) == want["d1_type"]
    finally:
        e.close()


@pytest.mark.parametrize("case", ["synth256x4", "mixed128x4", "twocolor128", "noise64"])
def test_exact_resummation_path_bit_exact(oracle_built, case):
    """kernel v2 screens the mode-selection sums in tree order and falls back to the reference's sequential order only
    for ambiguous tiles; flag 16 forces that fallback for every tile, and the result must still be bit-exact."""
    from yaik_amd._lib import test_lib
    from yaik_amd.encoder import HipTileEncoder
    planes = {"synth256x4": lambda: synth_planes(256, n_planes=4), "mixed128x4": lambda: edge_image(128, 128, "mixed", 4),
              "twocolor128": lambda: edge_image(128, 128, "twocolor", 3), "noise64": lambda: edge_image(64, 64, "noise", 3)}[case]()
    e = HipTileEncoder(0, hooks=True)                        # the test build of the same kernel source: the switch exists only there
    try:
        test_lib().yk_set_ablation(e._h, 16)
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


def test_samples_outside_0_255_are_refused():
    """SURVEY 7 / include/yaik_hip.h: the path is defined for samples in 0..255 (the fused kernel keeps the low byte, the reference reads the whole
    int, framework.h:116-121).  yk_upload_planes refuses host planes that break it (YK_ERR_BAD_ARG, nothing bound); planes bound in place are
    checked on request."""
    import torch
    from yaik_amd._lib import YaikError
    from yaik_amd.encoder import HipTileEncoder
    planes = synth_planes(128, n_planes=3)
    e = HipTileEncoder(0)
    try:
        e.set_image(planes)
        assert e.validate_planes() == 0
        bad = planes.copy()
        bad[1, 17, 33] = 256
        bad[2, 100, 5] = -1
        with pytest.raises(YaikError, match="0..255"):
            e.set_image(bad)
        with pytest.raises(YaikError):
            e.encode(3, False, False)                            # nothing is bound after the refusal
        t = torch.from_numpy(bad).to("cuda").contiguous()
        e.set_image(t)                                           # bound in place: the caller vouches for the contents ...
        assert e.validate_planes() == 2                          # ... or asks
        e.set_image(planes)
        assert not compare_encode(planes, e, False)
    finally:
        e.close()
