"""Randomised parity: many small images whose 8x8 tiles are drawn from the regimes where the encoder's shortcuts change
behaviour — ranges straddling 15/16 and 32 (where rangeDecode leaves its minimum), values >= 223 (LUT entries reach 256), zeros (terms the
reference skips), two- and three-level tiles (exact ties between modes), gradients with +-3/+-4 deviations (the accept
boundary of FittingQuadSmooth), random alpha masks per 16x16 tile.  HIP (both kernel generations) vs the oracle, bit-exact."""
import numpy as np
import pytest

from tests.parity import compare_encode

pytestmark = pytest.mark.gpu


def _tile_mix(rng, size):
    """RGB image [3, size, size] assembled from 8x8 tiles of random regimes."""
    out = np.zeros((3, size, size), np.int64)
    y, x = np.mgrid[0:8, 0:8]
    for ty in range(size // 8):
        for tx in range(size // 8):
            kind = rng.integers(0, 9)
            base = rng.integers(0, 256, 3)
            if kind == 0:                                   # range exactly 14..17 around a random base
                r = rng.integers(14, 18)
                t = base[:, None, None] + rng.integers(0, r + 1, (3, 8, 8))
            elif kind == 1:                                 # bright: min >= 223
                t = rng.integers(223, 256, (3, 8, 8))
            elif kind == 2:                                 # zeros and small values
                t = rng.integers(0, 4, (3, 8, 8)) * rng.integers(0, 2, (3, 8, 8))
            elif kind == 3:                                 # two levels
                a, b = rng.integers(0, 256, (2, 3))
                sel = rng.integers(0, 2, (8, 8))
                t = np.where(sel[None], a[:, None, None], b[:, None, None])
            elif kind == 4:                                 # three levels close together
                lv = base[:, None] + np.array([0, 1, 2])[None] * rng.integers(1, 12)
                t = lv[np.arange(3)[:, None, None], rng.integers(0, 3, (8, 8))[None]]
            elif kind == 5:                                 # smooth gradient +- deviations at the accept boundary
                gx, gy = rng.integers(-4, 5, (2, 3))
                t = base[:, None, None] + (gx[:, None, None] * x[None] + gy[:, None, None] * y[None]) // 2 + rng.integers(-4, 5, (3, 8, 8)) * (rng.integers(0, 6, (3, 8, 8)) == 0)
            elif kind == 6:                                 # full-range noise
                t = rng.integers(0, 256, (3, 8, 8))
            elif kind == 7:                                 # flat
                t = np.broadcast_to(base[:, None, None], (3, 8, 8))
            else:                                           # mid range 20..60
                r = rng.integers(20, 61)
                t = base[:, None, None] // 2 + rng.integers(0, r + 1, (3, 8, 8))
            out[:, ty * 8:ty * 8 + 8, tx * 8:tx * 8 + 8] = t
    return np.clip(out, 0, 255)


def _image(seed, size, n_planes):
    rng = np.random.default_rng(seed)
    rgb = _tile_mix(rng, size)
    if seed % 3 == 0:                                       # a smooth background so that gradient tiles of every size appear
        y, x = np.mgrid[0:size, 0:size]
        bg = np.stack([(x * 200) // size + 20, (y * 180) // size + 30, ((x + y) * 100) // size + 10])
        keep = np.repeat(np.repeat(rng.integers(0, 3, (size // 16, size // 16)) == 0, 16, 0), 16, 1)
        rgb = np.where(keep[None], rgb, bg)
    planes = [rgb[0], rgb[1], rgb[2]]
    if n_planes == 4:
        a = np.repeat(np.repeat(rng.integers(0, 3, (size // 16, size // 16)) != 0, 16, 0), 16, 1).astype(np.int64) * 255
        a[:16, :] = 0                                       # a transparent border so that the bbox shrinks (otherwise rejects are discarded)
        a[:, :16] = 0
        sparse = rng.integers(0, 40, (size, size)) == 0     # a few isolated non-zero alphas inside rejected tiles
        a = np.where((a == 0) & sparse & (np.arange(size)[None] >= 16) & (np.arange(size)[:, None] >= 16) & (seed % 2 == 0), 3, a)
        planes.append(a)
    return np.ascontiguousarray(np.stack(planes).astype(np.int32))


@pytest.fixture(scope="module", params=[2, 1], ids=["kernel_v2", "kernel_v1"])
def hip(request):
    from tests.parity import encoder_for_kernel_version
    e = encoder_for_kernel_version(request.param)
    yield e
    e.close()


@pytest.mark.parametrize("seed", range(24))
def test_random_tile_regimes_bit_exact(hip, oracle_built, seed):
    size = (64, 128, 256)[seed % 3]
    planes = _image(1000 + seed, size, 4 if seed % 2 else 3)
    for m3 in (False, True):
        bad = compare_encode(planes, hip, m3, want_dst=(seed % 4 == 0), check_corners=True)
        assert not bad, (seed, m3, bad)
    # the live 1-D range path on what the gradient passes left uncovered (a15)
    from oracle.pyoracle import PASSES, OracleEncoder
    ora = OracleEncoder(planes)
    if planes.shape[0] == 4:
        ora.mip_prefilter()
    for sx, sy in PASSES:
        ora.fitting_quad_smooth(sx, sy)
    for p in range(3):
        ora.dynamic_tile_compressor(p)
    opix, otyp = ora.streams_1d()
    gpix, gtyp = hip.dynamic_tile_compressor()
    assert np.array_equal(gpix, opix) and np.array_equal(gtyp, otyp), seed
    # and through the fused kernel's pixel cache (yk_set_pixel_cache: 4 B per uncovered pixel instead of the planes)
    hip.set_pixel_cache(True)
    try:
        hip.encode(3, True, False)
        gpix, gtyp = hip.dynamic_tile_compressor()
        assert np.array_equal(gpix, opix) and np.array_equal(gtyp, otyp), ("pixel cache", seed)
    finally:
        hip.set_pixel_cache(False)
