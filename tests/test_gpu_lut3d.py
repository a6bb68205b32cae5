"""GPU 3-D LUT tile search through the C-ABI: pattern tables, the six search passes, their streams and maps, the per-plane
coverage they leave and the 1-D streams behind them -- against the CPU oracle and the reference fixtures (tests/golden/lut_*.npz)."""
import os

import numpy as np
import pytest

from oracle.pyoracle import PASSES, OracleEncoder
from tests.blobs import LUT_PASSES
from tests.golden.make_golden import LUT3D
from tests.lutbank import bank_patterns, lut_image, random_bank

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture(scope="module")
def hip():
    from yaik_amd.encoder import HipTileEncoder
    e = HipTileEncoder(0)
    yield e
    e.close()


def cells(plane8: np.ndarray) -> np.ndarray:
    h, w = plane8.shape
    return plane8[: h // 4 * 4: 4, : w // 4 * 4: 4] != 0


def test_pattern_tables_match_oracle(hip, oracle_built):
    """Load3DPattern's morton sort + Set3DPointCloud: factor tables, nearest-entry tables at 6/5/4/3 bits, distance field."""
    pats = bank_patterns()
    rng = np.random.default_rng(5)
    pats.append(rng.integers(0, 64, (33, 3)).astype(np.uint8))             # unordered points, duplicates likely
    pats.append(np.array([[7, 7, 7]], np.uint8))                           # a single point
    ora = OracleEncoder(lut_image(64, 64))
    hip.lut_clear()
    for k, p in enumerate(pats):
        assert ora.lut_load(p) == k and hip.lut_load(p) == k
        fac, dist, pos = ora.lut_tables(k)
        gfac, gdist, gpos = hip.lut_tables(k)
        assert np.array_equal(gfac, fac), k
        assert np.array_equal(gdist.astype(np.int32), dist), k
        assert np.array_equal(gpos, pos), k
    from yaik_amd._lib import YaikError
    with pytest.raises(YaikError):
        hip.lut_load(np.zeros((65, 3), np.uint8))
    with pytest.raises(YaikError):
        hip.lut_load(np.full((4, 3), 64, np.uint8))
    hip.lut_clear()


def test_bank_holds_64_patterns_and_refuses_the_65th(hip):
    from yaik_amd._lib import YaikError
    hip.lut_clear()
    for p in random_bank(106, 64):
        hip.lut_load(p)
    with pytest.raises(YaikError):                                            # "LUT 3D more than 64 entries", EncoderContext.cpp:7912
        hip.lut_load(random_bank(107, 1)[0])
    hip.lut_clear()


CASES = {
    "random_bank_a": lambda: (lut_image(128, 128, random_bank(101), seed=41), random_bank(101)),
    "random_bank_b": lambda: (lut_image(144, 112, random_bank(102, 7), seed=42), random_bank(102, 7)),
    "random_bank_c": lambda: (lut_image(256, 256, random_bank(104, 9), seed=43), random_bank(104, 9)),
    # a full bank: 64 patterns (the LDS tables of the search are sized by the bank, the pair list is at its longest)
    "full_bank_64": lambda: (lut_image(192, 160, random_bank(105, 64)[:6], seed=44), random_bank(105, 64)),
    "lut128": lambda: (lut_image(128, 128, seed=11), bank_patterns()),
    "lut200x136": lambda: (lut_image(200, 136, seed=5), bank_patterns()),
    "lut256_3patterns": lambda: (lut_image(256, 256, bank_patterns(3), seed=2), bank_patterns(3)),
    "lut512": lambda: (lut_image(512, 512, seed=9), bank_patterns()),
    "lut256_rgba": lambda: (np.concatenate([lut_image(256, 256, seed=4), np.full((1, 256, 256), 255, np.int32)]), bank_patterns()),
}


@pytest.mark.parametrize("name", sorted(CASES))
def test_lut_search_matches_oracle(hip, oracle_built, name):
    planes, pats = CASES[name]()
    n, h, w = planes.shape
    ora = OracleEncoder(planes)
    if n == 4:
        ora.mip_prefilter()
    for sx, sy in PASSES:
        ora.fitting_quad_smooth(sx, sy)
    hip.lut_clear()
    for p in pats:
        ora.lut_load(p); hip.lut_load(p)
    hip.set_image(planes)
    if n == 4:
        hip.mip_prefilter()
    hip.encode(3, False, False)
    ora.lut_start(); hip.lut_start()
    total = 0
    for sx, sy in LUT_PASSES:
        want = ora.lut_search(sx, sy)
        got = hip.lut_search(sx, sy)
        assert got == want, (sx, sy, got, want)
        total += want
    assert total > 0
    so, sg = ora.lut_streams(), hip.lut_streams()
    for k in ("tileType", "color", "idx3", "idx4", "idx5", "idx6") + tuple(f"map{i}" for i in range(6)):
        assert np.array_equal(sg[k], so[k]), k
    for p in range(3):
        assert np.array_equal(hip.coverage_plane(p), cells(ora.state("mapSmoothTile", p))), p
    assert np.array_equal(hip.coverage(), cells(ora.state("smoothMap")))          # LUT tiles never touch smoothMap
    for p in range(3):
        ora.dynamic_tile_compressor(p)
    pix, typ = ora.streams_1d()
    gpix, gtyp = hip.dynamic_tile_compressor()
    assert np.array_equal(gtyp, typ) and np.array_equal(gpix, pix)
    hip.lut_clear()


@pytest.mark.parametrize("name", sorted(LUT3D))
def test_lut_search_matches_reference_fixture(hip, name):
    ref = dict(np.load(os.path.join(GOLD, name + ".npz")))
    planes, pats = LUT3D[name]()
    hip.lut_clear()
    for p in pats:
        hip.lut_load(p)
    hip.set_image(planes)
    if planes.shape[0] == 4:
        hip.mip_prefilter()
    hip.encode(3, False, False)
    hip.lut_start()
    counts = np.frombuffer(ref["lut_counts"].tobytes(), np.int32).reshape(6, 6)
    for i, (sx, sy) in enumerate(LUT_PASSES):
        hip.lut_search(sx, sy)
        s = hip.lut_streams()
        assert [s["tileType"].size, s["color"].size, s["idx3"].size, s["idx4"].size, s["idx5"].size, s["idx6"].size] == counts[i].tolist(), i
    assert np.array_equal(s["tileType"].view(np.uint8), ref["lut_tileType"]) and np.array_equal(s["color"], ref["lut_color"])
    for bits in (3, 4, 5, 6):
        assert np.array_equal(s[f"idx{bits}"], ref[f"lut_idx{bits}"]), bits
    for k in range(6):
        assert np.array_equal(s[f"map{k}"], ref[f"lut_map_{k}"]), k
    n, h, w = planes.shape
    for p in range(3):
        assert np.array_equal(hip.coverage_plane(p), cells(ref[f"lut_mapSmoothTile_{p}"].reshape(h, w)))
    gpix, gtyp = hip.dynamic_tile_compressor()
    assert np.array_equal(gpix, ref["d1_pix"]) and np.array_equal(gtyp, ref["d1_type"])
    hip.lut_clear()


@pytest.fixture(scope="module")
def dec():
    from yaik_amd.decoder import HipTileDecoder
    d = HipTileDecoder(0)
    yield d
    d.close()


@pytest.mark.parametrize("name", sorted(LUT3D))
def test_lut_decode_matches_reference_fixture(hip, dec, oracle_built, name):
    """YAIK_AssignLUT + Tile3D_* + Decompress1D on the reference's own streams == the reference decoder's planes and mask."""
    from oracle.pyoracle import palette_decompress, palette_remap
    ref = dict(np.load(os.path.join(GOLD, name + ".npz")))
    planes, pats = LUT3D[name]()
    n, h, w = planes.shape
    ora = OracleEncoder(planes)
    dec.begin(w, h)
    for sx, sy in PASSES:
        cnt, bm, rgb = ora.fitting_quad_smooth(sx, sy)
        if cnt:
            dec.decompress_gradient(sx, sy, bm, palette_decompress(ora.palette_compress(rgb), rgb.size, 250))
    dec.assign_lut(ref["lut_file"])
    zin = [ref[f"lut_zin_{k}"] for k in range(len([k for k in ref if k.startswith("lut_zin_")]))]
    counts = np.frombuffer(ref["lut_counts"].tobytes(), np.int32).reshape(6, 6)[-1]
    it = iter(zin[6:])
    tiles = next(it).view(np.uint16) if counts[0] else np.zeros(0, np.uint16)
    colors = palette_remap(next(it), 250) if counts[1] else np.zeros(0, np.uint8)
    idx = [next(it) if counts[2 + f] else np.zeros(0, np.uint8) for f in range(4)]
    used = dec.decompress_lut3d(zin[:6], tiles, colors, idx)
    assert used.tolist() == np.frombuffer(ref["lut_dec_consumed"].tobytes(), np.int32).tolist()
    assert np.array_equal(dec.planes().ravel(), ref["lut_dec_planes"])
    assert np.array_equal(dec.tile4x4().ravel(), ref["lut_dec_tile4x4"])
    dec.decompress_1d(ref["d1_type"], ref["d1_pix"])
    assert np.array_equal(dec.planes().ravel(), ref["dec_planes_full"])


@pytest.mark.parametrize("w,h,seed", [(128, 128, 21), (256, 192, 22), (512, 512, 23)])
def test_lut_round_trip_on_gpu(hip, dec, oracle_built, w, h, seed):
    """GPU encode (gradient passes, 3-D LUT search, 1-D path) -> GPU decode == the oracle's decode of the same streams; PSNR stated."""
    from oracle.pyoracle import OracleDecoder, detile, palette_remap, yko_compress_f
    pats = bank_patterns()
    planes = lut_image(w, h, pats, seed)
    ora = OracleEncoder(planes)
    hip.lut_clear()
    for p in pats:
        hip.lut_load(p); ora.lut_load(p)
    hip.set_image(planes)
    hip.encode(3, False, False)
    od = OracleDecoder(w, h)
    dec.begin(w, h)
    for i, (sx, sy) in enumerate(PASSES):
        bm, rgb = hip.gradient_bitmap(i), hip.gradient_corners(i)
        if rgb.size:
            dq = palette_remap(rgb, 250)
            od.gradient(sx, sy, bm, dq); dec.decompress_gradient(sx, sy, bm, dq)
    hip.lut_start()
    for sx, sy in LUT_PASSES:
        hip.lut_search(sx, sy)
    s = hip.lut_streams()
    assert s["tileType"].size > 0
    lut_file = ora.lut_file()
    colors = palette_remap(yko_compress_f(s["color"], 250), 250)
    idx = [(s[f"idx{b}"].astype(np.uint16) * 3).astype(np.uint8) for b in (3, 4, 5, 6)]
    maps = [s[f"map{k}"] for k in range(6)]
    want_used = od.lut3d(lut_file, maps, s["tileType"], colors, idx)
    dec.assign_lut(lut_file)
    used = dec.decompress_lut3d(maps, s["tileType"], colors, idx)
    assert used.tolist() == want_used.tolist() == [s["tileType"].size * 2, s["color"].size] + [i.size for i in idx]
    assert np.array_equal(dec.planes(), od.planes()) and np.array_equal(dec.tile4x4().ravel(), od.tile4x4().ravel())
    pix, typ = hip.dynamic_tile_compressor()
    od.split_masks()
    assert od.decode_1d(typ, pix) == (typ.size, pix.size)
    dec.decompress_1d(typ, pix)
    gp = dec.planes()
    assert np.array_equal(gp, od.planes())
    rec = np.stack([detile(gp[c], w, h) for c in range(3)]).astype(np.int64)
    mse = float(np.mean((rec - planes) ** 2))
    assert 10 * np.log10(255.0 ** 2 / mse) > 35.0
    hip.lut_clear()


@pytest.mark.parametrize("cut", ["idx", "tiles"])
def test_lut_decode_refuses_streams_shorter_than_the_maps_need(hip, dec, oracle_built, cut):
    """A truncated / crafted '3DTL' chunk: the tile maps claim more tiles or index bytes than the streams hold.  The fill kernel reads the
    streams at offsets derived from the maps, so the call must be refused (YK_ERR_RANGE) before it runs: no read past a buffer, and the
    image planes as they were."""
    from oracle.pyoracle import palette_remap, yko_compress_f
    from yaik_amd._lib import YaikError
    pats = bank_patterns()
    planes = lut_image(128, 128, pats, 31)
    ora = OracleEncoder(planes)
    hip.lut_clear()
    for p in pats:
        hip.lut_load(p); ora.lut_load(p)
    hip.set_image(planes)
    hip.encode(3, False, False)
    hip.lut_start()
    for sx, sy in LUT_PASSES:
        hip.lut_search(sx, sy)
    s = hip.lut_streams()
    assert s["tileType"].size > 4
    colors = palette_remap(yko_compress_f(s["color"], 250), 250)
    idx = [(s[f"idx{b}"].astype(np.uint16) * 3).astype(np.uint8) for b in (3, 4, 5, 6)]
    maps = [s[f"map{k}"] for k in range(6)]
    dec.begin(128, 128)
    dec.assign_lut(ora.lut_file())
    before = dec.planes().copy()
    tiles = s["tileType"]
    if cut == "idx":
        f = max(range(4), key=lambda k: idx[k].size)
        idx[f] = idx[f][: idx[f].size // 2]
    else:
        tiles = tiles[: tiles.size // 2]; colors = colors[: tiles.size * 6]
    with pytest.raises(YaikError):
        dec.decompress_lut3d(maps, tiles, colors, idx)
    assert np.array_equal(dec.planes(), before)
    hip.lut_clear()
