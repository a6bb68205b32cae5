"""The C++ drop-in of the reference's operator surface (yaik_amd/host: Plane / Image / EncoderContext with the same
method names and call order) driven by a small C++ program, compared blob by blob with the CPU oracle."""
import os
import struct
import subprocess
import tempfile

import numpy as np
import pytest

from oracle.refrun import parse_blobs
from tests.blobs import oracle_blobs
from tests.images import edge_image, synth_planes

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DRIVER = os.path.join(ROOT, "yaik_amd", "host", "host_driver")


def _run(planes, mode3):
    if not os.path.exists(DRIVER):
        subprocess.run(["make", "-C", os.path.dirname(DRIVER)], check=True)
    n, h, w = planes.shape
    with tempfile.TemporaryDirectory() as d:
        fin, fout = os.path.join(d, "in.bin"), os.path.join(d, "out.blobs")
        with open(fin, "wb") as f:
            f.write(struct.pack("<3i", w, h, n)); f.write(np.ascontiguousarray(planes, np.int32).tobytes())
        subprocess.run([DRIVER, fin, fout, mode3 if isinstance(mode3, str) else ("1" if mode3 else "0")], check=True)
        return parse_blobs(fout)


@pytest.mark.parametrize("case,mode3", [("synth256x4", False), ("synth256x4", True), ("mixed128x4", False), ("ramp72x40", False), ("synth512x3", False)])
def test_cpp_operator_surface_matches_oracle(oracle_built, case, mode3):
    planes = {"synth256x4": lambda: synth_planes(256, n_planes=4), "mixed128x4": lambda: edge_image(128, 128, "mixed", 4),
              "ramp72x40": lambda: edge_image(72, 40, "ramp", 3), "synth512x3": lambda: synth_planes(512, n_planes=3)}[case]()
    got = _run(planes, mode3)
    want = oracle_blobs(planes)
    from oracle.pyoracle import PASSES, OracleEncoder
    ora = OracleEncoder(planes)
    if planes.shape[0] == 4:
        ora.mip_prefilter()
    bad = []
    for i, (sx, sy) in enumerate(PASSES):
        cnt, bm, rgb = ora.fitting_quad_smooth(sx, sy)
        if got[f"grad_bitmap_{i}"] != bm.tobytes():
            bad.append(f"bitmap{i}")
        if got[f"grad_rgbraw_{i}"] != rgb.tobytes():
            bad.append(f"rgb{i}")
    m = 1 if mode3 else 0
    names = ["grad_counts", "d1_pix", "d1_type"] + [f"plnt_{k}_{m}_{p}" for k in ("defs", "idx", "dst") for p in range(3)] + [f"preview_{p}" for p in range(3)]
    if planes.shape[0] == 4:
        names += ["mip_bounds", "_mip_bitmap"]
    for k in names:
        if got[k] != want[k]:
            bad.append(k)
    assert not bad, bad


@pytest.mark.parametrize("case", ["planemix128x3", "planemix256x4", "planemix200x72x3"])
def test_cpp_partial_plane_passes(oracle_built, case):
    """FittingQuadSmooth with NULL planes through the C++ mirror (the RB, RG, GB, R, G, B 4x4 calls of Convert(), :9261-9415), the 1-D
    compressor on the per-plane maps behind them, and the resulting .yaik stream decoded back through the YAIK_* API."""
    from oracle.pyoracle import PASSES, OracleDecoder, OracleEncoder, palette_decompress
    from tests.blobs import PP_MASKS
    planes = {"planemix128x3": lambda: edge_image(128, 128, "planemix", 3), "planemix256x4": lambda: edge_image(256, 256, "planemix", 4),
              "planemix200x72x3": lambda: edge_image(200, 72, "planemix", 3)}[case]()
    n, h, w = planes.shape
    got = _run(planes, "pp")
    ora = OracleEncoder(planes)
    if n == 4:
        ora.mip_prefilter()
    whole = w % 16 == 0 and h % 16 == 0
    dec = OracleDecoder(w, h) if whole else None
    for sx, sy in PASSES:
        cnt, bm, rgb = ora.fitting_quad_smooth(sx, sy)
        if cnt and dec:
            dec.gradient(sx, sy, bm, palette_decompress(ora.palette_compress(rgb), rgb.size, 250))
    if dec:
        dec.split_masks()
    counts = []
    for i, m in enumerate(PP_MASKS):
        cnt, bm, rgb = ora.fitting_quad_smooth(2, 2, plane_bit=m)
        counts.append(cnt)
        assert got[f"pp_bitmap_{i}"] == bm.tobytes() and got[f"pp_rgbraw_{i}"] == rgb.tobytes(), (i, m)
        if cnt and dec:
            dec.gradient_planes(m, bm, palette_decompress(ora.palette_compress(rgb), rgb.size, 250), consistent_marks=True)
    assert np.frombuffer(got["pp_counts"], np.int32).tolist() == counts and sum(counts) > 0
    ends = []
    for p in range(3):
        ora.dynamic_tile_compressor(p)
        ends.append(ora.streams_1d()[0].size)
    pix, typ = ora.streams_1d()
    assert got["d1_pix"] == pix.tobytes() and got["d1_type"] == typ.tobytes()
    assert np.frombuffer(got["d1_pix_ends"], np.int32).tolist() == ends
    if dec:
        assert dec.decode_1d(typ, pix) == (typ.size, pix.size)
        assert got["yaik_planes_tiled_1"] == dec.planes().tobytes()
        assert len(got["yaik_planes_tiled_0"]) == len(got["yaik_planes_tiled_1"])      # reference-exact marks: decodes, but desynchronised


@pytest.mark.parametrize("w,h,seed", [(128, 128, 31), (256, 192, 32)])
def test_cpp_lut3d_surface(oracle_built, w, h, seed):
    """(f)4 through the C++ mirror: Load3DPattern from bank files, StartCorrelationSearch, Correlation3DSearch x6, EndCorrelationSearch ('3DTL'),
    the 1-D compressor behind them, LutFile -> YAIK_AssignLUT -> YAIK_DecodeImage; decoded planes == the oracle's decode of the oracle's streams."""
    from oracle.pyoracle import PASSES, OracleDecoder, OracleEncoder, palette_decompress, palette_remap, yko_compress_f
    from tests.blobs import LUT_PASSES
    from tests.lutbank import bank_bytes, bank_patterns, lut_image
    pats = bank_patterns()
    planes = lut_image(w, h, pats, seed)
    with tempfile.TemporaryDirectory() as d:
        fb = os.path.join(d, "bank.bin")
        with open(fb, "wb") as f:
            f.write(bank_bytes(pats))
        fin, fout = os.path.join(d, "in.bin"), os.path.join(d, "out.blobs")
        with open(fin, "wb") as f:
            f.write(struct.pack("<3i", w, h, 3)); f.write(np.ascontiguousarray(planes, np.int32).tobytes())
        if not os.path.exists(DRIVER):
            subprocess.run(["make", "-C", os.path.dirname(DRIVER)], check=True)
        subprocess.run([DRIVER, fin, fout, "lut", fb], check=True)
        got = parse_blobs(fout)
    ora = OracleEncoder(planes)
    od = OracleDecoder(w, h)
    for sx, sy in PASSES:
        cnt, bm, rgb = ora.fitting_quad_smooth(sx, sy)
        if cnt:
            od.gradient(sx, sy, bm, palette_decompress(ora.palette_compress(rgb), rgb.size, 250))
    for p in pats:
        ora.lut_load(p)
    ora.lut_start()
    matched = [ora.lut_search(sx, sy) for sx, sy in LUT_PASSES]
    assert np.frombuffer(got["lut_matched"], np.int32).tolist() == matched and sum(matched) > 0
    assert got["lut_file"] == ora.lut_file().tobytes()
    s = ora.lut_streams()
    od.lut3d(ora.lut_file(), [s[f"map{k}"] for k in range(6)], s["tileType"], palette_remap(yko_compress_f(s["color"], 250), 250),
             [(s[f"idx{b}"].astype(np.uint16) * 3).astype(np.uint8) for b in (3, 4, 5, 6)])
    for p in range(3):
        ora.dynamic_tile_compressor(p)
    pix, typ = ora.streams_1d()
    assert got["d1_pix"] == pix.tobytes() and got["d1_type"] == typ.tobytes()
    od.split_masks()
    assert od.decode_1d(typ, pix) == (typ.size, pix.size)
    codes = np.frombuffer(got["yaik_lut_codes"], np.int32).tolist()
    assert codes[0] != 0 and codes[1] == 0, codes                # refused without the LUT, decoded with it
    assert got["yaik_planes_tiled"] == od.planes().tobytes()
    # crafted '3DTL' headers (a tile count that wraps the 32-bit "colours == 6 * tiles" product; a 3 GB index stream): both refused, with
    # an error code, before anything is expanded
    crafted = np.frombuffer(got["yaik_lut_crafted"], np.int32).tolist()
    assert crafted[0] == 0 and crafted[1] != 0 and crafted[2] == 0 and crafted[3] != 0, crafted


@pytest.mark.parametrize("case,n", [("synth512x4", 2), ("synth512x4", 8), ("synth1024x4", 3), ("mixed256x3", 4), ("synth256x4", 6)])
def test_cpp_row_stripes_equal_the_whole_image(case, n):
    """EncoderContext::ConvertHotPathStripes (n row stripes, one handle each; distinct devices are gathered by ONE grouped RCCL transfer,
    stripes that share the only device of this box are read back one by one) against the same C++ surface on the whole image."""
    planes = {"synth512x4": lambda: synth_planes(512, n_planes=4), "synth1024x4": lambda: synth_planes(1024, n_planes=4),
              "mixed256x3": lambda: edge_image(256, 256, "mixed", 3), "synth256x4": lambda: synth_planes(256, n_planes=4)}[case]()
    if not os.path.exists(DRIVER):
        subprocess.run(["make", "-C", os.path.dirname(DRIVER)], check=True)
    c, h, w = planes.shape
    with tempfile.TemporaryDirectory() as d:
        fin, fout = os.path.join(d, "in.bin"), os.path.join(d, "out.blobs")
        with open(fin, "wb") as f:
            f.write(struct.pack("<3i", w, h, c)); f.write(np.ascontiguousarray(planes, np.int32).tobytes())
        subprocess.run([DRIVER, fin, fout, "stripes", str(n)], check=True)
        got = parse_blobs(fout)
    bad = [f"bitmap{i}" for i in range(7) if got[f"st_bitmap_{i}"] != got[f"wh_bitmap_{i}"]]
    for p in range(3):
        bad += [f"{k}{p}" for k in ("defs", "nibbles", "nn") if got[f"st_{k}_{p}"] != got[f"wh_{k}_{p}"]]
    info = np.frombuffer(got["st_info"], np.int32)
    if info[:4].tolist() != np.frombuffer(got["wh_bounds"], np.int32).tolist():
        bad.append(f"bounds {info[:4].tolist()}")
    assert not bad, bad
    assert info[4] == min(n, (h + 63) // 64)                 # stripes that own rows
    import torch
    assert info[5] == (info[4] if torch.cuda.device_count() >= info[4] > 1 else 0)   # ranks of the RCCL gather, 0 when the stripes shared a device
