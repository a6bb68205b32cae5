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
    names = ["grad_counts", "d1_pix", "d1_type"] + [f"plnt_{k}_{m}_{p}" for k in ("defs", "idx", "dst") for p in range(3)]
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
