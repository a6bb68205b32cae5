"""SURVEY §8(f) 1-3 on the GPU: the C++ drop-in writes the reference's chunk stream ('MIPM', 'GTIL' x7, 'PLNT', '1DTL') from the
HIP kernels' outputs, and the YAIK_* decoder API reads a .yaik stream back through the HIP decode kernels.
Compared with (i) the oracle's streams framed by the same entropy stage, (ii) the committed reference chunk streams
(tests/golden), (iii) the oracle's decode of the same streams."""
import os
import struct
import subprocess
import tempfile

import numpy as np
import pytest

from oracle.refrun import parse_blobs
from tests import chunks
from tests.golden.make_golden import FULL
from tests.images import edge_image, synth_planes

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST = os.path.join(ROOT, "yaik_amd", "host")
DRIVER = os.path.join(HOST, "host_driver")
GOLDEN = os.path.join(ROOT, "tests", "golden")


@pytest.fixture(scope="module")
def built(oracle_built):
    subprocess.run(["make", "-C", HOST], check=True, stdout=subprocess.DEVNULL)
    return True


def _run(planes, mode3=False):
    n, h, w = planes.shape
    with tempfile.TemporaryDirectory() as d:
        fin, fout, fy = os.path.join(d, "in.bin"), os.path.join(d, "out.blobs"), os.path.join(d, "out.yaik")
        with open(fin, "wb") as f:
            f.write(struct.pack("<3i", w, h, n)); f.write(np.ascontiguousarray(planes, np.int32).tobytes())
        subprocess.run([DRIVER, fin, fout, "1" if mode3 else "0", fy], check=True, stdout=subprocess.DEVNULL)
        with open(fy, "rb") as f:
            return parse_blobs(fout), f.read()


def _select(parsed: dict, keep) -> dict:
    """re-number the chunks of a parsed stream after filtering by tag"""
    n = int(np.frombuffer(parsed["chunk_count_terminated"], np.int32)[0])
    out, j = {}, 0
    for i in range(n):
        if not keep(i, parsed[f"c{i}_tag"]):
            continue
        for k, v in parsed.items():
            if k.startswith(f"c{i}_"):
                out[f"c{j}_" + k[len(f"c{i}_"):]] = v
        j += 1
    return out


@pytest.mark.parametrize("name", sorted(FULL))
def test_gpu_chunk_stream_matches_golden_reference_stream(built, name):
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    planes = FULL[name]()
    n, h, w = planes.shape
    if w % 16 or h % 16:
        pytest.skip("the decode half of the driver needs multiples of 16")
    ref = chunks.parse(z["chunks_file"].tobytes(), w, h)
    for mode3 in (False, True):
        got, _ = _run(planes, mode3)
        ours = chunks.parse(got["chunks_file"], w, h)
        # the reference driver runs DynamicTileEncode for both start modes (6 'PLNT' chunks); the C++ driver for one (3)
        plnt = [i for i in range(int(np.frombuffer(ref["chunk_count_terminated"], np.int32)[0])) if ref[f"c{i}_tag"] == b"PLNT"]
        drop = set(plnt[:3] if mode3 else plnt[3:])
        want = _select(ref, lambda i, tag: i not in drop)
        have = _select(ours, lambda i, tag: True)
        bad = chunks.compare_parsed(want, have)
        assert not bad, (mode3, bad)


@pytest.mark.parametrize("case", ["synth256x3", "synth256x4", "mixed128x4", "smooth128x3", "synth1024x3"])
def test_yaik_stream_round_trip_through_decoder_api(built, case):
    planes = {"synth256x3": lambda: synth_planes(256, n_planes=3), "synth256x4": lambda: synth_planes(256, n_planes=4),
              "mixed128x4": lambda: edge_image(128, 128, "mixed", 4), "smooth128x3": lambda: edge_image(128, 128, "smooth", 3),
              "synth1024x3": lambda: synth_planes(1024, n_planes=3)}[case]()
    n, h, w = planes.shape
    got, stream = _run(planes)
    # (o) the threaded entropy stage (ConvertHotPathBegin / Finish: PaletteCompressor in pass order, ZStd streams on worker threads) writes
    #     the same file, byte for byte
    assert got["yaik_file"] == stream and got["yaik_file_parallel"] == got["yaik_file"]
    # (i) the stream is what the oracle's passes give when framed by the same entropy stage: header, [MIPM], GTILs, 1DTL, end
    s = chunks.oracle_streams(planes)
    for k in list(s):
        if k.startswith("plnt_"):
            del s[k]
    want = chunks.parse(chunks.frame(s, with_file_header=True), w, h)
    have = chunks.parse(stream, w, h)
    bad = chunks.compare_parsed(want, have)
    assert not bad, bad
    # (ii) decoded image == the oracle's decode of the same streams (Decompress* restated, pinned against the reference)
    from oracle.pyoracle import PASSES, OracleDecoder, OracleEncoder, detile, palette_decompress
    enc = OracleEncoder(planes)
    if n == 4:
        enc.mip_prefilter()
    dec = OracleDecoder(w, h)
    for sx, sy in PASSES:
        cnt, bm, rgb = enc.fitting_quad_smooth(sx, sy)
        if cnt and rgb.size:
            dec.gradient(sx, sy, bm, palette_decompress(enc.palette_compress(rgb), rgb.size, 250))
    for p in range(3):
        enc.dynamic_tile_compressor(p)
    pix, typ = enc.streams_1d()
    dec.split_masks()
    dec.decode_1d(typ, pix)
    tiled = dec.planes()
    assert got["yaik_planes_tiled"] == tiled.tobytes()
    rgb_img = np.stack([detile(tiled[i], w, h) for i in range(3)], axis=-1)
    dims = np.frombuffer(got["yaik_dims"], np.int32).tolist()
    assert dims == [w, h, 4 if n == 4 else 3]
    # The stream holds no 'ALPM' chunk, so like the reference's pCtx->alphaChannel the alpha plane is NULL and the default builder
    # writes RGB triples (YAIK_API.cpp:1316, YAIK_DefaultCallback.cpp:63) even under an RGBA header; the driver gave rows of
    # w * bpp bytes, whose tail must stay untouched (zero-initialised by the driver).
    rows = np.frombuffer(got["yaik_image"], np.uint8).reshape(h, w * dims[2])
    img = rows[:, : w * 3].reshape(h, w, 3)
    assert np.array_equal(img, rgb_img)
    assert not rows[:, w * 3:].any()
    # PSNR of the round trip against the source on the pixels the stream defines (everything: gradient + 1-D range)
    err = img.astype(np.float64) - np.stack([planes[i] for i in range(3)], axis=-1)
    psnr = 10 * np.log10(255.0 ** 2 / max(np.mean(err ** 2), 1e-12))
    assert psnr > 30.0, psnr
    # (iii) API error convention: Decode without Pre fails, the sticky code reads once (YAIK_DECIMG_INVALIDCTX = 9) and resets
    assert np.frombuffer(got["yaik_error_convention"], np.int32).tolist() == [0, 9, 0]
    # malformed streams: wrong magic -> YAIK_INVALID_HEADER (7); unknown tag / chunk past the end -> YAIK_INVALID_TAG_ID (20)
    assert np.frombuffer(got["yaik_malformed"], np.int32).tolist() == [0, 7, 0, 20, 0, 20]
