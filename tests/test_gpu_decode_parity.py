"""GPU decode loops (through the C-ABI) vs the CPU oracle: gradient fill, 1-D range fill, reject-mask expansion,
plus the encode -> decode round trip (PSNR vs source stated, GPU decode vs oracle decode must be identical)."""
import numpy as np
import pytest

from oracle.pyoracle import PASSES, OracleDecoder, OracleEncoder, detile, palette_remap
from tests.images import edge_image, synth_planes

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dec():
    from yaik_amd.decoder import HipTileDecoder
    d = HipTileDecoder(0)
    yield d
    d.close()


@pytest.fixture(scope="module")
def hip():
    from yaik_amd.encoder import HipTileEncoder
    e = HipTileEncoder(0)
    yield e
    e.close()


def _oracle_streams(planes):
    enc = OracleEncoder(planes)
    streams = []
    for sx, sy in PASSES:
        cnt, bm, rgb = enc.fitting_quad_smooth(sx, sy)
        streams.append((sx, sy, cnt, bm, palette_remap(rgb, 250)))
    for p in range(3):
        enc.dynamic_tile_compressor(p)
    pix, typ = enc.streams_1d()
    return streams, typ, pix


CASES = [("synth", 256, 256), ("synth", 512, 512), ("mixed", 128, 128), ("mixed", 208, 144), ("noise", 64, 64), ("flat", 64, 64),
         ("smooth", 256, 128), ("twocolor", 128, 128)]


@pytest.mark.parametrize("kind,w,h", CASES)
def test_decode_matches_oracle(dec, oracle_built, kind, w, h):
    planes = synth_planes(w, n_planes=3) if kind == "synth" else edge_image(w, h, kind, 3)
    streams, typ, pix = _oracle_streams(planes)
    od = OracleDecoder(w, h)
    dec.begin(w, h)
    for sx, sy, cnt, bm, rgb in streams:
        if cnt:
            od.gradient(sx, sy, bm, rgb)
            dec.decompress_gradient(sx, sy, bm, rgb)
    assert np.array_equal(dec.planes(), od.planes()), "gradient fill differs"
    assert np.array_equal(dec.tile4x4(), od.tile4x4()), "tile4x4Mask differs"
    od.split_masks()
    od.decode_1d(typ, pix)
    dec.decompress_1d(typ, pix)
    gp, op = dec.planes(), od.planes()
    assert np.array_equal(gp, op), "1-D range fill differs"
    # every pixel is now defined: report PSNR vs the source (reference figure on YAIK-synth v1: 38.8-39.0 dB)
    rec = np.stack([detile(gp[c], w, h) for c in range(3)]).astype(np.int64)
    err = rec - planes[:3]
    mse = float(np.mean(err ** 2))
    psnr = 99.0 if mse == 0 else 10 * np.log10(255.0 ** 2 / mse)
    assert psnr > 30.0, psnr
    # a20 default image builder: interleaved RGB rows at outputImageStride == the oracle's restatement of internal_imageBuilderFunc
    # (pinned against the compiled reference), row padding untouched
    from oracle.pyoracle import image_builder
    img = dec.image(stride=w * 3 + 13, fill=0xA5)
    assert np.array_equal(img, image_builder(gp, w, h, w * 3 + 13))
    want = np.stack([detile(gp[c], w, h) for c in range(3)], axis=-1).reshape(h, w * 3)
    assert np.array_equal(img[:, : w * 3], want) and (img[:, w * 3:] == 0xA5).all()
    alpha = (np.arange(h * w, dtype=np.int64).reshape(h, w) % 251).astype(np.uint8)
    rgba = dec.image(alpha=alpha).reshape(h, w, 4)                      # the documented RGBA layout (include/YAIK.h)
    assert np.array_equal(rgba[..., :3], want.reshape(h, w, 3)) and np.array_equal(rgba[..., 3], alpha)
    ref_rgba = dec.image(alpha=alpha, stride=w * 4 + 20, fill=0xA5, reference_rgba=True)   # the reference's RGBA branch as it executes
    assert np.array_equal(ref_rgba, image_builder(gp, w, h, w * 4 + 20, alpha=alpha))


@pytest.mark.parametrize("size", [256, 1024])
def test_gpu_encode_gpu_decode_roundtrip(hip, dec, oracle_built, size):
    """GPU encoder output (bitmaps + corner streams) fed straight into the GPU decoder == oracle decode of oracle encode."""
    planes = synth_planes(size, n_planes=3)
    hip.set_image(planes)
    hip.encode(3, False, False)
    streams, typ, pix = _oracle_streams(planes)
    od = OracleDecoder(size, size)
    dec.begin(size, size)
    for i, (sx, sy, cnt, bm, rgb) in enumerate(streams):
        gbm = hip.gradient_bitmap(i)
        grgb = hip.gradient_corners(i)
        assert np.array_equal(gbm, bm)
        assert np.array_equal(palette_remap(grgb, 250), rgb)
        if cnt:
            od.gradient(sx, sy, bm, rgb)
            dec.decompress_gradient(sx, sy, gbm, palette_remap(grgb, 250))
    assert np.array_equal(dec.planes(), od.planes())
    # gradient-covered pixels: reference decoder vs source max |err| = 6 on YAIK-synth v1
    cov = np.repeat(np.repeat(hip.coverage(), 4, axis=0), 4, axis=1)
    rec = np.stack([detile(dec.planes()[c], size, size) for c in range(3)]).astype(np.int64)
    assert np.abs(rec - planes[:3])[:, cov].max() <= 6


GOLDEN_RGBA = ["mixed128_rgba", "synth256_rgba"]


@pytest.mark.parametrize("name", GOLDEN_RGBA + ["synth64_rgb", "twocolor64_rgb"])
def test_decode_against_reference_fixtures(dec, name):
    """The decode kernels fed with the REFERENCE's own streams (tests/golden, captured from the compiled reference) must reproduce
    the reference's own outputs: tiled planes, tile4x4Mask, Decompress1BitTiled's mask (a18) and internal_imageBuilderFunc's rows
    (a20) -- no oracle in between."""
    import os
    from tests.blobs import RGB_OUT_PAD, RGBA_OUT_PAD, parse_mip_chunk
    from tests.golden.make_golden import FULL
    ref = dict(np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", name + ".npz")))
    planes = FULL[name]()
    n, h, w = planes.shape
    counts = ref["grad_counts"].view(np.int32)
    dec.begin(w, h)
    for i, (sx, sy) in enumerate(PASSES):
        if counts[i]:
            dec.decompress_gradient(sx, sy, ref[f"grad_bitmap_{i}"], ref[f"grad_rgbdq_{i}"])
    assert dec.planes().tobytes() == ref["dec_planes_grad"].tobytes()
    assert dec.tile4x4().tobytes() == ref["dec_tile4x4"].tobytes()
    dec.decompress_1d(ref["d1_type"], ref["d1_pix"])
    assert dec.planes().tobytes() == ref["dec_planes_full"].tobytes()
    stride = int(ref["dec_rgb_out_info"].view(np.int32)[0])
    assert stride == w * 3 + RGB_OUT_PAD
    assert dec.image(stride=stride, fill=0xA5).tobytes() == ref["dec_rgb_out"].tobytes()
    if n == 4:
        stride_a = int(ref["dec_rgba_out_info"].view(np.int32)[0])
        assert stride_a == w * 4 + RGBA_OUT_PAD
        got = dec.image(alpha=planes[3].astype(np.uint8), stride=stride_a, fill=0xA5, reference_rgba=True)
        assert got.tobytes() == ref["dec_rgba_out"].tobytes()
        bbox, level, bits = parse_mip_chunk(ref["mip_chunk"].tobytes())
        assert level == 4
        assert (ref["dec_mask_bbox"].view(np.int32) == np.array(bbox, np.int32) * 16).all()
        assert dec.decompress_1bit_tiled(bits, int(bbox[2]), int(bbox[3])).tobytes() == ref["dec_mask"].tobytes()


def test_mask_decode(dec, oracle_built):
    from oracle import pyoracle
    rng = np.random.default_rng(3)
    for bw, bh in ((4, 4), (13, 7), (64, 33)):
        bits = rng.integers(0, 256, (bw * bh + 7) // 8, dtype=np.uint8)
        out = dec.decompress_1bit_tiled(bits, bw, bh)
        ref = np.zeros(bw * bh * 32, dtype=np.uint8)
        pyoracle.lib().yko_dec_mask(bits.ctypes.data, bw, bh, ref.ctypes.data)
        assert np.array_equal(out, ref)
        # hand-derived property: every set source bit becomes a 16x16 block of ones = 4 u64 words of all ones
        words = out.view(np.uint64).reshape(bh, 2, bw, 2)
        src = ((bits[np.arange(bw * bh) >> 3] >> (np.arange(bw * bh) & 7)) & 1).reshape(bh, bw).astype(bool)
        assert np.array_equal(words[:, 0, :, 0] == np.uint64(0xFFFFFFFFFFFFFFFF), src)
        assert np.array_equal(words[:, 1, :, 1] == np.uint64(0xFFFFFFFFFFFFFFFF), src)


@pytest.mark.parametrize("size,npl", [(256, 3), (512, 4), (1024, 4)])
def test_decode_from_device_streams_equals_decode_from_host_streams(size, npl):
    """yk_decode_gradient_device / yk_decode_1d_device (streams read where the encoder left them in HBM, corner streams remapped on the way
    in) against the host-stream entry points the YAIK.h boundary uses: planes and tile4x4Mask identical."""
    from oracle.pyoracle import PASSES, palette_remap
    from yaik_amd.decoder import HipTileDecoder
    from yaik_amd.encoder import HipTileEncoder
    planes = synth_planes(size, n_planes=npl)
    enc = HipTileEncoder(0)
    a, b = HipTileDecoder(0), HipTileDecoder(0)
    try:
        enc.set_image(planes)
        if npl == 4:
            enc.mip_prefilter()
        enc.encode(3, False, False)
        counts = enc.gradient_counts()
        a.begin(size, size)
        for i, (sx, sy) in enumerate(PASSES):
            if counts[i]:
                a.decompress_gradient(sx, sy, enc.gradient_bitmap(i), palette_remap(enc.gradient_corners(i), 250))
        pix, typ = enc.dynamic_tile_compressor()
        a.decompress_1d(typ, pix)
        b.begin(size, size)
        b.decode_from_encoder(enc)                                   # the gradient chunks through ONE yk_decode_gradient_all_device call
        assert np.array_equal(a.planes(), b.planes())
        assert np.array_equal(a.tile4x4(), b.tile4x4())
        b.begin(size, size)
        b.decode_from_encoder(enc, per_pass=True)                    # one yk_decode_gradient_device call per chunk
        assert np.array_equal(a.planes(), b.planes())
        assert np.array_equal(a.tile4x4(), b.tile4x4())
    finally:
        a.close(); b.close(); enc.close()


@pytest.mark.parametrize("seed,size,density", [(1, 256, 0.05), (2, 512, 0.3), (3, 384, 0.02), (4, 512, 0.6)])
def test_all_pass_gradient_decode_on_arbitrary_streams(seed, size, density):
    """yk_decode_gradient_all_device against the per-pass host entry point (pinned to the reference by the fixtures above) on streams no encoder
    writes: random bitmaps per pass, so tiles of different passes overlap (the later pass must win, corners are popped once per lattice
    point in pass and scan order), a colour stream that ends early (missing bytes read as 0) and a pass order that is not the file's."""
    import ctypes as C
    import torch
    from oracle.pyoracle import PASSES, palette_remap
    from yaik_amd._lib import lib
    from yaik_amd.decoder import HipTileDecoder
    from yaik_amd.encoder import _chk
    rng = np.random.default_rng(seed)
    order = list(range(7))
    if seed & 1:
        rng.shuffle(order)
    w = h = size
    bitmaps, rgbs = [], []
    for i in order:
        sx, sy = PASSES[i]
        bigX, bigY = (32 if sx == 2 else 64), (32 if sy == 2 else 64)
        nbits = ((w + bigX - 1) // bigX) * ((h + bigY - 1) // bigY) * (bigX >> sx) * (bigY >> sy)
        bits = (rng.random(nbits) < density * (0.2 if i == 0 else 1.0)).astype(np.uint8)
        bitmaps.append(np.packbits(bits, bitorder="little"))
        n = int(bits.sum()) * 12
        rgbs.append(rng.integers(0, 251, size=max(n - (7 if i == order[-1] else 0), 0), dtype=np.uint8))   # the last stream is short
    a, b = HipTileDecoder(0), HipTileDecoder(0)
    try:
        a.begin(w, h)
        for k, i in enumerate(order):
            a.decompress_gradient(PASSES[i][0], PASSES[i][1], bitmaps[k], palette_remap(rgbs[k], 250))
        b.begin(w, h)
        dev_b = [torch.from_numpy(x.copy()).cuda() for x in bitmaps]
        dev_r = [torch.from_numpy(np.concatenate([x, np.zeros(1, np.uint8)])).cuda() for x in rgbs]
        n = 7
        sx = (C.c_int * n)(*[PASSES[i][0] for i in order]); sy = (C.c_int * n)(*[PASSES[i][1] for i in order])
        bm = (C.c_void_p * n)(*[t.data_ptr() for t in dev_b]); nb = (C.c_size_t * n)(*[x.size for x in bitmaps])
        rp = (C.c_void_p * n)(*[t.data_ptr() for t in dev_r]); nr = (C.c_size_t * n)(*[x.size for x in rgbs])
        torch.cuda.synchronize()
        _chk(b._h, lib().yk_decode_gradient_all_device(b._h, n, sx, sy, bm, nb, rp, nr, 250))
        b.synchronize()
        assert np.array_equal(a.planes(), b.planes())
        assert np.array_equal(a.tile4x4(), b.tile4x4())
    finally:
        a.close(); b.close()
