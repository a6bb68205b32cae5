"""One-off long fuzz of the round-2 code (not part of the suite): the fused kernel on tile-regime images, plane-subset passes with random
masks and shapes, the 3-D LUT search and decode on random banks -- all against the CPU oracle.  usage: gpu_fuzz_round2.py <seed0> <seed1>"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import pyoracle
from oracle.pyoracle import PASSES, OracleDecoder, OracleEncoder, palette_remap, yko_compress_f
from tests.lutbank import lut_image, random_bank
from tests.images import edge_image
from tests.parity import compare_encode
from tests.test_gpu_fuzz_parity import _image
from yaik_amd.decoder import HipTileDecoder
from yaik_amd.encoder import HipTileEncoder

pyoracle.build()
e = HipTileEncoder(0)
d = HipTileDecoder(0)
n0, n1 = int(sys.argv[1]), int(sys.argv[2])
bad_total = 0
t0 = time.time()
SHAPES = [(4, 4), (4, 3), (3, 4), (3, 3), (3, 2), (2, 3), (2, 2)]
LUTP = ((4, 3), (3, 4), (3, 3), (3, 2), (2, 3), (2, 2))


def cells(p8):
    h, w = p8.shape
    return p8[: h // 4 * 4: 4, : w // 4 * 4: 4] != 0


for seed in range(n0, n1):
    rng = np.random.default_rng(seed)
    size = (64, 128, 256)[seed % 3]
    # (a) fused kernel
    planes = _image(9000 + seed, size, 4 if seed % 2 else 3)
    for m3 in (False, True):
        bad = compare_encode(planes, e, m3, want_dst=False, check_corners=True)
        if bad:
            bad_total += 1; print("MISMATCH encode seed", seed, "m3", m3, bad[:3], flush=True)
    # (a') the live 1-D path behind the seven RGB passes (common coverage: offsets from the coverage, coder writing straight into the streams)
    ora = OracleEncoder(planes)
    if planes.shape[0] == 4:
        ora.mip_prefilter()
    for sx, sy in PASSES:
        ora.fitting_quad_smooth(sx, sy)
    for p in range(3):
        ora.dynamic_tile_compressor(p)
    pix, typ = ora.streams_1d()
    e.set_image(planes)
    if planes.shape[0] == 4:
        e.mip_prefilter()
    e.encode(3, False, False)
    gp, gt = e.dynamic_tile_compressor()
    if not (np.array_equal(gp, pix) and np.array_equal(gt, typ)):
        bad_total += 1; print("MISMATCH 1-D seed", seed, flush=True)
    # (a'') decode of that frame from the encoder's device-resident streams (all gradient chunks in one call) == pass-by-pass host-stream decode
    hh, ww = planes.shape[1], planes.shape[2]
    if hh % 16 == 0 and ww % 16 == 0:
        d.begin(ww, hh)
        cnts = e.gradient_counts()
        for i, (sx, sy) in enumerate(PASSES):
            if cnts[i]:
                d.decompress_gradient(sx, sy, e.gradient_bitmap(i), palette_remap(e.gradient_corners(i), 250))
        d.decompress_1d(gt, gp)
        ref_planes, ref_mask = d.planes().copy(), d.tile4x4().copy()
        d.begin(ww, hh)
        d.decode_from_encoder(e)
        if not (np.array_equal(d.planes(), ref_planes) and np.array_equal(d.tile4x4(), ref_mask)):
            bad_total += 1; print("MISMATCH device-stream decode seed", seed, flush=True)
    # (b) plane-subset passes, random masks and shapes
    kind = ("planemix", "mixed", "photo")[seed % 3]
    img = edge_image(size + 16 * (seed % 2), size, kind, 3, seed=seed)
    ora = OracleEncoder(img)
    for sx, sy in PASSES:
        ora.fitting_quad_smooth(sx, sy)
    e.set_image(img); e.encode(3, False, False)
    for _ in range(4):
        m = int(rng.integers(1, 8)); sx, sy = SHAPES[int(rng.integers(0, 7))]
        cnt, bm, rgb = ora.fitting_quad_smooth(sx, sy, plane_bit=m)
        g = e.fitting_quad_smooth_planes(m, sx, sy)
        if (g[0], g[1].tobytes(), g[2].tobytes()) != (cnt, bm.tobytes(), rgb.tobytes()):
            bad_total += 1; print("MISMATCH partial seed", seed, m, sx, sy, flush=True)
    for p in range(3):
        if not np.array_equal(e.coverage_plane(p), cells(ora.state("mapSmoothTile", p))):
            bad_total += 1; print("MISMATCH partial coverage seed", seed, p, flush=True)
        ora.dynamic_tile_compressor(p)
    pix, typ = ora.streams_1d()
    gp, gt = e.dynamic_tile_compressor()
    if not (np.array_equal(gp, pix) and np.array_equal(gt, typ)):
        bad_total += 1; print("MISMATCH partial 1-D seed", seed, flush=True)
    # (c) 3-D LUT search + decode, random bank
    pats = random_bank(7000 + seed, int(rng.integers(1, 9)))
    w = h = (128, 144, 256)[seed % 3]
    limg = lut_image(w, h, pats, seed=seed)
    ora = OracleEncoder(limg); od = OracleDecoder(w, h)
    e.lut_clear()
    for p in pats:
        ora.lut_load(p); e.lut_load(p)
    e.set_image(limg); e.encode(3, False, False)
    d.begin(w, h)
    for i, (sx, sy) in enumerate(PASSES):
        ora.fitting_quad_smooth(sx, sy)
        bm, rgb = e.gradient_bitmap(i), e.gradient_corners(i)
        if rgb.size:
            dq = palette_remap(rgb, 250); od.gradient(sx, sy, bm, dq); d.decompress_gradient(sx, sy, bm, dq)
    ora.lut_start(); e.lut_start()
    for sx, sy in LUTP:
        if ora.lut_search(sx, sy) != e.lut_search(sx, sy):
            bad_total += 1; print("MISMATCH lut count seed", seed, sx, sy, flush=True)
    so, sg = ora.lut_streams(), e.lut_streams()
    for k in ("tileType", "color", "idx3", "idx4", "idx5", "idx6") + tuple(f"map{i}" for i in range(6)):
        if not np.array_equal(sg[k], so[k]):
            bad_total += 1; print("MISMATCH lut stream seed", seed, k, flush=True)
    colors = palette_remap(yko_compress_f(sg["color"], 250), 250)
    idx = [(sg[f"idx{b}"].astype(np.uint16) * 3).astype(np.uint8) for b in (3, 4, 5, 6)]
    maps = [sg[f"map{k}"] for k in range(6)]
    lf = ora.lut_file()
    od.lut3d(lf, maps, sg["tileType"], colors, idx)
    d.assign_lut(lf); d.decompress_lut3d(maps, sg["tileType"], colors, idx)
    if not (np.array_equal(d.planes(), od.planes()) and np.array_equal(d.tile4x4().ravel(), od.tile4x4().ravel())):
        bad_total += 1; print("MISMATCH lut decode seed", seed, flush=True)
    e.lut_clear()
    if seed % 10 == 9:
        print(f"seed {seed} done, {time.time() - t0:.0f} s, mismatching cases so far: {bad_total}", flush=True)
print("seeds", n0, n1, "mismatching cases:", bad_total)
