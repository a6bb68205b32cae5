"""Recomputes, with the CPU oracle, the named blobs that oracle/ref_driver.cpp dumps from the unmodified reference,
so that golden fixtures (tests/golden/*.npz, hashes.json) and live reference runs can be compared name by name."""
import numpy as np

from oracle.pyoracle import (PASSES, OracleDecoder, OracleEncoder, dec_mask, image_builder, palette_decompress)

# blobs whose bytes depend on uninitialised reference memory or that the fixtures do not keep; dec_mask_info carries the error
# code CheckInBound2D's missing `return true` (decoder/YAIK_Alpha.cpp:12-23, UB) happens to produce
SKIP = {"meta", "mip_chunk", "dec_mapRGB", "chunks_file", "stage_seconds", "dec_mask_info"}      # chunks_file: see tests/test_host_chunks.py
RGB_OUT_PAD, RGBA_OUT_PAD = 13, 20                     # row padding ref_driver.cpp gives the two outputImageStride values


def oracle_blobs(planes: np.ndarray) -> dict:
    n, h, w = planes.shape
    out = {}
    enc = OracleEncoder(planes)
    if n == 4:
        mip = enc.mip_prefilter()
        out["mip_bounds"] = np.array(list(mip["bounds"]) + [16, mip["remaining"]], dtype=np.int32).tobytes()
        out["mip_mask"] = enc.state("mipmapMask").tobytes()
        out["_mip_bitmap"] = mip["bitmap"].tobytes()
        out["_mip_tile_bbox"] = mip["tile_bbox"].astype(np.int16).tobytes()
        out["_mip_has_chunk"] = bytes([int(mip["has_chunk"])])
    counts, streams = [], []
    for i, (sx, sy) in enumerate(PASSES):
        cnt, bm, rgb = enc.fitting_quad_smooth(sx, sy)
        counts.append(cnt)
        out[f"grad_bitmap_{i}"] = bm.tobytes()
        if cnt:
            pal = enc.palette_compress(rgb)                       # same process-global code table as the reference
            dq = palette_decompress(pal, rgb.size, 250)
            out[f"grad_palette_{i}"] = pal.tobytes()
        else:
            dq = np.zeros(0, np.uint8)
        out[f"grad_rgbdq_{i}"] = dq.tobytes()
        streams.append((sx, sy, cnt, bm, dq))
    out["grad_counts"] = np.array(counts, dtype=np.int32).tobytes()
    out["smoothMap"] = enc.state("smoothMap").tobytes()
    out["mipmapMask_post"] = enc.state("mipmapMask").tobytes()
    out["bounds_post"] = enc.bounds().astype(np.int32).tobytes()
    for p in range(3):
        out[f"mapSmoothTile_{p}"] = enc.state("mapSmoothTile", p).tobytes()
        out[f"preview_{p}"] = enc.state("preview", p).astype(np.int16).tobytes()
    for m in range(2):
        for p in range(3):
            defs, nib, nn, dst = enc.dynamic_tile_encode(p, bool(m))
            out[f"plnt_defs_{m}_{p}"] = defs.tobytes()
            out[f"plnt_idx_{m}_{p}"] = nib.tobytes()
            out[f"plnt_dst_{m}_{p}"] = dst.astype(np.int16).tobytes()
    ends = []
    for p in range(3):
        _, dbg = enc.dynamic_tile_compressor(p)
        out[f"d1_out_{p}"] = dbg.astype(np.int16).tobytes()
        pix, typ = enc.streams_1d()
        ends.append((pix.size, typ.size))
    pix, typ = enc.streams_1d()
    out["d1_pix"] = pix.tobytes()
    out["d1_type"] = typ.tobytes()
    out["d1_ends"] = np.array([e[0] for e in ends] + [e[1] for e in ends], dtype=np.int32).tobytes()
    if w % 16 == 0 and h % 16 == 0:
        dec = OracleDecoder(w, h)
        for sx, sy, cnt, bm, dq in streams:
            if cnt:
                dec.gradient(sx, sy, bm, dq)
        out["dec_planes_grad"] = dec.planes().tobytes()
        out["dec_tile4x4"] = dec.tile4x4().tobytes()
        out["dec_mapRGBMask"] = dec.map_rgb_mask().tobytes()
        dec.split_masks()
        tp, pp = dec.decode_1d(typ, pix)
        out["dec_1d_consumed"] = np.array([tp, pp], dtype=np.int32).tobytes()
        out["dec_planes_full"] = dec.planes().tobytes()
        # a20 default image builder on the decoded planes (RGB at a padded stride; RGBA as the reference executes it)
        out["dec_rgb_out_info"] = np.array([w * 3 + RGB_OUT_PAD, 3], dtype=np.int32).tobytes()
        out["dec_rgb_out"] = image_builder(dec.planes(), w, h, w * 3 + RGB_OUT_PAD).tobytes()
        if n == 4:
            out["dec_rgba_out_info"] = np.array([w * 4 + RGBA_OUT_PAD, 4], dtype=np.int32).tobytes()
            out["dec_rgba_out"] = image_builder(dec.planes(), w, h, w * 4 + RGBA_OUT_PAD, alpha=planes[3].astype(np.uint8)).tobytes()
        # a18 reject-mask expansion of the 'MIPM' payload
        if n == 4 and mip["has_chunk"]:
            bx, by, bw, bh = (int(v) for v in mip["tile_bbox"])
            out["dec_mask_bbox"] = np.array([bx * 16, by * 16, bw * 16, bh * 16], dtype=np.int32).tobytes()
            out["dec_mask"] = dec_mask(mip["bitmap"], bw, bh).tobytes()
    return out


PP_MASKS = (5, 3, 6, 1, 2, 4)          # RB, RG, GB, R, G, B: the order of Convert()'s (disabled) 4x4 passes, EncoderContext.cpp:9261-9415
PP_KEEP = ("pp_", "d1_pix", "d1_type", "d1_ends")


def oracle_partial_blobs(planes: np.ndarray) -> dict:
    """The blobs of `ref_driver <in> <out> partial`: the seven RGB passes, then the six partial-plane 4x4 passes, then the 1-D compressor
    on the per-plane coverage, then DecompressGradient4x4(planeBit) as the reference executes it (consistentMarks = 0)."""
    n, h, w = planes.shape
    out = {}
    enc = OracleEncoder(planes)
    if n == 4:
        enc.mip_prefilter()
    streams = []
    for sx, sy in PASSES:
        cnt, bm, rgb = enc.fitting_quad_smooth(sx, sy)
        dq = palette_decompress(enc.palette_compress(rgb), rgb.size, 250) if cnt else np.zeros(0, np.uint8)
        streams.append((sx, sy, cnt, bm, dq))
    pp, counts = [], []
    for i, m in enumerate(PP_MASKS):
        cnt, bm, rgb = enc.fitting_quad_smooth(2, 2, plane_bit=m)
        counts.append(cnt)
        out[f"pp_bitmap_{i}"] = bm.tobytes()
        out[f"_pp_rgb_{i}"] = rgb.tobytes()
        if cnt:
            pal = enc.palette_compress(rgb)
            dq = palette_decompress(pal, rgb.size, 250)
            out[f"pp_palette_{i}"] = pal.tobytes()
            out[f"pp_header_{i}"] = np.array([m, (2 << 3) | 2], dtype=np.int32).tobytes()
        else:
            dq = np.zeros(0, np.uint8)
        out[f"pp_rgbdq_{i}"] = dq.tobytes()
        pp.append((m, cnt, bm, dq))
    out["pp_counts"] = np.array(counts, dtype=np.int32).tobytes()
    out["pp_smoothMap"] = enc.state("smoothMap").tobytes()
    for p in range(3):
        out[f"pp_mapSmoothTile_{p}"] = enc.state("mapSmoothTile", p).tobytes()
    ends = []
    for p in range(3):
        enc.dynamic_tile_compressor(p)
        pix, typ = enc.streams_1d()
        ends.append((pix.size, typ.size))
    out["d1_pix"] = pix.tobytes()
    out["d1_type"] = typ.tobytes()
    out["d1_ends"] = np.array([e[0] for e in ends] + [e[1] for e in ends], dtype=np.int32).tobytes()
    if w % 16 == 0 and h % 16 == 0:
        dec = OracleDecoder(w, h)
        for sx, sy, cnt, bm, dq in streams:
            if cnt:
                dec.gradient(sx, sy, bm, dq)
        dec.split_masks()
        for m, cnt, bm, dq in pp:
            if cnt:
                dec.gradient_planes(m, bm, dq, consistent_marks=False)
        out["pp_dec_planes_grad"] = dec.planes().tobytes()
        out["pp_dec_tile4x4"] = dec.tile4x4(True).tobytes()
        out["pp_dec_mapRGBMask"] = dec.map_rgb_mask(True).tobytes()
    return out


def compare_partial(ref: dict, ours: dict) -> list:
    bad = []
    for k, v in ref.items():
        if not k.startswith(PP_KEEP):
            continue
        if k not in ours:
            bad.append(f"missing {k}")
        elif bytes(v) != ours[k]:
            bad.append(f"{k}: {len(v)} vs {len(ours[k])} bytes")
    return bad


LUT_PASSES = ((4, 3), (3, 4), (3, 3), (3, 2), (2, 3), (2, 2))          # Correlation3DSearch calls of Convert(), EncoderContext.cpp:9144-9199
LUT_KEEP = ("lut_counts", "lut_tileType", "lut_color", "lut_idx", "lut_map", "lut_mapSmoothTile", "lut_zin", "lut_file", "lut_dec_", "d1_pix", "d1_type",
            "d1_ends", "dec_1d_consumed", "dec_planes_full")


def oracle_lut_blobs(planes: np.ndarray, patterns, tables: bool = True) -> dict:
    """The blobs of `ref_driver <in> <out> lut3d <bank>`: seven RGB passes, the bank loaded, six 3-D LUT search passes, the streams as
    EndCorrelationSearch hands them to ZStd (lut_zin_*: maps, tile types, CompressF'd colours, indices x 3), then the 1-D compressor."""
    from oracle.pyoracle import palette_remap, yko_compress_f
    n, h, w = planes.shape
    out = {}
    enc = OracleEncoder(planes)
    if n == 4:
        enc.mip_prefilter()
    grad = []
    for sx, sy in PASSES:
        cnt, bm, rgb = enc.fitting_quad_smooth(sx, sy)
        grad.append((sx, sy, cnt, bm, palette_decompress(enc.palette_compress(rgb), rgb.size, 250) if cnt else np.zeros(0, np.uint8)))
    for k, p in enumerate(patterns):
        enc.lut_load(p)
        if tables:
            fac, dist, pos = enc.lut_tables(k)
            out[f"lut_factors_{k}"] = np.concatenate([fac[i, c, :nn] for i, nn in enumerate((64, 32, 16, 8)) for c in range(3)]).tobytes()
            out[f"lut_distanceField_{k}"] = dist.tobytes()
            out[f"lut_positions_{k}"] = pos.tobytes()
    enc.lut_start()
    counts = []
    for sx, sy in LUT_PASSES:
        enc.lut_search(sx, sy)
        s = enc.lut_streams()
        counts.append([s["tileType"].size, s["color"].size, s["idx3"].size, s["idx4"].size, s["idx5"].size, s["idx6"].size])
    out["lut_counts"] = np.array(counts, dtype=np.int32).tobytes()
    out["lut_tileType"] = s["tileType"].tobytes()
    out["lut_color"] = s["color"].tobytes()
    for bits in (3, 4, 5, 6):
        out[f"lut_idx{bits}"] = s[f"idx{bits}"].tobytes()
    for k in range(6):
        out[f"lut_map_{k}"] = s[f"map{k}"].tobytes()
    for p in range(3):
        out[f"lut_mapSmoothTile_{p}"] = enc.state("mapSmoothTile", p).tobytes()
    out["_lut_preview"] = s["preview"].tobytes()
    # what EndCorrelationSearch compresses, in call order (:7397-7586): 6 maps, tile types, colours after CompressF(., 250), 3/4/5/6-bit x 3
    zin = [s[f"map{k}"].tobytes() for k in range(6)]
    if s["tileType"].size:
        zin.append(s["tileType"].tobytes())
    if s["color"].size:
        zin.append(yko_compress_f(s["color"], 250).tobytes())
    for bits in (3, 4, 5, 6):
        if s[f"idx{bits}"].size:
            zin.append((s[f"idx{bits}"].astype(np.uint16) * 3).astype(np.uint8).tobytes())
    for k, z in enumerate(zin):
        out[f"lut_zin_{k}"] = z
    out["lut_file"] = enc.lut_file().tobytes()
    ends = []
    for p in range(3):
        enc.dynamic_tile_compressor(p)
        pix, typ = enc.streams_1d()
        ends.append((pix.size, typ.size))
    out["d1_pix"] = pix.tobytes()
    out["d1_type"] = typ.tobytes()
    out["d1_ends"] = np.array([e[0] for e in ends] + [e[1] for e in ends], dtype=np.int32).tobytes()
    if w % 16 == 0 and h % 16 == 0:
        # decode: gradient chunks, the '3DTL' chunk (Tile3D_* on the single-plane mask), mask split, '1DTL'
        dec = OracleDecoder(w, h)
        for sx, sy, cnt, bm, dq in grad:
            if cnt:
                dec.gradient(sx, sy, bm, dq)
        used = dec.lut3d(enc.lut_file(), [s[f"map{k}"] for k in range(6)], s["tileType"], palette_remap(yko_compress_f(s["color"], 250), 250),
                         [(s[f"idx{b}"].astype(np.uint16) * 3).astype(np.uint8) for b in (3, 4, 5, 6)])
        out["lut_dec_consumed"] = used.astype(np.int32).tobytes()
        out["lut_dec_planes"] = dec.planes().tobytes()
        out["lut_dec_tile4x4"] = dec.tile4x4().tobytes()
        dec.split_masks()
        tp, pp = dec.decode_1d(typ, pix)
        out["dec_1d_consumed"] = np.array([tp, pp], dtype=np.int32).tobytes()
        out["dec_planes_full"] = dec.planes().tobytes()
    return out


def compare_lut(ref: dict, ours: dict, tables: bool = True) -> list:
    bad = []
    keep = LUT_KEEP + (("lut_factors", "lut_distanceField", "lut_positions") if tables else ())
    for k, v in ref.items():
        if not k.startswith(keep):
            continue
        if k not in ours:
            bad.append(f"missing {k}")
        elif bytes(v) != ours[k]:
            bad.append(f"{k}: {len(v)} vs {len(ours[k])} bytes")
    return bad


def parse_mip_chunk(chunk: bytes):
    """'MIPM' chunk as written by MipPrefilter (EncoderContext.cpp:1367-1396): HeaderBase(8) + MipmapHeader(16) + bitmap.
    MipmapHeader.streamSize is never initialised by the reference, so only bbox / level / bitmap are comparable."""
    if not chunk:
        return None
    bbox = np.frombuffer(chunk[8:16], np.int16).copy()
    level = chunk[21]
    nbytes = (int(bbox[2]) * int(bbox[3]) + 7) // 8
    return bbox, level, np.frombuffer(chunk[24:24 + nbytes], np.uint8).copy()


def compare_with_reference(ref: dict, ours: dict, decode_ok: bool = True) -> list:
    bad = []
    for k, v in ref.items():
        if k in SKIP or (not decode_ok and k.startswith("dec_")):
            continue
        if k not in ours:
            bad.append(f"missing {k}")
        elif bytes(v) != ours[k]:
            bad.append(f"{k}: {len(v)} vs {len(ours[k])} bytes")
    if "mip_chunk" in ref:
        parsed = parse_mip_chunk(bytes(ref["mip_chunk"]))
        has = ours["_mip_has_chunk"] == b"\x01"
        if (parsed is not None) != has:
            bad.append("mip chunk presence")
        elif parsed is not None:
            bbox, level, bm = parsed
            if bbox.tobytes() != ours["_mip_tile_bbox"] or level != 4 or bm.tobytes() != ours["_mip_bitmap"]:
                bad.append("mip chunk content")
    return bad
