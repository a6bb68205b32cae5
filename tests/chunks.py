"""Helpers for the chunk-framing tests: feed raw pass outputs (from the CPU oracle, or from the HIP path) to
yaik_amd/host/entropy_tool and compare the framed stream, chunk by chunk, with the reference's own chunk stream
(blob `chunks_file` of oracle/ref_driver.cpp / tests/golden/*.npz).  ZStd payload bytes are not compared (the reference
vendors zstd 1.3.4, the image ships 1.4.8): headers and DEcompressed payloads are."""
import os
import struct
import subprocess
import tempfile

import numpy as np

from oracle.refrun import parse_blobs

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST = os.path.join(ROOT, "yaik_amd", "host")
TOOL = os.path.join(HOST, "entropy_tool")


def build_tool() -> str:
    subprocess.run(["make", "-C", HOST, "entropy_tool"], check=True, stdout=subprocess.DEVNULL)
    return TOOL


def write_blobs(path: str, blobs: dict) -> None:
    with open(path, "wb") as f:
        for k, v in blobs.items():
            v = bytes(v)
            f.write(struct.pack("<I", len(k))); f.write(k.encode()); f.write(struct.pack("<Q", len(v))); f.write(v)


def oracle_streams(planes: np.ndarray) -> dict:
    """Raw outputs of the passes in the order oracle/ref_driver.cpp runs them, before any entropy coding."""
    from oracle.pyoracle import PASSES, OracleEncoder
    n, h, w = planes.shape
    out = {"meta": np.array([w, h, n], np.int32).tobytes()}
    enc = OracleEncoder(planes)
    if n == 4:
        m = enc.mip_prefilter()
        out["_mip_bitmap"] = m["bitmap"].tobytes()
        out["_mip_tile_bbox"] = m["tile_bbox"].astype(np.int16).tobytes()
        out["_mip_has_chunk"] = bytes([int(m["has_chunk"])])
    for i, (sx, sy) in enumerate(PASSES):
        cnt, bm, rgb = enc.fitting_quad_smooth(sx, sy)
        out[f"grad_bitmap_{i}"] = bm.tobytes()
        out[f"grad_rgbraw_{i}"] = rgb.tobytes()
    out["bounds_post"] = enc.bounds().astype(np.int32).tobytes()
    for m in range(2):
        for p in range(3):
            defs, nib, nn, dst = enc.dynamic_tile_encode(p, bool(m))
            out[f"plnt_defs_{m}_{p}"] = defs.tobytes()
            out[f"plnt_idx_{m}_{p}"] = nib.tobytes()
    for p in range(3):
        enc.dynamic_tile_compressor(p)
    pix, typ = enc.streams_1d()
    out["d1_pix"] = pix.tobytes()
    out["d1_type"] = typ.tobytes()
    return out


def frame(streams: dict, with_file_header: bool = False) -> bytes:
    """entropy_tool write: frames the streams with the product's chunk writer (chunks.cpp + palette.cpp + libzstd)."""
    with tempfile.TemporaryDirectory() as d:
        s = dict(streams)
        if with_file_header:
            s["with_file_header"] = b"\x01"
        write_blobs(os.path.join(d, "s.blobs"), s)
        subprocess.run([TOOL, "write", os.path.join(d, "s.blobs"), os.path.join(d, "o.yaik")], check=True)
        with open(os.path.join(d, "o.yaik"), "rb") as f:
            return f.read()


def parse(stream: bytes, w: int, h: int) -> dict:
    with tempfile.TemporaryDirectory() as d:
        with open(os.path.join(d, "i.bin"), "wb") as f:
            f.write(stream)
        subprocess.run([TOOL, "parse", os.path.join(d, "i.bin"), str(w), str(h), os.path.join(d, "o.blobs")], check=True)
        return parse_blobs(os.path.join(d, "o.blobs"))


def compare_parsed(ref: dict, ours: dict) -> list:
    bad = []
    for k, v in ref.items():
        if k not in ours:
            bad.append(f"missing {k}")
        elif bytes(v) != bytes(ours[k]):
            a, b = np.frombuffer(v, np.uint8), np.frombuffer(ours[k], np.uint8)
            first = int(np.argmax(a[:min(a.size, b.size)] != b[:min(a.size, b.size)])) if min(a.size, b.size) else -1
            bad.append(f"{k}: {a.size} vs {b.size} bytes, first diff @{first}")
    for k in ours:
        if k not in ref:
            bad.append(f"extra {k}")
    return bad
