"""TEST INFRASTRUCTURE ONLY.  Runs oracle/_ref/ref_driver (the unmodified reference, built by
oracle/Makefile) on int32 planes and returns its named output blobs.  Only the golden-vector
generator and CPU tests that are skipped when the binary is absent may call this."""
from __future__ import annotations

import os
import struct
import subprocess
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF_DRIVER = os.path.join(HERE, "_ref", "ref_driver")


def have_ref() -> bool:
    return os.path.exists(REF_DRIVER)


def parse_blobs(path: str) -> dict[str, bytes]:
    out: dict[str, bytes] = {}
    with open(path, "rb") as f:
        data = f.read()
    pos = 0
    while pos < len(data):
        (nl,) = struct.unpack_from("<I", data, pos); pos += 4
        name = data[pos:pos + nl].decode(); pos += nl
        (dl,) = struct.unpack_from("<Q", data, pos); pos += 8
        out[name] = data[pos:pos + dl]; pos += dl
    return out


def run_reference(planes: np.ndarray, partial: bool = False, lut_bank: bytes | None = None) -> dict[str, bytes]:
    planes = np.ascontiguousarray(planes, dtype=np.int32)
    n, h, w = planes.shape
    with tempfile.TemporaryDirectory() as d:
        fin = os.path.join(d, "in.bin")
        fout = os.path.join(d, "out.blobs")
        with open(fin, "wb") as f:
            f.write(struct.pack("<3i", w, h, n))
            f.write(planes.tobytes())
        extra = ["partial"] if partial else []
        if lut_bank is not None:                                      # (f)4: a synthetic 3-D LUT bank in Load3DPattern's file format
            fbank = os.path.join(d, "bank.bin")
            with open(fbank, "wb") as f:
                f.write(lut_bank)
            extra = ["lut3d", fbank]
        subprocess.run([REF_DRIVER, fin, fout] + extra, check=True, stdout=subprocess.DEVNULL)
        return parse_blobs(fout)
