"""Generates the golden vectors under tests/golden/ from the UNMODIFIED reference (oracle/_ref/ref_driver, built by
oracle/Makefile from the sources under /root/reference).  The reference ships no fixtures of its own, so
these captured outputs are the parity anchor that travels with the repo; the reference itself never does.

    python tests/golden/make_golden.py          # needs oracle/_ref/ref_driver (i.e. a checkout with /root/reference present)

Fixtures are data only: inputs are regenerated from seeds (yaik_amd.synth / tests.images), outputs are the blobs the
reference produced: small images keep every blob (npz, compressed), larger ones keep a SHA-256 per blob.
"""
import hashlib
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from oracle.refrun import have_ref, run_reference  # noqa: E402
from tests.blobs import LUT_KEEP, PP_KEEP  # noqa: E402
from tests.lutbank import bank16, bank_bytes, bank_patterns, lut_image, lut_image_rgba, random_bank  # noqa: E402
from tests.images import edge_image, lineart_image, natural_photo, synth_planes  # noqa: E402

PHOTO_SRC = "/opt/conda/lib/python3.9/site-packages/skimage/data/astronaut.png"       # present in the build image; never read by tests
PHOTO_INPUT = os.path.join(HERE, "photo_astronaut256_input.npz")

FULL = {   # name -> planes factory; every reference blob is stored
    "synth64_rgb": lambda: synth_planes(64, n_planes=3),
    "mixed128_rgba": lambda: edge_image(128, 128, "mixed", 4),
    "twocolor64_rgb": lambda: edge_image(64, 64, "twocolor", 3),
    "ramp72x40_rgb": lambda: edge_image(72, 40, "ramp", 3),
    "synth256_rgba": lambda: synth_planes(256, n_planes=4),
}
HASHED = {  # name -> planes factory; SHA-256 per blob
    "synth512_rgba": lambda: synth_planes(512, n_planes=4),
    "synth1024_rgba": lambda: synth_planes(1024, n_planes=4),
    "mixed208x144_rgb": lambda: edge_image(208, 144, "mixed", 3),
    "lineart1024_rgba": lambda: lineart_image(1024, 4),          # procedurally drawn flat-colour illustration, anti-aliased edges
    "photo_astronaut256_rgb": natural_photo,                     # natural photograph (committed crop, see write_photo_input)
}
PARTIAL = {  # name -> planes factory; `ref_driver ... partial` (six partial-plane 4x4 passes after the RGB passes): pp_* + 1-D blobs kept
    "pp_planemix128_rgb": lambda: edge_image(128, 128, "planemix", 3),
    "pp_planemix144x80_rgb": lambda: edge_image(144, 80, "planemix", 3),
    "pp_planemix256_rgba": lambda: edge_image(256, 256, "planemix", 4),
    "pp_mixed128_rgba": lambda: edge_image(128, 128, "mixed", 4),
}
LUT3D = {  # name -> (planes, patterns); `ref_driver ... lut3d <bank>` with the synthetic bank of tests/lutbank.py: streams + maps + 1-D blobs
    "lut_lutmix128_rgb": lambda: (lut_image(128, 128, seed=3), bank_patterns()),
    "lut_lutmix192x144_rgb": lambda: (lut_image(192, 144, seed=7), bank_patterns()),
    "lut_lutmix256_rgba_bank16": lambda: (lut_image_rgba(256, 256, bank16(), seed=9), bank16()),     # 16 patterns, alpha plane in front
    "lut_random_bank160x112_rgb": lambda: (lut_image(160, 112, random_bank(311, 9), seed=21), random_bank(311, 9)),   # unordered / duplicated / clustered points, partial 64x64 blocks
    "lut_full_bank192x160_rgb": lambda: (lut_image(192, 160, random_bank(312, 64)[:8], seed=22), random_bank(312, 64)),   # a full bank: 64 patterns
}
# blobs that only serve debugging or are derivable from the others are dropped to keep the fixtures small
DROP_PREFIX = ("preview_", "d1_out_", "mapSmoothTile_")
# not hashed: wall-clock timings; chunks whose headers carry uninitialised reference memory (MipmapHeader.streamSize, EncoderContext.cpp:1367-1396)
UNSTABLE = ("chunks_file", "stage_seconds", "mip_chunk")


def write_photo_input():
    """(Re)creates the committed input crop from the PNG when the build image has it; otherwise the committed file is used as is."""
    if not os.path.exists(PHOTO_SRC):
        return
    from PIL import Image
    im = np.asarray(Image.open(PHOTO_SRC).convert("RGB"))
    np.savez_compressed(PHOTO_INPUT, rgb=np.ascontiguousarray(im[32:288, 96:352]))


def main():
    only = sys.argv[1] if len(sys.argv) > 1 else None                # `make_golden.py <lut fixture name>`: that fixture alone
    if only in LUT3D:
        planes, pats = LUT3D[only]()
        blobs = run_reference(planes, lut_bank=bank_bytes(pats))
        keep = {k: np.frombuffer(v, dtype=np.uint8) for k, v in blobs.items() if k.startswith(LUT_KEEP)}
        np.savez_compressed(os.path.join(HERE, only + ".npz"), **keep)
        print(only, sum(v.size for v in keep.values()), "bytes raw")
        return 0
    write_photo_input()
    if not have_ref():
        print("oracle/_ref/ref_driver is missing: run `make -C oracle` on a machine that has /root/reference", file=sys.stderr)
        return 1
    for name, mk in FULL.items():
        blobs = run_reference(mk())
        keep = {k: np.frombuffer(v, dtype=np.uint8) for k, v in blobs.items() if not k.startswith(DROP_PREFIX)}
        np.savez_compressed(os.path.join(HERE, name + ".npz"), **keep)
        print(name, sum(v.size for v in keep.values()), "bytes raw")
    for name, mk in PARTIAL.items():
        blobs = run_reference(mk(), partial=True)
        keep = {k: np.frombuffer(v, dtype=np.uint8) for k, v in blobs.items() if k.startswith(PP_KEEP)}
        np.savez_compressed(os.path.join(HERE, name + ".npz"), **keep)
        print(name, sum(v.size for v in keep.values()), "bytes raw")
    for name, mk in LUT3D.items():
        planes, pats = mk()
        blobs = run_reference(planes, lut_bank=bank_bytes(pats))
        keep = {k: np.frombuffer(v, dtype=np.uint8) for k, v in blobs.items() if k.startswith(LUT_KEEP)}
        np.savez_compressed(os.path.join(HERE, name + ".npz"), **keep)
        print(name, sum(v.size for v in keep.values()), "bytes raw")
    hashes = {}
    for name, mk in HASHED.items():
        blobs = run_reference(mk())
        # chunks_file holds ZStd payloads and uninitialised header bytes: pinned in parsed form by tests/test_host_chunks.py, not by hash
        hashes[name] = {k: hashlib.sha256(v).hexdigest() for k, v in blobs.items() if not k.startswith(DROP_PREFIX) and k not in UNSTABLE}
        hashes[name]["grad_counts_values"] = np.frombuffer(blobs["grad_counts"], np.int32).tolist()
    with open(os.path.join(HERE, "hashes.json"), "w") as f:
        json.dump(hashes, f, indent=1, sort_keys=True)
    return 0


if __name__ == "__main__":
    sys.exit(main())
