"""Shared comparison: HIP path (through the C-ABI) vs the CPU oracle on the same planes."""
import numpy as np

from oracle.pyoracle import PASSES, OracleEncoder


def cells_from_plane(p8: np.ndarray) -> np.ndarray:
    """w*h 0/255 coverage plane -> [h/4, w/4] bool sampled at the top-left pixel of each 4x4 cell."""
    return p8[::4, ::4] != 0


def compare_encode(planes: np.ndarray, hip, mode3bit_only: bool, want_dst: bool = True, check_corners: bool = False):
    """Runs the whole encode on both sides; returns list of mismatch descriptions (empty = bit-exact)."""
    n, h, w = planes.shape
    bad = []

    def chk(name, a, b):
        a = np.asarray(a).ravel(); b = np.asarray(b).ravel()
        if a.shape != b.shape or not np.array_equal(a, b):
            where = ""
            if a.shape == b.shape:
                idx = np.nonzero(a != b)[0]
                where = f" first@{idx[:4].tolist()} n={idx.size}"
            bad.append(f"{name}: {a.shape} vs {b.shape}{where}")

    ora = OracleEncoder(planes)
    hip.set_image(planes)
    if n == 4:
        mo = ora.mip_prefilter()
        mh = hip.mip_prefilter()
        chk("alpha.has_chunk", [mh["has_chunk"]], [mo["has_chunk"]])
        chk("alpha.bounds", mh["bounds"], mo["bounds"])
        chk("alpha.remaining", [mh["remaining"]], [mo["remaining"]])
        if mo["has_chunk"]:
            chk("alpha.tile_bbox", mh["tile_bbox"], mo["tile_bbox"])
            chk("alpha.bitmap", mh["bitmap"], mo["bitmap"])
    hip.encode(3, mode3bit_only, want_dst)
    counts_o = []
    for i, (sx, sy) in enumerate(PASSES):
        cnt, bm, rgb = ora.fitting_quad_smooth(sx, sy)
        counts_o.append(cnt)
        chk(f"grad.bitmap{i}", hip.gradient_bitmap(i), bm)
        if check_corners:
            chk(f"grad.corners{i}", hip.gradient_corners(i), rgb)
    chk("grad.counts", hip.gradient_counts(), counts_o)
    chk("grad.coverage", hip.coverage(), cells_from_plane(ora.state("smoothMap")))
    for p in range(3):
        defs, nib, nn, dst = ora.dynamic_tile_encode(p, mode3bit_only)
        d2, n2, nn2 = hip.range_streams(p)
        chk(f"range.nNibbles{p}", [nn2], [nn])
        chk(f"range.defs{p}", d2, defs)
        chk(f"range.nibbles{p}", n2, nib)
        if want_dst:
            chk(f"range.dst{p}", hip.range_dst(p), dst)
    return bad


def encoder_for_kernel_version(version: int, device: int = 0):
    """version 2 = the PRODUCT library's fused kernel (yaik_amd/libyaik_hip.so, no hooks involved); 1 = the test suite's independent
    first-generation implementation (tests/csrc/libyaik_v1check.so), registered with the TEST build of the library
    (tests/csrc/libyaik_hip_test.so, include/yaik_hip_test.h) through its cross-check hook."""
    import ctypes as C
    import os
    from yaik_amd._lib import test_lib
    from yaik_amd.encoder import HipTileEncoder
    if version == 2:
        return HipTileEncoder(device)
    here = os.path.dirname(os.path.abspath(__file__))
    path = os.path.join(here, "csrc", "libyaik_v1check.so")
    if not os.path.exists(path):                                           # normally built by __graft_entry__.build() / make -C yaik_amd/csrc
        import subprocess
        subprocess.run(["make", "-C", os.path.join(os.path.dirname(here), "yaik_amd", "csrc")], check=True)
    v1 = C.CDLL(path)
    encoder_for_kernel_version._keep = v1                                  # the library keeps a raw function pointer
    enc = HipTileEncoder(device, hooks=True)
    assert test_lib().yk_set_cross_check_launcher(C.cast(v1.yk_v1_launch, C.c_void_p)) == 0
    assert test_lib().yk_set_kernel_version(enc._h, 1) == 0
    return enc
