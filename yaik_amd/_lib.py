"""Loader of the in-tree HIP shared library (yaik_amd/libyaik_hip.so) and its ctypes signatures.

The product path has no CPU fallback: if the library is missing, or no HIP device is usable,
callers get a loud YaikError.  The symbol list below is checked against include/yaik_hip.h by
tests/test_host_logic.py (no GPU needed to load the library and resolve symbols).
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
# YK_LIB=/path/to/variant.so selects another build of the same C-ABI (same-box A/B runs, instrumented builds): the product
# library in the tree is never overwritten by an experiment (tools/ab.sh).
LIB_PATH = os.environ.get("YK_LIB") or os.path.join(HERE, "libyaik_hip.so")
CSRC = os.path.join(HERE, "csrc")


class YaikError(RuntimeError):
    pass


def build(force: bool = False) -> str:
    """hipcc --offload-arch=gfx950 build of the library (cross-compiles without a GPU)."""
    if force and os.path.exists(LIB_PATH):
        os.remove(LIB_PATH)
    subprocess.run(["make", "-C", CSRC], check=True, stdout=subprocess.DEVNULL)
    return LIB_PATH


vp, ip = C.c_void_p, C.POINTER(C.c_int)
sz, szp = C.c_size_t, C.POINTER(C.c_size_t)
i32p = C.POINTER(C.c_int32)

# name -> (restype, argtypes); every exported symbol of include/yaik_hip.h
SIGNATURES = {
    "yk_create": (C.c_int, [C.c_int, C.POINTER(vp)]),
    "yk_destroy": (None, [vp]),
    "yk_last_error": (C.c_char_p, [vp]),
    "yk_set_stream": (C.c_int, [vp, vp]),
    "yk_synchronize": (C.c_int, [vp]),
    "yk_device_count": (C.c_int, []),
    "yk_set_image": (C.c_int, [vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]),
    "yk_upload_planes": (C.c_int, [vp, C.POINTER(vp), C.c_int]),
    "yk_bind_device_planes": (C.c_int, [vp, C.POINTER(vp), C.c_int]),
    "yk_validate_planes": (C.c_int, [vp, szp]),
    "yk_alpha_reject": (C.c_int, [vp]),
    "yk_get_stripe_bbox": (C.c_int, [vp, vp]),
    "yk_alpha_finish": (C.c_int, [vp, vp]),
    "yk_alpha_result": (C.c_int, [vp, vp, ip, ip, vp]),
    "yk_alpha_bitmap": (C.c_int, [vp, vp, sz, szp]),
    "yk_encode_tiles": (C.c_int, [vp, C.c_int, C.c_int, C.c_int]),
    "yk_encode_frame": (C.c_int, [vp, C.c_int, C.c_int]),
    "yk_set_batch": (C.c_int, [vp, C.c_int]),
    "yk_bind_device_batch": (C.c_int, [vp, C.POINTER(vp), C.c_int, sz]),
    "yk_encode_batch": (C.c_int, [vp, C.c_int, C.c_int]),
    "yk_select_frame": (C.c_int, [vp, C.c_int]),
    "yk_order_fused_after": (C.c_int, [vp, vp]),
    "yk_gradient_bitmap_bytes": (sz, [vp, C.c_int]),
    "yk_gradient_bitmap": (C.c_int, [vp, C.c_int, vp, sz]),
    "yk_gradient_bitmap_device": (vp, [vp, C.c_int]),
    "yk_gradient_counts": (C.c_int, [vp, vp]),
    "yk_coverage": (C.c_int, [vp, vp, sz]),
    "yk_gradient_corners": (C.c_int, [vp, C.c_int, vp, sz, szp]),
    "yk_gradient_partial_pass": (C.c_int, [vp, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int)]),
    "yk_gradient_preview": (C.c_int, [vp, C.c_int, vp, sz]),
    "yk_partial_bitmap": (C.c_int, [vp, vp, sz, szp]),
    "yk_partial_corners": (C.c_int, [vp, vp, sz, szp]),
    "yk_coverage_plane": (C.c_int, [vp, C.c_int, vp, sz]),
    "yk_lut_clear": (C.c_int, [vp]),
    "yk_lut_load_pattern": (C.c_int, [vp, vp, vp, vp, C.c_int, C.POINTER(C.c_int)]),
    "yk_lut_pattern_tables": (C.c_int, [vp, C.c_int, vp, vp, vp]),
    "yk_lut_start": (C.c_int, [vp]),
    "yk_lut_search": (C.c_int, [vp, C.c_int, C.c_int, C.POINTER(C.c_int)]),
    "yk_lut_stream": (C.c_int, [vp, C.c_int, vp, sz, szp]),
    "yk_gradient_corner_edges": (C.c_int, [vp, vp, vp, sz]),
    "yk_range_sizes": (C.c_int, [vp, C.c_int, szp, szp]),
    "yk_range_streams": (C.c_int, [vp, C.c_int, vp, sz, vp, sz]),
    "yk_range_defs_device": (vp, [vp, C.c_int]),
    "yk_range_nibbles_device": (vp, [vp, C.c_int]),
    "yk_range_dst": (C.c_int, [vp, C.c_int, vp, sz]),
    "yk_set_dst_fill": (C.c_int, [vp, C.c_int32]),
    "yk_range1d_encode": (C.c_int, [vp]),
    "yk_set_pixel_cache": (C.c_int, [vp, C.c_int]),
    "yk_range1d_streams": (C.c_int, [vp, vp, sz, szp, vp, sz, szp]),
    "yk_range1d_plane_ends": (C.c_int, [vp, vp, vp]),
    "yk_export_capacity": (sz, [vp]),
    "yk_export_tile_maps": (C.c_int, [vp, vp, sz, vp]),
    "yk_export_tile_maps_async": (C.c_int, [vp, vp, sz, vp, vp]),
    "yk_export_tile_maps_framed": (C.c_int, [vp, vp, sz, vp]),
    "yk_device_alloc": (C.c_int, [vp, sz, C.POINTER(vp)]),
    "yk_device_free": (None, [vp, vp]),
    "yk_device_download": (C.c_int, [vp, vp, vp, sz]),
    "yk_comm_available": (C.c_int, []),
    "yk_comm_unique_id": (C.c_int, [vp]),
    "yk_comm_init_rank": (C.c_int, [vp, vp, C.c_int, C.c_int, C.POINTER(vp)]),
    "yk_comm_init_all": (C.c_int, [C.POINTER(vp), C.c_int, C.POINTER(vp)]),
    "yk_comm_ranks": (C.c_int, [vp, ip, ip]),
    "yk_comm_destroy": (None, [vp]),
    "yk_gather_maps": (C.c_int, [vp, vp, C.c_int, vp, sz, vp, szp, szp]),
    "yk_gather_maps_all": (C.c_int, [C.POINTER(vp), C.POINTER(vp), C.c_int, C.c_int, C.POINTER(vp), szp, vp, szp]),
    "yk_stream_wait_for": (C.c_int, [vp, vp]),
    "yk_stream_handoff": (C.c_int, [vp, vp]),
    "yk_decode_begin": (C.c_int, [vp, C.c_int, C.c_int]),
    "yk_decode_gradient": (C.c_int, [vp, C.c_int, C.c_int, vp, sz, vp, sz]),
    "yk_decode_1d": (C.c_int, [vp, vp, sz, vp, sz, C.c_int]),
    "yk_decode_1d_device": (C.c_int, [vp, vp, sz, vp, sz, C.c_int]),
    "yk_decode_gradient_device": (C.c_int, [vp, C.c_int, C.c_int, vp, sz, vp, sz, C.c_int]),
    "yk_decode_gradient_all_device": (C.c_int, [vp, C.c_int, vp, vp, vp, vp, vp, vp, C.c_int]),
    "yk_range1d_streams_device": (C.c_int, [vp, C.POINTER(vp), szp, C.POINTER(vp), szp]),
    "yk_gradient_corners_device": (C.c_int, [vp, C.c_int, C.POINTER(vp), szp]),
    "yk_decode_mask": (C.c_int, [vp, vp, C.c_int, C.c_int, vp, sz]),
    "yk_decode_planes": (C.c_int, [vp, vp, vp, vp, sz]),
    "yk_decode_planes_device": (vp, [vp, szp]),
    "yk_gradient_corners_run": (C.c_int, [vp]),
    "yk_stage_ms": (C.c_int, [vp, C.c_int, C.POINTER(C.c_float), ip]),
    "yk_measure_roof": (C.c_int, [vp, sz, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "yk_decode_output": (C.c_int, [vp, vp, sz, vp, C.c_int]),
    "yk_decode_output_reference_rgba": (C.c_int, [vp, vp, sz, vp, C.c_int]),
    "yk_decode_tile4x4": (C.c_int, [vp, vp, sz]),
    "yk_decode_tile4x4_planes": (C.c_int, [vp, vp, sz]),
    "yk_decode_gradient_planes": (C.c_int, [vp, C.c_int, C.c_int, vp, sz, vp, sz]),
    "yk_decode_split_masks": (C.c_int, [vp]),
    "yk_decode_assign_lut": (C.c_int, [vp, vp, sz]),
    "yk_decode_lut3d": (C.c_int, [vp, vp, vp, vp, sz, vp, vp, vp, vp]),
    "yk_last_kernel_ms": (C.c_int, [vp, C.POINTER(C.c_float), C.POINTER(C.c_float), C.POINTER(C.c_float)]),
}

# include/yaik_hip_test.h: only in the -DYK_TEST_HOOKS build (tests/csrc/libyaik_hip_test.so), never in the product library
TEST_SIGNATURES = {
    "yk_selftest": (C.c_int, [vp, C.c_int, ip]),
    "yk_set_ablation": (C.c_int, [vp, C.c_int]),
    "yk_set_kernel_version": (C.c_int, [vp, C.c_int]),
    "yk_set_cross_check_launcher": (C.c_int, [vp]),
}
TEST_LIB_PATH = os.environ.get("YK_TEST_LIB") or os.path.join(os.path.dirname(HERE), "tests", "csrc", "libyaik_hip_test.so")

_lib = None
_test_lib = None


def test_lib() -> C.CDLL:
    """The test build of the library (same sources + the hooks of include/yaik_hip_test.h).  Test infrastructure: only tests/ and tools/ load it."""
    global _test_lib
    if _test_lib is None:
        lib()                                                # the product library first (and torch's HIP runtime before both)
        if not os.path.exists(TEST_LIB_PATH):
            raise YaikError(f"{TEST_LIB_PATH} is missing: make -C yaik_amd/csrc")
        L = C.CDLL(TEST_LIB_PATH)
        for name, (res, args) in {**SIGNATURES, **TEST_SIGNATURES}.items():
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        _test_lib = L
    return _test_lib


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise YaikError(f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                            "(there is no CPU fallback for the product path)")
        # PyTorch bundles its own HIP runtime (same SONAME as the system one).  Brought up first, this library binds to that
        # one instance; in the reverse order the process holds two and torch fails with "No HIP GPUs are available".
        # torch is only plumbing here, so this is best effort.
        try:
            import torch
            if torch.cuda.is_available():
                torch.cuda.init()
        except Exception:  # noqa: BLE001
            pass
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib
