"""CPU suite: host-side logic, and that the C-ABI library loads and exports every symbol include/yaik_hip.h declares
(no compute calls: there is no GPU here, and the product path must refuse to run without one)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    hdr = open(os.path.join(ROOT, "include", "yaik_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    return sorted(set(re.findall(r"\b(yk_[a-z0-9_]+)\s*\(", hdr)))


def test_library_exports_every_declared_symbol():
    from yaik_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        _lib.build()
    L = C.CDLL(_lib.LIB_PATH)
    declared = _declared_symbols()
    assert len(declared) >= 35
    missing = [s for s in declared if not hasattr(L, s)]
    assert not missing, missing
    assert sorted(_lib.SIGNATURES) == declared, (set(declared) ^ set(_lib.SIGNATURES))


def test_product_library_carries_no_test_hooks():
    """The hooks of include/yaik_hip_test.h (self-tests, ablations, cross-check registration) exist only in the test build."""
    from yaik_amd import _lib
    hdr = open(os.path.join(ROOT, "include", "yaik_hip_test.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    hooks = sorted(set(re.findall(r"\b(yk_[a-z0-9_]+)\s*\(", hdr)))
    assert hooks == sorted(_lib.TEST_SIGNATURES) and len(hooks) == 4
    product = C.CDLL(_lib.LIB_PATH)
    assert not [s for s in hooks if hasattr(product, s)]
    test = C.CDLL(_lib.TEST_LIB_PATH)
    assert not [s for s in hooks + _declared_symbols() if not hasattr(test, s)]


def test_product_path_fails_loudly_without_gpu():
    """No CPU fallback: without a HIP device the handle cannot even be created."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    from yaik_amd._lib import YaikError
    from yaik_amd.encoder import HipTileEncoder
    with pytest.raises(YaikError):
        HipTileEncoder(0)


def test_product_sources_never_touch_the_oracle():
    """Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use oracle/."""
    pkg = os.path.join(ROOT, "yaik_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp", ".hpp")) or f == "Makefile":
                text = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "oracle" not in text.replace("checked against the oracle", ""), os.path.join(dirpath, f)


def test_curve_constants_match_glibc_powf(oracle_built):
    """yk_curves.h (hex floats baked into the kernels) == the constants DynamicTile::buildTable derives with powf."""
    text = open(os.path.join(ROOT, "yaik_amd", "csrc", "yk_curves.h")).read()
    rows = re.findall(r"/\* (\w+) \*/ \{([^}]*)\}", text)
    table = {name: [float.fromhex(x.strip().rstrip("f")) for x in body.split(",")] for name, body in rows}
    ref = np.zeros(48, dtype=np.float32)
    oracle_built.lib().yko_curve_constants(ref.ctypes.data)
    assert np.array_equal(np.array(table["exp4"], np.float32), ref[0:16])
    assert np.array_equal(np.array(table["log4"], np.float32), ref[16:32])
    assert np.array_equal(np.array(table["exp3"][:8], np.float32), ref[32:40])
    assert np.array_equal(np.array(table["log3"][:8], np.float32), ref[40:48])
    lin4 = (np.arange(16, dtype=np.float32) / np.float32(15.0)).astype(np.float32)
    lin3 = (np.arange(8, dtype=np.float32) / np.float32(7.0)).astype(np.float32)
    assert np.array_equal(np.array(table["lin4"], np.float32), lin4)
    assert np.array_equal(np.array(table["lin3"][:8], np.float32), lin3)


def test_blend_reformulation_is_exact():
    """The kernels test D = S' - 256*cur instead of the reference's 1024*1024-scaled blend (EncoderContext.cpp:3929-3991).
    Exhaustive over all weights and a dense sample of corner values: blendCO == S'>>8 and blendC == (S'+127)>>8."""
    rng = np.random.default_rng(0)
    for T in (4, 8, 16):
        wts = 1024 - np.arange(T) * (1024 // T)
        lF, tF = np.meshgrid(wts, wts)
        for _ in range(200):
            tl, tr, bl, br = rng.integers(0, 256, 4)
            for c in ((tl, tr, bl, br), (0, 255, 255, 0), (255, 255, 255, 255), (255, 0, 0, 255)):
                a, b, c2, d = [int(v) for v in c]
                S = (a * lF + b * (1024 - lF)) * tF + (c2 * lF + d * (1024 - lF)) * (1024 - tF)
                blendC, blendCO = (S + (1 << 19) - 1) // (1 << 20), S // (1 << 20)
                lx, wy = lF // 64, tF // 64
                Sp = (a * lx + b * (16 - lx)) * wy + (c2 * lx + d * (16 - lx)) * (16 - wy)
                assert Sp.max() <= 65280
                assert np.array_equal(blendCO, Sp >> 8) and np.array_equal(blendC, (Sp + 127) >> 8)


def test_synth_generators_agree():
    import torch
    from yaik_amd.synth import synth_planes, synth_planes_torch
    for w, h, n, seed in ((64, 64, 4, 12345), (256, 128, 3, 7), (512, 512, 4, 12346)):
        assert np.array_equal(synth_planes(w, h, n_planes=n, seed=seed), synth_planes_torch(w, h, n_planes=n, seed=seed, device="cpu").numpy())
    assert not torch.cuda.is_available() or True


def test_stripe_partition_and_stream_concatenation():
    from yaik_amd import distributed as ykd
    for full_h, world in ((16384, 8), (8192, 8), (4096, 3), (2048, 5), (1024, 5), (64, 4), (200 * 8, 7)):
        rows = [ykd.stripe_rows(full_h, world, r) for r in range(world)]
        y = 0
        for y0, h, halo in rows:
            assert y0 == y and y0 % 64 == 0
            y += h
            if h:
                assert halo == (1 if y0 + h < full_h else 0)
                assert h % 64 == 0 or y0 + h == full_h
        assert y == full_h
        blocks = (full_h + 63) // 64
        if blocks >= world:
            assert all(h >= 8 for _, h, _ in rows), (full_h, world, rows)      # nobody is left without rows while there are enough blocks
            assert max(h for _, h, _ in rows) - min(h for _, h, _ in rows) <= 64
        else:
            assert [ykd.stripe_is_empty(full_h, world, r) for r in range(world)] == [r >= blocks for r in range(world)]
    assert ykd.combine_bboxes([[9999999, 9999999, -1, -1], [32, 64, 200, 128], [16, 256, 100, 300]]).tolist() == [16, 64, 200, 300]
    # nibble streams of stripes concatenate into the image-wide stream even across odd boundaries
    rng = np.random.default_rng(1)
    parts, counts, allnib = [], [], []
    for n in (5, 0, 8, 3, 1):
        nib = rng.integers(0, 16, n).astype(np.uint8)
        allnib.append(nib)
        packed = np.zeros((n + 1) // 2, np.uint8)
        for i, v in enumerate(nib):
            packed[i >> 1] |= v << ((i & 1) * 4)
        parts.append(packed); counts.append(n)
    out, total = ykd.concat_nibble_streams(parts, counts)
    flat = np.concatenate(allnib)
    assert total == flat.size
    want = np.zeros((flat.size + 1) // 2, np.uint8)
    for i, v in enumerate(flat):
        want[i >> 1] |= v << ((i & 1) * 4)
    assert np.array_equal(out, want)


def test_merge_corner_streams_drops_the_later_emission():
    """Root-side reconciliation of the lattice row two stripes share: the copy of the later pass goes, stripe s+1's on a tie."""
    from yaik_amd.distributed import merge_corner_streams
    NONE = 0xFFFFFFFF
    n = 5

    def key(p, pos, corner):
        return (p << 27) | (pos << 2) | corner
    # stripe 0: pass 0 emits corners a0,a1 (a1 = boundary point x=1), pass 2 emits a2 (boundary x=3)
    s0 = [np.array([10, 11, 12, 20, 21, 22], np.uint8)] + [np.zeros(0, np.uint8)] + [np.array([30, 31, 32], np.uint8)] + [np.zeros(0, np.uint8)] * 4
    # stripe 1: pass 0 emits b0 (boundary x=3), pass 1 emits b1 (boundary x=1) and b2 (interior)
    s1 = [np.array([40, 41, 42], np.uint8), np.array([50, 51, 52, 60, 61, 62], np.uint8)] + [np.zeros(0, np.uint8)] * 5
    k0 = np.full((2, n), NONE, np.uint32); i0 = np.full((2, n), NONE, np.uint32)
    k1 = np.full((2, n), NONE, np.uint32); i1 = np.full((2, n), NONE, np.uint32)
    k0[1, 1], i0[1, 1] = key(0, 7, 2), 1          # stripe 0 last row, x=1: pass 0, second corner of its stream
    k0[1, 3], i0[1, 3] = key(2, 5, 3), 0          # x=3: pass 2, first corner
    k1[0, 1], i1[0, 1] = key(1, 0, 0), 0          # stripe 1 first row, x=1: pass 1 (later than stripe 0's pass 0 -> dropped)
    k1[0, 3], i1[0, 3] = key(0, 2, 1), 0          # x=3: pass 0 (earlier than stripe 0's pass 2 -> stripe 0's copy dropped)
    k1[0, 4], i1[0, 4] = key(0, 9, 0), 0          # only stripe 1 touches x=4: kept (index reused here on purpose: not consulted)
    out = merge_corner_streams([s0, s1], [(k0, i0), (k1, i1)])
    assert out[0].tolist() == [10, 11, 12, 20, 21, 22, 40, 41, 42]
    assert out[1].tolist() == [60, 61, 62]
    assert out[2].tolist() == []
    # tie on the pass: the upper stripe's tiles come first in scan order
    k1[0, 1] = key(0, 0, 0); i1[0, 1] = 0
    s1b = [np.array([40, 41, 42, 70, 71, 72], np.uint8), np.array([60, 61, 62], np.uint8)] + [np.zeros(0, np.uint8)] * 5
    i1[0, 3] = 0; i1[0, 1] = 1
    out = merge_corner_streams([s0, s1b], [(k0, i0), (k1, i1)])
    assert out[0].tolist() == [10, 11, 12, 20, 21, 22, 40, 41, 42] and out[1].tolist() == [60, 61, 62]
