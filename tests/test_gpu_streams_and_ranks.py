"""Stream hand-over of the C-ABI (yk_set_stream after work was queued on the old stream) and the N > 1 product path on real devices:
the 2-rank RCCL run of bench.py.  The latter needs two GPUs and is skipped on the 1-GPU boxes this repo is developed on -- until it has
run somewhere, the nccl + device-side hand-over branch of bench.py stays UNVERIFIED (README)."""
import ctypes as C
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from tests.images import edge_image, synth_planes

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_stream_switch_between_set_image_and_encode(oracle_built):
    """yk_set_image / yk_upload_planes queue clears and copies on the handle's stream; a caller that then passes its own stream
    (include/yaik_hip.h: "pass the producer's / consumer's stream here") must still see them finished: yk_set_stream orders the new
    stream after the old one."""
    from oracle.pyoracle import PASSES, OracleEncoder
    from yaik_amd._lib import lib
    from yaik_amd.encoder import HipTileEncoder
    for planes in (synth_planes(512, n_planes=4), edge_image(256, 256, "mixed", 3)):
        enc = HipTileEncoder(0)
        try:
            s = torch.cuda.Stream()
            enc.set_image(planes)                                   # host planes: uploaded on the handle's own stream
            assert lib().yk_set_stream(enc._h, C.c_void_p(s.cuda_stream)) == 0
            if planes.shape[0] == 4:
                enc.mip_prefilter()
            enc.encode(3, False, False)
            ora = OracleEncoder(planes)
            if planes.shape[0] == 4:
                ora.mip_prefilter()
            for i, (sx, sy) in enumerate(PASSES):
                cnt, bm, rgb = ora.fitting_quad_smooth(sx, sy)
                assert np.array_equal(enc.gradient_bitmap(i), bm), i
            for p in range(3):
                defs, nib, nn, _ = ora.dynamic_tile_encode(p, False)
                d2, n2, nn2 = enc.range_streams(p)
                assert nn2 == nn and np.array_equal(d2, defs) and np.array_equal(n2, nib), p
            # and back to the handle's own stream with work pending on the caller's
            assert lib().yk_set_stream(enc._h, None) == 0
            enc.encode(3, False, False)
            assert np.array_equal(enc.gradient_bitmap(0), ora_first_bitmap(planes))
        finally:
            enc.close()
        s.synchronize()


def ora_first_bitmap(planes):
    from oracle.pyoracle import OracleEncoder
    o = OracleEncoder(planes)
    if planes.shape[0] == 4:
        o.mip_prefilter()
    return o.fitting_quad_smooth(4, 4)[1]


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs (the N > 1 RCCL path is unverified until this has run)")
@pytest.mark.parametrize("layout", ["frames", "stripes"])
def test_two_rank_rccl_bench(layout):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", "29631",
           os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "2", "--no-cpu"] + (["--layout", "stripes"] if layout == "stripes" else [])
    out = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    line = [ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1]
    d = json.loads(line)
    assert d["n_gpus"] == 2 and d["value"] > 0
    assert str(d.get("gather_check", "")).startswith("ok"), d.get("gather_check")
