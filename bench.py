#!/usr/bin/env python
"""bench.py — headline metric of BASELINE.json: Mpix/s of YAIK tile encode (alpha tile-reject + 7 gradient passes +
8x8 4-bpp range quantiser of 3 planes) on an 8192x8192 RGBA frame of synthetic "YAIK-synth v1" data, inputs resident in HBM.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--size 8192] [--mode3] [--in-flight F] [--no-cpu] [--no-parity]
                  [--stage encode|all|corners|range1d|decode|lut3d] [--layout frames|stripes]

Default (--stage encode --layout frames) is the headline line described below.  --stage all = every GPU stage of ConvertHotPath per frame
(encode + corner streams + live 1-D range path) with frames in flight; --stage corners | range1d | decode | lut3d time the other
stages of the path on the same frame (corner-colour streams, live 1-D range path, GPU decode) and print a `roofline` for the
stage's dominant kernel; --layout stripes (N > 1) encodes ONE image as row stripes: stripe bbox all-reduce, one RCCL gather of the
tile maps, checked against a whole-image encode on rank 0 (strong scaling).

One "step" = one pass of the hot path over a batch of F frames per GPU kept in flight on F handles/streams (--in-flight F, default 2:
the HBM-bound alpha / pack kernels of one frame run under the VALU-bound fused kernel of the other; --in-flight 1 = one frame at a time):
    yk_alpha_reject -> yk_alpha_finish -> yk_encode_tiles (fused gradient+range kernel, then stream compaction)
    and, for N > 1, ONE RCCL gather of the per-rank tile maps onto rank 0 (frame sharding, weak scaling), double-buffered so
    that the transfer of frame i rides under the kernels of frame i+1, and verified by checksums outside the timed region.
Frames are queued back to back (no host synchronisation inside the timed loop); per-kernel times come from HIP events on the
launch stream, averaged over the steps.  N > 1 is launched by torch.distributed.run (one rank per GPU).  Rank 0 prints ONE
JSON line: the BASELINE metric + `roofline` (fused kernel, algorithmic bytes / event time; `traffic` from the committed PMC
pass) + `cpu_baseline` (the CPU restatement on one host core of this box, with the measured restatement/reference ratio).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0        # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6.3 TB/s is what a copy achieves
ABI_DEADLINE_S = 60.0        # post-timing C-ABI gather check (N > 1): a check that has not returned by then is a FAILED run (exit 3), never "ok"
ABI_INIT_DEADLINE_S = 45.0   # the communicator bootstrap inside it


def kernel_source_hash() -> str:
    import hashlib
    with open(os.path.join(ROOT, "yaik_amd", "csrc", "yk_encode2.hip"), "rb") as f:
        return hashlib.sha256(f.read()).hexdigest()


def _timed(steps, warmup, step, fence, world, dist, comm_dev):
    """W untimed steps, then exactly K steps between two fences (barrier + synchronize); returns the MAX over ranks of the elapsed time."""
    import torch
    for _ in range(warmup):
        step()
    fence()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    fence()
    elapsed = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=comm_dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    return elapsed


def bench_stage(args, rank, world, dist, dev, dev_index, comm_dev) -> int:
    """--stage corners | range1d | decode: the stages of the path outside the headline encode, one frame of --size per rank (replicas for
    N > 1: these stages shard like the encode, no collective).  `value` is the stage's whole-call rate including what its C-ABI entry
    point does on the host side (stream sizes read back, and for decode the host->device copies of the streams and the image coming
    back, as the YAIK.h boundary hands over host buffers); `roofline` is the dominant kernel alone, event-timed on the launch stream."""
    import numpy as np
    import torch
    from yaik_amd.decoder import HipTileDecoder
    from yaik_amd.encoder import HipTileEncoder
    from yaik_amd.synth import synth_planes_torch
    W = args.size
    planes = synth_planes_torch(W, n_planes=4, seed=12345 + rank, device=dev)
    torch.cuda.synchronize()
    enc = HipTileEncoder(dev_index)
    enc.set_image(planes)
    cached = args.stage == "range1d" and not args.no_pixel_cache
    if cached:
        enc.set_pixel_cache(True)                            # the 1-D path reads the fused kernel's pixel cache (4 B per uncovered pixel), not the planes
    enc.alpha_reject(); enc.alpha_finish(None)
    enc.encode(3, args.mode3, False)
    enc.synchronize()

    def fence():
        torch.cuda.synchronize()
        enc.synchronize()
        if dec is not None:
            dec.synchronize()
        if world > 1:
            dist.barrier()

    dec = None
    ST = {"corners": (0,), "range1d": (1, 2), "decode": (3, 4, 5), "lut3d": (6,)}[args.stage]
    if args.stage == "corners":
        step = enc.gradient_corners_run
    elif args.stage == "range1d":
        step = lambda: lib_call(enc)
    elif args.stage == "lut3d":
        # SURVEY 8(f)4 on the synthetic bank (the reference's own is not in its repository): the six search passes of Convert() on what the
        # gradient passes of the bench frame left uncovered.  Every step re-encodes first (untimed part of the step would distort: the
        # encode is 0.4 ms against tens of ms of search), because a search pass consumes the coverage it paints.
        from yaik_amd.synth import bank_patterns
        for pat in bank_patterns():
            enc.lut_load(pat)
        lut_matched = [0]

        def step():
            enc.encode(3, args.mode3, False)
            enc.lut_start()
            lut_matched[0] = sum(enc.lut_search(sx, sy) for sx, sy in ((4, 3), (3, 4), (3, 3), (3, 2), (2, 3), (2, 2)))
    else:
        # streams for the decoder come from the GPU encoder: tile bitmaps, corner streams (de-quantised like PaletteFullRangeRemapping,
        # decoder/YAIK_GenericFunctions.cpp:128-137: v * ((255 << 16) / 250) >> 16) and the 1-D streams
        bitmaps = [enc.gradient_bitmap(i) for i in range(7)]
        inv = (255 << 16) // 250
        corners = [((enc.gradient_corners(i).astype(np.uint32) * inv) >> 16).astype(np.uint8) for i in range(7)]
        counts = enc.gradient_counts()
        pix, typ = enc.dynamic_tile_compressor()
        dec = HipTileDecoder(dev_index)
        out = np.zeros((W, W * 3), dtype=np.uint8)
        shapes = [(4, 4), (4, 3), (3, 4), (3, 3), (3, 2), (2, 3), (2, 2)]

        if args.device_streams:
            # the encoder's outputs feed the decoder where they lie in HBM (BASELINE config 5: encode + decode round trip on the GPU): no host
            # streams, no PCIe; a step = all gradient passes + the 1-D chunk, the image stays in HBM as 8x8-tiled planes
            dev_calls = dec.encoder_streams(enc)

            def step():
                dec.begin(W, W)
                dec.decode_streams(dev_calls, sync=False)
        else:
            def step():
                dec.begin(W, W)
                for i, (sx, sy) in enumerate(shapes):
                    if counts[i]:
                        dec.decompress_gradient(sx, sy, bitmaps[i], corners[i])
                dec.decompress_1d(typ, pix)
                dec.image_into(out)

    def lib_call(e):
        from yaik_amd._lib import lib
        from yaik_amd.encoder import _chk
        _chk(e._h, lib().yk_range1d_encode(e._h))

    timer = dec if dec is not None else enc
    for _ in range(args.warmup):
        step()
    for st in ST:
        timer.stage_ms(st)                                   # drop the warm-up intervals
    elapsed = _timed(args.steps, 0, step, fence, world, dist, comm_dev)
    ms = {st: timer.stage_ms(st) for st in ST}
    if rank != 0:
        if world > 1:
            dist.destroy_process_group()
        return 0
    per = {st: (ms[st][0] / max(1, args.steps)) for st in ST}  # kernel ms per step
    nd_nn = [enc.range_streams(p) for p in range(3)]
    cov = enc.coverage()
    uncovered = int((~cov).sum()) * 16                       # pixels the gradient passes left to the 1-D path
    bitmap_bytes = sum(enc.gradient_bitmap(p).size for p in range(7))
    if args.stage == "corners":
        lat = (W // 4 + 1) ** 2
        nb = sum(enc.gradient_corners(i).size for i in range(7))
        alg = bitmap_bytes + 8 * lat + 5 * nb                # bitmaps read + lattice owner cleared and resolved (4 B per point, twice) + per colour byte: 4 B sample read, 1 B written
        kname, kms, note, bound = "yk_corner_* (lattice clear + owner + count + scan + emit, 7 passes)", per[0], "sparse scatter / gather: latency- and atomics-bound, far from the HBM roof by construction", "latency"
    elif args.stage == "range1d":
        pixn = uncovered * 3
        alg = (4 if cached else 12) * uncovered + pixn + (W // 8) * (W // 8) * 3 * 3   # the packed pixels (cache) or the int32 samples of the uncovered 4x4 cells + 1 B per uncovered pixel and plane written + 3 parameter bytes per tile-plane
        kname, kms, note, bound = "yk_range1d_kernel", per[1], f"tile offsets from the coverage + one scan (the coder writes straight into the streams): {per[2]:.4f} ms per frame on top", "hbm"
    elif args.stage == "lut3d":
        cand = int((~cov).sum()) * 16                         # pixels of tiles with anything left to code, an upper bound of what the passes read
        alg = 12 * cand * 6                                   # every pass reads the three int32 samples of its candidate tiles' pixels once
        kname, kms, bound = "yk_lut_search_kernel (6 tile shapes)", per[6], "valu"
        note = (f"VALU-bound, not byte-bound: <= 128 pixels x 6 patterns x 48 orientations x 8 points squared distances per candidate tile (v_mfma_i32_32x32x16_i8 for tiles of 32+ pixels, v_dot4_i32_i8 for 4x4), then four entry depths per pixel and pattern; "
                f"{lut_matched[0]} tiles matched on this frame; algorithmic bytes = the candidate tiles' samples once per pass")
    else:
        pixn = uncovered * 3
        alg = 2 * pixn + (W // 8) * (W // 8) * 9             # yk_dec1d_kernel: 1 B per uncovered pixel and plane read, the same written, 3 parameter bytes per tile-plane
        kname, kms, bound = "yk_dec1d_kernel (+ count / scan kernels)", per[4], "hbm"
        gcalls = ms[3][1] // max(1, args.steps)
        gform = ("all passes in ONE call (yk_decode_gradient_all_device: lattice clear + owner + count + scan + emit for all passes, then one render per pass = 11 launches)"
                 if args.device_streams and gcalls == 1 else f"owner/corner/render x{gcalls} passes (5 launches + 3 copies / clears each)")
        note = (f"other decode kernels per frame: gradient {gform} {per[3]:.4f} ms, "
                f"yk_dec_detile_kernel {per[5]:.4f} ms (6 B/pixel moved = {6 * W * W / (per[5] * 1e-3) / 1e9 if per[5] > 0 else 0:.0f} GB/s)")
    achieved = alg / (kms * 1e-3) / 1e9 if kms > 0 else 0.0
    result = {
        "metric": f"Mpix/s {args.stage} stage, 8K RGBA frame" if W == 8192 else f"Mpix/s {args.stage} stage, {W}x{W} RGBA frame",
        "value": round(W * W * world * args.steps / 1e6 / elapsed, 1), "unit": "Mpix/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "u8/int32", "data": "synthetic (YAIK-synth v1, seed 12345+rank)",
        "config": {"workload": f"{W}x{W} RGBA frame per GPU, stage '{args.stage}' of the tile path after a full encode; a step is one call sequence through the "
                               "C-ABI incl. its host-side part (decode: " + ("the encoder's streams read where they lie in HBM, 8x8-tiled planes left in HBM: no PCIe" if getattr(args, "device_streams", False) and args.stage == "decode" else "host streams in, interleaved RGB image out over PCIe") + ")",
                   "parallelism": f"replicas x{world} (no collective)" if world > 1 else "single GPU"},
        # `bound` names what limits the stage's dominant kernel; achieved / frac are its algorithmic bytes against the HBM figure either way
        # (for a latency- or VALU-bound kernel that fraction says how far from a byte-bound kernel it is, not how well it is tuned)
        "roofline": {"bound": bound, "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4),
                     "traffic": None, "kernel": kname, "kernel_ms": round(kms, 4), "algorithmic_bytes": int(alg), "note": note},
    }
    print(json.dumps(result))
    if world > 1:
        dist.destroy_process_group()
    return 0


def bench_all(args, rank, world, dist, dev, dev_index, comm_dev) -> int:
    """--stage all: everything the GPU does for a frame of ConvertHotPath, frames in flight on --in-flight handles (default 3):
        alpha tile-reject -> fused gradient / range kernel -> stream compaction -> the seven corner-colour streams (a6 rgbStream) ->
        the live 1-D range path (a15, its scan + pack).
    No call in the loop waits on the host (the corner / 1-D stream lengths stay on the device until someone reads a stream); the fused kernels
    of the frames take turns, and the memory-bound helper stages of one frame (alpha, compaction, corners: atomics / latency, 1-D: HBM) run
    under the VALU-bound fused kernel of another.  N > 1: replicas (these stages shard like the encode)."""
    import numpy as np
    import torch
    from yaik_amd._lib import lib
    from yaik_amd.encoder import HipTileEncoder, _chk
    from yaik_amd.synth import synth_planes_torch
    W = args.size
    K = max(1, args.in_flight if args.in_flight != 2 else 3)
    frames = [synth_planes_torch(W, n_planes=4, seed=12345 + rank * K + j, device=dev) for j in range(K)]
    torch.cuda.synchronize()
    encs = [HipTileEncoder(dev_index) for _ in range(K)]
    for e, f in zip(encs, frames):
        e.set_image(f)
    L = lib()
    ordered = K > 1 and not args.free_overlap and W * W >= 8192 * 8192
    if not args.no_pixel_cache:
        for e in encs:
            e.set_pixel_cache(True)                          # the fused kernel leaves the uncovered cells' pixels for the 1-D path: planes read once

    def step():
        for j, e in enumerate(encs):
            if ordered:
                e.order_fused_after(encs[j - 1])
            e.alpha_reject()
            e.alpha_finish(None)
            e.encode(3, args.mode3, False)
            e.gradient_corners_run()
            _chk(e._h, L.yk_range1d_encode(e._h))

    def fence():
        torch.cuda.synchronize()
        for e in encs:
            e.synchronize()
        if world > 1:
            dist.barrier()

    for _ in range(args.warmup):
        step()
    fence()
    for e in encs:
        e.kernel_ms()
        for st in (0, 1, 2):
            e.stage_ms(st)
    kms = {"encode": 0.0, "alpha": 0.0, "pack": 0.0}
    t0 = time.perf_counter()
    for i in range(args.steps):
        step()
        if (i + 1) % 32 == 0 or i + 1 == args.steps:
            done = (i % 32) + 1
            for e in encs:
                k = e.kernel_ms()
                for n in kms:
                    kms[n] += k[n] * done / len(encs)
    fence()
    elapsed = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=comm_dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    for n in kms:
        kms[n] /= max(1, args.steps)
    st = {s_: [0.0, 0] for s_ in (0, 1, 2)}
    for e in encs:
        for s_ in st:
            m = e.stage_ms(s_)
            st[s_][0] += m[0]; st[s_][1] += m[1]
    per = {s_: (st[s_][0] / st[s_][1] if st[s_][1] else 0.0) for s_ in st}   # mean interval of the stage's kernels per frame (sharing the chip)
    if rank != 0:
        if world > 1:
            dist.destroy_process_group()
        return 0
    enc = encs[0]
    # algorithmic bytes of the whole path per frame: SURVEY 8(d) for the encode (four int32 planes read once + bitmaps + definitions + nibbles)
    # + the corner streams written + the 1-D path's planes read once more, its pixel bytes and 3 parameter bytes per coded tile-plane
    nd_nn = [enc.range_streams(p) for p in range(3)]
    bitmap_bytes = sum(enc.gradient_bitmap(p).size for p in range(7))
    out_bytes = sum(2 * d.size + nb.size for d, nb, _ in nd_nn)
    corner_bytes = sum(enc.gradient_corners(i).size for i in range(7))
    pix, typ = enc.dynamic_tile_compressor()
    # every input sample once (16 B/pixel) + every output of the path.  With the pixel cache the path does read the planes once (the 1-D
    # coder takes the uncovered cells' pixels from the fused kernel: an intermediate like the nibble slots, not counted); without it the 1-D coder's
    # second read of the planes is REAL traffic but not algorithmic, so it is not counted either -- the fraction then shows the waste.
    alg = 16 * W * W + bitmap_bytes + out_bytes + corner_bytes + pix.size + typ.size
    ms_frame = elapsed / args.steps * 1e3 / K
    roof = max(measured_stream_roof(enc))
    achieved = alg / (ms_frame * 1e-3) / 1e9
    result = {
        "metric": "Mpix/s whole tile path on the GPU (alpha reject + fused gradient/range kernel + compaction + corner streams + 1-D range path), 8K RGBA"
                  if W == 8192 else f"Mpix/s whole tile path on the GPU, {W}x{W} RGBA",
        "value": round(W * W * world * K * args.steps / 1e6 / elapsed, 1), "unit": "Mpix/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "u8/int32 (+f32 mode-selection sums)", "data": "synthetic (YAIK-synth v1, seed 12345+rank)",
        "config": {"workload": f"{W}x{W} RGBA frame per GPU, every GPU stage of ConvertHotPath, inputs resident in HBM, outputs left in HBM",
                   "frames_per_step": world * K, "frames_in_flight_per_gpu": K, "ms_per_frame": round(ms_frame, 4),
                   "parallelism": f"replicas x{world} (no collective)" if world > 1 else "single GPU"},
        "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4),
                     "peak_measured": round(roof, 1), "frac_of_measured": round(achieved / roof, 4), "traffic": None,
                     "kernel": "whole path (all kernels of a frame; time = wall clock per frame with the frames in flight)", "kernel_ms": round(ms_frame, 4),
                     "algorithmic_bytes": int(alg),
                     "stages_ms_per_frame": {"fused yk_encode2_kernel": round(kms["encode"], 4), "alpha": round(kms["alpha"], 4), "scan+pack": round(kms["pack"], 4),
                                             "corner streams": round(per[0], 4), "1-D range kernel": round(per[1], 4), "1-D scan+pack": round(per[2], 4)},
                     "note": "stage intervals are event-timed on each frame's stream and include the time a stage shares the chip with other frames' kernels; "
                             "whole-path ms per frame minus the fused kernel = what the helper stages could not hide"},
    }
    print(json.dumps(result))
    if world > 1:
        dist.destroy_process_group()
    return 0


def bench_stripes(args, rank, world, dist, dev, dev_index, comm_dev, rehearsal) -> int:
    """--layout stripes: ONE --size x --size RGBA image, rank r owns a band of 64-row blocks (+ 1 halo row it reads but does not own).
    Per step: alpha reject on the stripe -> stripe bbox -> two tiny all-reduces (image-wide kept-tile box) -> alpha finish -> fused
    encode + compaction -> export -> ONE gather of the tile maps onto rank 0 (double-buffered: the transfer of step i rides under the
    kernels of step i+1).  Outside the timed region rank 0 encodes the whole image itself and checks that the gathered stripes
    concatenate to it bit for bit (bitmaps, tile definitions, nibble streams)."""
    import numpy as np
    import torch
    from yaik_amd import distributed as ykd
    from yaik_amd.encoder import HipTileEncoder
    from yaik_amd.synth import synth_planes_torch
    W = args.size
    y0, h, halo = ykd.stripe_rows(W, world, rank)
    enc = None
    if h:
        stripe = synth_planes_torch(W, W, n_planes=4, seed=12345, device=dev, row0=y0, rows=h + halo)
        torch.cuda.synchronize()
        enc = HipTileEncoder(dev_index)
        enc.set_image(stripe, full_h=W, y0=y0, halo_rows=halo)
    cap = torch.tensor([enc.export_capacity() if enc else 0], dtype=torch.int64, device=comm_dev)
    if world > 1:
        dist.all_reduce(cap, op=dist.ReduceOp.MAX)
    pipe = ykd.TileMapGatherPipeline(dist, comm_dev, int(cap.item()), dst=0, staging_device=dev) if world > 1 else None
    empty_sizes = np.zeros(15, dtype=np.int64)

    def step():
        box = np.array(ykd.EMPTY_BBOX, dtype=np.int32)
        if enc:
            enc.alpha_reject()
            box = enc.stripe_bbox()
        gb = ykd.allreduce_bbox(box, dist, comm_dev) if world > 1 else box
        if enc:
            enc.alpha_finish(gb)
            enc.encode(3, args.mode3, False)
        if world > 1:
            blob, _ = pipe.acquire()
            if enc:
                enc.export_tile_maps_framed(blob, torch.cuda.current_stream(dev).cuda_stream)
            else:
                pipe.put(np.zeros(0, np.uint8), empty_sizes)   # a rank without rows still joins the gather, with an empty payload (its header)
            pipe.submit()

    def fence():
        torch.cuda.synchronize()
        if enc:
            enc.synchronize()
        if world > 1:
            pipe.flush()
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    if enc and args.warmup:
        enc.kernel_ms()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    elapsed = time.perf_counter() - t0
    kms = enc.kernel_ms() if enc else {"encode": 0.0, "alpha": 0.0, "pack": 0.0}
    if world > 1:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=comm_dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())

    # ---- outside the timed region: one more gather, checked on rank 0 against a whole-image encode ----
    gather_check = None
    if world > 1:
        blob, _ = pipe.acquire()
        if enc:
            enc.export_tile_maps_framed(blob, torch.cuda.current_stream(dev).cuda_stream)
        else:
            pipe.put(np.zeros(0, np.uint8), empty_sizes)
        pipe.submit()
        res = pipe.flush()[-1]
        if rank == 0:
            if W > 16384:
                gather_check = "skipped (whole image too large for one reference encode)"
            else:
                whole = HipTileEncoder(dev_index)
                whole.set_image(synth_planes_torch(W, n_planes=4, seed=12345, device=dev))
                whole.alpha_reject(); whole.alpha_finish(None)
                whole.encode(3, args.mode3, False)
                parts = [ykd.split_blob(sz, payload) for sz, payload in res if int(sz[14])]
                bad = []
                for i in range(7):
                    if not np.array_equal(np.concatenate([p["bitmaps"][i] for p in parts]), whole.gradient_bitmap(i)):
                        bad.append(f"bitmap {i}")
                for pl in range(3):
                    d, nb, nn = whole.range_streams(pl)
                    if not np.array_equal(np.concatenate([p["defs"][pl] for p in parts]), d):
                        bad.append(f"defs {pl}")
                    cat, total = ykd.concat_nibble_streams([p["nibbles"][pl] for p in parts], [p["n_nibbles"][pl] for p in parts])
                    if total != nn or not np.array_equal(cat, nb):
                        bad.append(f"nibbles {pl}")
                whole.close()
                gather_check = "ok: gathered stripes == whole-image encode (7 bitmaps, 3x tile defs, 3x nibble streams)" if not bad else "MISMATCH " + ", ".join(bad)
    if rank != 0:
        if world > 1:
            dist.destroy_process_group()
        return 0
    nblocks = (W + 63) // 64
    result = {
        "metric": "Mpix/s tile encode (alpha reject + 7 gradient passes + 8x8 range quant), 8K RGBA" if W == 8192 else
                  f"Mpix/s tile encode (alpha reject + 7 gradient passes + 8x8 range quant), {W}x{W} RGBA",
        "value": round(W * W * args.steps / 1e6 / elapsed, 1), "unit": "Mpix/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
        "dtype": "u8/int32 (+f32 mode-selection sums)", "data": "synthetic (YAIK-synth v1, seed 12345)",
        "config": {"workload": f"ONE {W}x{W} RGBA image per step, full encode ({'3' if args.mode3 else '4'}-bpp range), inputs resident in HBM",
                   "layout": "row stripes", "stripe_rows_rank0": int(h), "blocks_of_64_rows": nblocks,
                   "parallelism": (f"row stripes x{world}: one 4-int all-reduce of the stripe bounding boxes + ONE gather of the tile maps per image"
                                   if world > 1 else "single GPU (whole image = one stripe)"),
                   "collective_backend": (dist.get_backend() if world > 1 else None), "collective_world_size": (dist.get_world_size() if world > 1 else 1)},
        "rank0_kernel_ms": {k: round(v, 4) for k, v in kms.items()},
    }
    # roofline of the dominant kernel on rank 0's stripe: its algorithmic bytes (three int32 samples per owned pixel + the stripe's bitmaps,
    # tile definitions and nibbles) over the event-timed duration of yk_encode2_kernel
    nd_nn = [enc.range_streams(p) for p in range(3)] if enc else []
    out_bytes = sum(2 * d.size + nb.size for d, nb, _ in nd_nn) + (sum(enc.gradient_bitmap(p).size for p in range(7)) if enc else 0)
    alg = 12 * W * int(h) + out_bytes
    achieved = alg / (kms["encode"] * 1e-3) / 1e9 if kms["encode"] > 0 else 0.0
    result["roofline"] = {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4),
                          "traffic": None, "kernel": "yk_encode2_kernel (rank 0's stripe)", "kernel_ms": round(kms["encode"], 4), "algorithmic_bytes": int(alg)}
    if gather_check is not None:
        result["gather_check"] = gather_check
    print(json.dumps(result))
    if world > 1:
        dist.destroy_process_group()
    return 0 if (gather_check is None or not gather_check.startswith("MISMATCH")) else 1


def abi_gather_check(enc, dist, dev, rank, world, blob, need, allsums, checksum):
    """The same gather through the C-ABI (yk_comm_* / yk_gather_maps: RCCL bound by the library itself, not through torch), outside the
    timed region: returns (ok, ranks as RCCL reports them, note).  The communicator is created in a worker thread with a deadline, so a
    bootstrap that cannot complete on this node is reported instead of hanging the run."""
    import ctypes as C
    import threading
    import torch
    from yaik_amd._lib import lib
    L = lib()
    if not L.yk_comm_available():
        return True, None, "RCCL could not be loaded by libyaik_hip.so (torch's own path was used and checked)"
    idbuf = torch.zeros(128, dtype=torch.uint8)
    if rank == 0:
        arr = (C.c_uint8 * 128)()
        if L.yk_comm_unique_id(arr) != 0:
            return False, None, "yk_comm_unique_id failed"
        idbuf = torch.tensor(list(arr), dtype=torch.uint8)
    idbuf = idbuf.to(dev)
    dist.broadcast(idbuf, src=0)
    idbytes = bytes(idbuf.cpu().tolist())
    comm = C.c_void_p()
    state = {}

    def init():
        state["rc"] = L.yk_comm_init_rank(enc._h, idbytes, world, rank, C.byref(comm))
    th = threading.Thread(target=init, daemon=True)
    th.start()
    th.join(ABI_INIT_DEADLINE_S)
    hung = th.is_alive()
    flag = torch.tensor([2 if hung else (0 if state.get("rc") == 0 else 1)], dtype=torch.int32, device=dev)
    dist.all_reduce(flag, op=dist.ReduceOp.MAX)
    if int(flag.item()) == 2:
        # a communicator stuck in its bootstrap: a FAILURE of the run (the caller stops using the handle and every rank exits non-zero)
        return "hung", None, f"yk_comm_init_rank did not complete on every rank within {ABI_INIT_DEADLINE_S:.0f} s"
    if int(flag.item()):
        return False, None, f"yk_comm_init_rank failed (rc {state.get('rc')} on rank {rank})"
    n, me = C.c_int(), C.c_int()
    L.yk_comm_ranks(comm, C.byref(n), C.byref(me))
    needs = [int(t.cpu().tolist()[1]) for t in allsums]
    offs = [sum(needs[:r]) for r in range(world)]
    recv = torch.empty(sum(needs), dtype=torch.uint8, device=dev) if rank == 0 else None
    rb = (C.c_size_t * world)(*needs)
    ro = (C.c_size_t * world)(*offs)
    torch.cuda.synchronize()
    rc = L.yk_gather_maps(enc._h, comm, 0, C.c_void_p(blob.data_ptr()), need, C.c_void_p(recv.data_ptr()) if rank == 0 else None,
                          rb if rank == 0 else None, ro if rank == 0 else None)
    enc.synchronize()
    bad = [] if rc == 0 else [f"yk_gather_maps rc {rc}"]
    if rank == 0 and rc == 0:
        for r in range(world):
            if checksum(recv[offs[r] + 128: offs[r] + needs[r]]) != int(allsums[r].cpu().tolist()[0]):
                bad.append(f"rank {r}")
    flag = torch.tensor([len(bad)], dtype=torch.int32, device=dev)
    dist.all_reduce(flag, op=dist.ReduceOp.MAX)
    L.yk_comm_destroy(comm)
    if int(flag.item()):
        return False, int(n.value), "payload mismatch through yk_gather_maps: " + ", ".join(bad)
    return True, int(n.value), f"yk_gather_maps over a {int(n.value)}-rank communicator of its own: every payload arrived byte for byte"


def measured_copy_roof(dev, nbytes: int = 1 << 30, reps: int = 5) -> float:
    """GB/s of a device-to-device copy of `nbytes` on this GPU (read + write counted), outside any timed region (the
    measured roof next to the 8 TB/s specification)."""
    import torch
    a = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    b = torch.empty_like(a)
    b.copy_(a)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 0.0
    for _ in range(reps):
        e0.record()
        b.copy_(a)
        e1.record()
        e1.synchronize()
        best = max(best, 2 * nbytes / (e0.elapsed_time(e1) * 1e-3) / 1e9)
    del a, b
    return best


def measured_stream_roof(enc, nbytes: int = 1 << 30, reps: int = 5):
    """(copy GB/s counted read + write, read-only GB/s) of the library's own streaming kernels (yk_measure_roof, yaik_amd/csrc/yk_roof.hip)."""
    import ctypes as C
    from yaik_amd._lib import lib
    L = lib()
    c, r = C.c_double(0.0), C.c_double(0.0)
    rc = L.yk_measure_roof(enc._h, C.c_size_t(nbytes), reps, C.byref(c), C.byref(r))
    if rc != 0:
        raise RuntimeError(f"yk_measure_roof failed: {rc} {L.yk_last_error(enc._h)}")
    return float(c.value), float(r.value)


def visible_gpus() -> int:
    """Number of HIP devices, without initialising one (torch.cuda.device_count() only counts on this image)."""
    import torch
    return int(torch.cuda.device_count())


def launch_ranks(n: int, argv=None, count=visible_gpus) -> int:
    """`python bench.py --gpus N` outside a launcher: start N rank processes of this script as fresh children
    (python -m torch.distributed.run, one rank per GPU, rendezvous on 127.0.0.1), relay rank 0's JSON line and exit with the
    launcher's code.  Nothing in this process has touched the GPU yet (a parent that had could not safely start GPU children).
    Fewer than N devices: an error, never an N = 1 line in disguise.  YK_BENCH_BACKEND=gloo (rehearsal: all ranks on GPU 0,
    gather through host memory) is exempt from the device count, limited to 6 ranks per card."""
    import socket
    import subprocess
    argv = list(sys.argv[1:] if argv is None else argv)
    rehearsal = os.environ.get("YK_BENCH_BACKEND", "nccl") != "nccl"
    have = count()
    if rehearsal:
        if have < 1 or n > 6:
            print(f"bench.py: rehearsal (YK_BENCH_BACKEND=gloo) needs a GPU and at most 6 ranks on it (have {have} GPU, asked for {n} ranks)", file=sys.stderr)
            return 2
    elif have < n:
        print(f"bench.py: --gpus {n} needs {n} HIP devices, this node shows {have}", file=sys.stderr)
        return 2
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")        # dmabuf IPC: RCCL's intra-node transport needs it on this driver
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + argv
    proc = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
    lines = [ln for ln in proc.stdout.splitlines() if ln.startswith("{")]
    for ln in proc.stdout.splitlines():
        if not ln.startswith("{"):
            print(ln, file=sys.stderr)                       # launcher / rank chatter stays out of the one-line contract
    if lines:
        print(lines[-1])
    if proc.returncode == 0 and not lines:
        print("bench.py: the ranks exited cleanly but rank 0 printed no result line", file=sys.stderr)
        return 1
    return proc.returncode


def main() -> int:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--size", type=int, default=8192)
    ap.add_argument("--mode3", action="store_true", help="3-bpp range modes only (DynamicTileEncode mode3BitOnly)")
    ap.add_argument("--in-flight", type=int, default=2, help="frames per GPU kept in flight on separate handles/streams (a step = that many frames per "
                    "GPU).  Default 2: a stream of frames, where the HBM-bound alpha / pack kernels of one frame run under the VALU-bound fused "
                    "kernel of the other (the fused kernels themselves take turns: yk_order_fused_after); 1 = strictly one frame at a time")
    ap.add_argument("--no-pixel-cache", action="store_true", help="--stage all / range1d: the 1-D path re-reads the planes (12 B per uncovered pixel) instead of the fused kernel's pixel cache")
    ap.add_argument("--free-overlap", action="store_true", help="with --in-flight > 1: do not order the fused kernels of the frames (they then share the chip)")
    ap.add_argument("--batch", type=int, default=0, help="frames per GPU held by ONE handle and encoded with one launch per kernel (yk_encode_batch): "
                    "the form for batches of small frames (BASELINE config 4: --size 2048 --batch 32); a step = that many frames per GPU")
    ap.add_argument("--graph", action="store_true", help="launch every frame as one replayed hipGraph (yk_encode_frame): for batches of small "
                    "frames, where the ~8 stream operations per frame are what limits the rate; per-kernel times are then one interval")
    ap.add_argument("--stage", choices=["encode", "all", "corners", "range1d", "decode", "lut3d"], default="encode", help="which stage of the path a step runs: encode = the "
                    "headline (alpha reject + fused gradient/range kernel + compaction); corners = the seven corner-colour streams (a6 rgbStream); "
                    "range1d = the live 1-D range path (a15); decode = gradient + 1-D decode + de-tile on the GPU (a16, a17, a20); lut3d = the 3-D LUT tile search "
                    "(f4) on a synthetic bank, after an encode")
    ap.add_argument("--layout", choices=["frames", "stripes"], default="frames", help="N > 1: frames = every rank encodes its own frames (weak scaling, "
                    "default); stripes = ONE image of --size, rank r owns a band of 64-row blocks + 1 halo row (strong scaling)")
    ap.add_argument("--device-streams", action="store_true", help="--stage decode: feed the decoder the encoder's streams where they lie in HBM "
                    "(yk_decode_gradient_device / yk_decode_1d_device) instead of host streams over PCIe")
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU baseline leg")
    ap.add_argument("--no-parity", action="store_true", help="skip the full-size bit-exactness check against the oracle")
    ap.add_argument("--cpu-size", type=int, default=0, help="side of the centred crop timed on the CPU (default: whole frame)")
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        return launch_ranks(args.gpus)                       # before anything touches the GPU in this process
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE=1: refusing to report a {args.gpus}-GPU line from one rank", file=sys.stderr)
            return 2
        args.gpus = world

    import numpy as np
    import torch
    import torch.distributed as dist

    from yaik_amd import distributed as ykd
    from yaik_amd.encoder import HipTileEncoder
    from yaik_amd.synth import synth_planes_torch

    if not torch.cuda.is_available():
        print("bench.py needs a HIP device (no CPU fallback on the product path)", file=sys.stderr)
        return 2
    # Rehearsal switch (1-GPU box / CI): YK_BENCH_BACKEND=gloo runs every rank on GPU 0 and moves the gather through host
    # memory, so the whole N > 1 code path can be exercised without N GPUs.  The driver's real runs use nccl (= RCCL).
    backend = os.environ.get("YK_BENCH_BACKEND", "nccl")
    rehearsal = backend != "nccl"
    dev_index = 0 if rehearsal else local_rank
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    comm_dev = torch.device("cpu") if rehearsal else dev
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearsal:
            dist.init_process_group(backend, rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    if args.stage == "all":
        return bench_all(args, rank, world, dist, dev, dev_index, comm_dev)
    if args.stage != "encode":
        return bench_stage(args, rank, world, dist, dev, dev_index, comm_dev)
    if args.layout == "stripes":
        return bench_stripes(args, rank, world, dist, dev, dev_index, comm_dev, rehearsal)

    W = args.size
    K = max(1, args.in_flight)
    BF = max(0, args.batch)
    if BF:
        K = 1
        if world > 1 or args.graph:
            print("--batch is a single-GPU, single-handle mode", file=sys.stderr)
            return 2
        batch = torch.stack([synth_planes_torch(W, n_planes=4, seed=12345 + j, device=dev) for j in range(BF)]).contiguous()
        torch.cuda.synchronize()
        planes = batch[0]
        encs = [HipTileEncoder(dev_index)]
        encs[0].set_batch(batch)
        K = BF                                   # frames per step
    else:
        frames = [synth_planes_torch(W, n_planes=4, seed=12345 + rank * K + j, device=dev) for j in range(K)]   # frame f uses seed 12345+f
        torch.cuda.synchronize()
        planes = frames[0]
        encs = [HipTileEncoder(dev_index) for _ in range(K)]
        for e, f in zip(encs, frames):
            e.set_image(f)
    enc = encs[0]
    # N > 1: every rank encodes its own frame; the ONE collective of the path, the gather of the packed tile maps onto rank 0,
    # is double-buffered so that the RCCL transfer of frame i rides under the encode kernels of frame i+1.
    pipe = ykd.TileMapGatherPipeline(dist, comm_dev, enc.export_capacity(), dst=0, staging_device=dev) if world > 1 else None
    # frames of at least four rounds of strip-waves: the fused kernels of the frames in flight take turns (two of them sharing the chip
    # only slow each other down); smaller frames need each other's waves to fill the chip and stay unordered
    ordered = K > 1 and not BF and not args.graph and not args.free_overlap and W * W >= 8192 * 8192

    def step():
        for j, e in enumerate(encs):           # no host synchronisation in here: K frames are in flight on K streams
            if ordered:
                e.order_fused_after(encs[j - 1])   # alpha / compaction overlap the other frame's fused kernel; the fused kernels take turns
            if BF:
                e.encode_batch(3, args.mode3)
            elif args.graph:
                e.encode_frame(3, args.mode3)
            else:
                e.alpha_reject()
                e.alpha_finish(None)
                e.encode(3, args.mode3, False)
        if world > 1:
            for e in encs:
                blob, _ = pipe.acquire()       # waits for the gather that used this buffer two submits ago
                # one packing kernel behind the encode on the handle's stream writes header + sections; the transfers are enqueued behind
                # torch's current stream, which is made to wait for that kernel on the device: the host never waits for the frame it queued
                e.export_tile_maps_framed(blob, torch.cuda.current_stream(dev).cuda_stream)
                pipe.submit()                  # the ONE collective of the step (per frame): a grouped launch of point-to-point transfers

    def fence():
        torch.cuda.synchronize()
        for e in encs:
            e.synchronize()
        if world > 1:
            pipe.flush()                       # every gather has landed on rank 0 before the clock stops
            dist.barrier()
        torch.cuda.synchronize()

    if world > 1:
        # communicator set-up (RCCL opens its peer connections at the first transfer) never lands in the timed steps, whatever --warmup says
        step()
        step()                                 # twice: the second frame of a buffer is the first one sent at an agreed length
        fence()
    for _ in range(args.warmup):
        step()
    fence()
    kms = {"encode": 0.0, "alpha": 0.0, "pack": 0.0}
    if args.warmup > 0:
        for e in encs:
            e.kernel_ms()                      # drop the warm-up steps' event sets
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()                                 # frames are queued back to back: no host synchronisation inside the timed loop
        if (_ + 1) % 32 == 0 or _ + 1 == args.steps:
            done = (_ % 32) + 1                # steps covered by this query (the handle keeps a ring of 64 event sets)
            for e in encs:
                k = e.kernel_ms()              # HIP events on the launch stream, averaged over the steps since the last query
                for n in kms:
                    kms[n] += k[n] * done / len(encs)
    fence()
    t1 = time.perf_counter()
    elapsed = t1 - t0
    if world > 1:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=comm_dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    for n in kms:
        kms[n] /= max(1, args.steps)

    # ---- N > 1: check the collective's data path outside the timed region: every rank's payload must arrive on rank 0 byte for
    # byte (compared through 64-bit checksums), through the pipeline the timed loop used and through the C-ABI gather (yk_gather_maps)
    gather_check, rccl_ranks, abi_note, abi_hung = None, None, None, False
    if world > 1:
        collectives_per_step = (pipe.collectives - pipe.regathers) / max(1, pipe.step)

        def checksum(t):
            v = t.to(torch.int64)
            idx = torch.arange(1, v.numel() + 1, device=v.device, dtype=torch.int64)
            return int(((v * (idx % 65521 + 1)).sum() + v.numel()).item())
        bad = []
        blob, _ = pipe.acquire()
        enc.export_tile_maps_framed(blob, torch.cuda.current_stream(dev).cuda_stream)
        pipe.submit()
        res = pipe.flush()[-1]
        enc.synchronize()
        need = ykd.HEADER_BYTES + int(blob[:8].cpu().view(torch.int64)[0].item())
        mine = torch.tensor([checksum(blob[ykd.HEADER_BYTES:need]), need], dtype=torch.int64, device=comm_dev)
        allsums = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allsums, mine)
        if rank == 0:
            for r, (sz, payload) in enumerate(res):
                want = allsums[r].cpu().tolist()
                if ykd.HEADER_BYTES + int(sz[14]) != want[1] or checksum(payload.to(dev)) != want[0]:
                    bad.append(f"pipeline: rank {r}")
        if not rehearsal:
            # the timed region is over and the torch.distributed gather is checked; now the same gather through the C-ABI.  It runs in a
            # worker thread with a deadline (ABI_DEADLINE_S): a check that fails, raises or does not return FAILS the run (exit code 1 / 3).
            import threading
            box = {}

            def abi():
                try:
                    box["r"] = abi_gather_check(enc, dist, dev, rank, world, blob, need, allsums, checksum)
                except Exception as ex:          # noqa: BLE001  (a check that raises is a failed check)
                    box["r"] = (False, None, f"C-ABI gather raised {type(ex).__name__}: {ex}")
            th = threading.Thread(target=abi, daemon=True)
            th.start()
            deadline = float(os.environ.get("YK_BENCH_ABI_DEADLINE", str(ABI_DEADLINE_S)))
            th.join(deadline)
            if "r" in box:
                ok, rccl_ranks, abi_note = box["r"]
                if ok == "hung":
                    ok, abi_hung = False, True
            else:
                ok, rccl_ranks, abi_note, abi_hung = False, None, f"the C-ABI gather did not return within its deadline of {deadline:.0f} s", True
            if not ok:
                bad.append(("C-ABI gather HUNG: " if abi_hung else "C-ABI gather: ") + abi_note)
        gather_check = "ok" if not bad else "MISMATCH " + ", ".join(bad)

    if abi_hung:
        # A hung collective is a failed run: the worker thread may still be inside yk_comm_init_rank / yk_gather_maps on this
        # rank's handle, and a handle is not thread-safe, so nothing touches `enc` from here on.  Rank 0 prints what was measured before the
        # hang, flagged, and every rank leaves with a non-zero code (a communicator stuck in its bootstrap cannot be torn down).
        if rank == 0:
            print(json.dumps({"metric": "Mpix/s tile encode (alpha reject + 7 gradient passes + 8x8 range quant), 8K RGBA",
                              "value": round(W * W * world * K * args.steps / 1e6 / elapsed, 1), "unit": "Mpix/s", "n_gpus": world,
                              "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 4),
                              "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "gather_check": gather_check,
                              "error": "the post-timing C-ABI gather check hung: run failed, no roofline reported"}))
            sys.stdout.flush()
        os._exit(3)
    if rank != 0:
        if world > 1:
            dist.destroy_process_group()
        return 0

    mpix = W * W * world * K * args.steps / 1e6
    value = mpix / elapsed

    # ---- roofline of the dominant kernel (yk_encode2_kernel): algorithmic bytes / event-timed duration -------------
    nd_nn = [enc.range_streams(p) for p in range(3)]
    bitmap_bytes = sum(enc.gradient_bitmap(p).size for p in range(7))
    out_bytes = sum(2 * d.size + nb.size for d, nb, _ in nd_nn)
    alg_bytes = 12 * W * W + bitmap_bytes + out_bytes            # 4 B x 3 planes read once + bitmaps + defs + nibbles
    achieved = alg_bytes / (kms["encode"] * 1e-3) / 1e9 if kms["encode"] > 0 else 0.0
    # HBM bytes per launch from the PMC passes of the same command (tools/profile_round.sh); only valid for the default workload
    # HBM bytes per launch come from separate rocprofv3 --pmc passes (tools/profile_round.sh writes profiles/traffic.json with the
    # SHA-256 of the kernel source it measured): reported only while that hash matches the kernel source in this tree, else null
    traffic, traffic_src, valu = None, None, None
    tpath = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "traffic.json")
    if os.path.exists(tpath) and W == 8192 and not args.mode3 and K <= 2 and not BF:
        with open(tpath) as f:
            tj = json.load(f)
        if tj.get("kernel") == "yk_encode2_kernel":
            if tj.get("kernel_source_sha256") == kernel_source_hash():
                traffic, traffic_src = int(tj["hbm_traffic_bytes"]), tj.get("source")
                valu = tj.get("valu_wave_instructions")
            else:
                traffic_src = "stale: profiles/traffic.json was measured on another version of yk_encode2.hip (re-run tools/profile_round.sh)"
    # the measured roof (after the timed region): the better of two hand-written 16-byte streaming kernels of the library (yk_roof.hip: a copy with
    # read + write bytes counted, and a read-only stream); torch's copy_ on uint8, the round-3 denominator, is kept next to it: it understates the roof
    roof_torch = measured_copy_roof(dev)
    roof_copy, roof_read = measured_stream_roof(enc)
    roof = max(roof_copy, roof_read)
    frame_ms = (W * W / 1e6) / (value / world) * 1e3                          # per GPU: what one frame costs at the measured whole-job rate
    frame_bytes = 16 * W * W + W * W // 256 // 8 + bitmap_bytes + out_bytes      # the four int32 planes once + every output of the frame
    # the same kernel with nothing else on the chip (after the timed region, rank 0, plain frames only): in the timed region above the other
    # frame's alpha kernel runs UNDER the fused kernel (that is what hides it) and stretches it by a few per cent
    alone_ms = None
    if world == 1 and not BF and not args.graph:
        e0 = encs[0]
        e0.synchronize()                                     # (its event sets of the timed region were read and dropped above)
        for _ in range(6):
            e0.alpha_reject(); e0.alpha_finish(None); e0.encode(3, args.mode3, False); e0.synchronize()
        alone_ms = e0.kernel_ms()["encode"]
    roofline = {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 4), "peak_measured": round(roof, 1), "frac_of_measured": round(achieved / roof, 4),
                "peak_measured_how": "better of yk_measure_roof's two kernels on this GPU (1 GiB, 16-byte accesses, 8 loads per lane in flight, best of 5)",
                "peak_measured_detail": {"copy_16B_read_plus_write": round(roof_copy, 1), "read_only_16B": round(roof_read, 1), "torch_copy_uint8": round(roof_torch, 1)},
                "frame": {"bytes": int(frame_bytes), "ms": round(frame_ms, 4), "achieved": round(frame_bytes / (frame_ms * 1e-3) / 1e9, 1),
                          "frac": round(frame_bytes / (frame_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4), "frac_of_measured": round(frame_bytes / (frame_ms * 1e-3) / 1e9 / roof, 4),
                          "what": "whole frame per GPU: 16 B/pixel of int32 planes read once + alpha bitmap + 7 bitmaps + defs + nibbles, over ms_per_step / frames per step"},
                "traffic": traffic, "traffic_source": traffic_src,
                "kernel": "yk_encode2_kernel", "kernel_ms": round(kms["encode"], 4), "algorithmic_bytes": int(alg_bytes),
                "other_kernels_ms": {"alpha (16-int copy + yk_alpha_kernel)": round(kms["alpha"], 4), "scan+pack (yk_scan1r_kernel + yk_pack_kernel)": round(kms["pack"], 4)}}
    if alone_ms:
        roofline["kernel_alone"] = {"kernel_ms": round(alone_ms, 4), "frac": round(alg_bytes / (alone_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                                    "what": "yk_encode2_kernel on the same frame with one frame at a time (6 frames after the timed region): its own duration; "
                                            "kernel_ms above is its duration in the timed region, sharing the chip with the other frame's alpha kernel"}
    if valu and kms["encode"] > 0:
        # the bound this kernel actually runs against: VALU issue.  Informational; `frac` above stays the HBM figure of the contract
        roofline["valu_issue"] = {"wave_instructions": int(valu), "cycles_per_instruction": 4.15, "simds": 1024, "clock_ghz": 2.2,
                                  "frac": round(valu * 4.15 / (1024 * kms["encode"] * 1e-3 * 2.2e9), 3), "source": tj.get("valu_source")}
    if BF:
        roofline["note"] = f"--batch {BF}: kernel times and algorithmic bytes are per launch = {BF} frames"
        roofline["algorithmic_bytes"] = int(alg_bytes) * BF
        roofline["achieved"] = round(alg_bytes * BF / (kms["encode"] * 1e-3) / 1e9, 1) if kms["encode"] > 0 else 0.0
        roofline["frac"] = round(roofline["achieved"] / HBM_PEAK_GBS, 4)
    elif args.graph:
        roofline["note"] = "--graph: kernel_ms is the whole frame (alpha stage + fused kernel + compaction replayed as one hipGraph)"
    elif ordered:
        roofline["note"] = (f"{K} frames in flight on {K} streams: the fused kernels take turns (yk_order_fused_after), the alpha / pack stages of one "
                            "frame run under the fused kernel of another; their intervals in other_kernels_ms include the time they share the chip")
    elif K >= 2:
        roofline["note"] = f"{K} frames in flight, unordered: each kernel's duration includes the time it shares the chip with the other frames' kernels"

    result = {
        "metric": "Mpix/s tile encode (alpha reject + 7 gradient passes + 8x8 range quant), 8K RGBA",
        "value": round(value, 1), "unit": "Mpix/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "u8/int32 (+f32 mode-selection sums)", "data": "synthetic (YAIK-synth v1, seed 12345+rank)",
        "config": {"workload": f"{W}x{W} RGBA frame per GPU, full encode: alpha reject bitmap + gradient tiles 16x16..4x4 + 8x8 "
                               f"{'3' if args.mode3 else '4'}-bpp range, inputs resident in HBM",
                   "frames_per_step": world * K, "frames_in_flight_per_gpu": K, "launch": ("one launch per kernel for the whole batch" if BF else "hipGraph per frame" if args.graph else "stream operations"),
                   "parallelism": f"frame-sharded x{world}, one RCCL gather of tile maps" if world > 1 else "single GPU"},
        "roofline": roofline,
    }
    if world > 1:
        result["rccl_ranks"] = rccl_ranks if rccl_ranks is not None else None
        result["collective"] = {"backend": dist.get_backend(), "world_size": dist.get_world_size(), "per_frame": round(collectives_per_step, 3),
                                "repeated_transfers": pipe.regathers, "impl": "torch.distributed.batch_isend_irecv: one grouped launch of ncclSend / ncclRecv per frame, "
                                "counts agreed from the 128-byte header of the same buffer's previous payload (no size exchange)",
                                "c_abi": abi_note}
    if gather_check is not None:
        result["gather_check"] = gather_check
        if gather_check != "ok":
            print(json.dumps(result))
            return 1

    # ---- parity at full size + CPU baseline (rank 0, N = 1 only; the oracle is the checker, never the thing shipped) ------
    if world == 1 and not (args.no_cpu and args.no_parity):
        from oracle import pyoracle
        pyoracle.build()
        cs = args.cpu_size or W
        if cs != W:
            off = (W - cs) // 2
            host = planes[:, off:off + cs, off:off + cs].contiguous().cpu().numpy()
        else:
            host = planes.cpu().numpy()
        ora = pyoracle.OracleEncoder(host)
        c0 = time.perf_counter()
        omip = ora.mip_prefilter()
        ograd = [ora.fitting_quad_smooth(sx, sy) for sx, sy in pyoracle.PASSES]
        obm = [g[1] for g in ograd]
        orng = [ora.dynamic_tile_encode(p, args.mode3)[:3] for p in range(3)]
        c1 = time.perf_counter()
        if not args.no_cpu:
            result["cpu_baseline"] = {"value": round(cs * cs / 1e6 / (c1 - c0), 3), "unit": "Mpix/s", "cores": 1, "kind": "port",
                                      "sample": f"{cs}x{cs} RGBA {'whole frame' if cs == W else 'centred crop'} of the same workload, "
                                                f"oracle/liboracle.so (C restatement pinned against the compiled reference), {c1 - c0:.1f} s"}
            rpath = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "cpu_ratio.json")
            if os.path.exists(rpath):                      # ratio restatement / unmodified reference, measured where the reference exists
                with open(rpath) as f:
                    ratio = float(json.load(f)["port_over_reference"])
                result["cpu_baseline"]["port_over_reference"] = ratio
                result["cpu_baseline"]["reference_equivalent"] = round(result["cpu_baseline"]["value"] / ratio, 3)
                result["cpu_baseline"]["ratio_source"] = "profiles/cpu_ratio.json (tools/measure_cpu_ratio.py, build container, same stages, one thread)"
        if not args.no_parity and cs == W:
            ok = all(np.array_equal(enc.gradient_bitmap(i), obm[i]) for i in range(7))
            for p in range(3):
                d, nb, nn = nd_nn[p]
                ok &= np.array_equal(d, orng[p][0]) and np.array_equal(nb, orng[p][1]) and nn == orng[p][2]
            ga = enc.alpha_result()                         # alpha tile-reject: bounds, kept pixels, tile box and the 1-bit bitmap
            ok &= bool(ga["has_chunk"]) == bool(omip["has_chunk"]) and np.array_equal(ga["bounds"], omip["bounds"]) and int(ga["remaining"]) == int(omip["remaining"])
            if omip["has_chunk"]:
                ok &= np.array_equal(ga["tile_bbox"], omip["tile_bbox"]) and np.array_equal(ga["bitmap"], omip["bitmap"])
            ok &= all(np.array_equal(enc.gradient_corners(i), ograd[i][2]) for i in range(7))      # corner-colour streams of the 7 passes
            result["parity"] = ("bit-exact vs oracle at full size (alpha reject bitmap + bounds, 7 tile bitmaps, 7 corner streams, 3x tile defs, "
                                "3x nibble streams)") if ok else "MISMATCH"
            if not ok:
                print(json.dumps(result))
                return 1
    print(json.dumps(result))
    if world > 1:
        dist.destroy_process_group()
    return 0


if __name__ == "__main__":
    sys.exit(main())
