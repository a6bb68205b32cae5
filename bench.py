#!/usr/bin/env python
"""bench.py — headline metric of BASELINE.json: Mpix/s of YAIK tile encode (alpha tile-reject + 7 gradient passes +
8x8 4-bpp range quantiser of 3 planes) on an 8192x8192 RGBA frame of synthetic "YAIK-synth v1" data, inputs resident in HBM.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--size 8192] [--mode3] [--in-flight F] [--no-cpu] [--no-parity]

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


def main() -> int:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--size", type=int, default=8192)
    ap.add_argument("--mode3", action="store_true", help="3-bpp range modes only (DynamicTileEncode mode3BitOnly)")
    ap.add_argument("--in-flight", type=int, default=2, help="frames per GPU kept in flight on separate handles/streams (a step = that many frames per "
                    "GPU).  Default 2: a stream of frames, where the HBM-bound alpha / pack kernels of one frame run under the VALU-bound fused "
                    "kernel of the other (the fused kernels themselves take turns: yk_order_fused_after); 1 = strictly one frame at a time")
    ap.add_argument("--free-overlap", action="store_true", help="with --in-flight > 1: do not order the fused kernels of the frames (they then share the chip)")
    ap.add_argument("--batch", type=int, default=0, help="frames per GPU held by ONE handle and encoded with one launch per kernel (yk_encode_batch): "
                    "the form for batches of small frames (BASELINE config 4: --size 2048 --batch 32); a step = that many frames per GPU")
    ap.add_argument("--graph", action="store_true", help="launch every frame as one replayed hipGraph (yk_encode_frame): for batches of small "
                    "frames, where the ~8 stream operations per frame are what limits the rate; per-kernel times are then one interval")
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU baseline leg")
    ap.add_argument("--no-parity", action="store_true", help="skip the full-size bit-exactness check against the oracle")
    ap.add_argument("--cpu-size", type=int, default=0, help="side of the centred crop timed on the CPU (default: whole frame)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and world != 1:
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

    use_async = world > 1 and not rehearsal    # device-side hand-over to the communicator's stream (needs the payload in HBM)
    deferred = []                              # rehearsal only: (nbytes, sizes) of exports whose gather is launched under the next step

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
            while deferred:
                pipe.submit(*deferred.pop(0))
            for j, e in enumerate(encs):
                blob, _ = pipe.acquire()       # waits for the gather that used this buffer two submits ago
                if use_async and pipe.pad != 0:
                    # one packing kernel behind the encode on the handle's stream; the RCCL ops are enqueued behind torch's current
                    # stream, which is made to wait for that kernel on the device: the host never waits for the frame it just queued
                    e.export_tile_maps_async(blob, pipe.meta_tensor(), torch.cuda.current_stream(dev).cuda_stream)
                    pipe.submit()
                else:
                    sizes = e.export_tile_maps(blob)   # packing kernel + stream sync: the blob is complete on return
                    if use_async or j + 1 < len(encs):
                        pipe.submit(int(sizes[14]), sizes)
                    else:
                        deferred.append((int(sizes[14]), sizes))

    def fence():
        torch.cuda.synchronize()
        for e in encs:
            e.synchronize()
        if world > 1:
            while deferred:
                pipe.submit(*deferred.pop(0))
            pipe.flush()                       # every gather has landed on rank 0 before the clock stops
            dist.barrier()
        torch.cuda.synchronize()

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
    # byte (compared through 64-bit checksums), for the host-fenced export and for the device-side hand-over
    gather_check = None
    if world > 1:
        def checksum(t):
            v = t.to(torch.int64)
            idx = torch.arange(1, v.numel() + 1, device=v.device, dtype=torch.int64)
            return int(((v * (idx % 65521 + 1)).sum() + v.numel()).item())
        bad = []
        for mode in (("host", "device") if use_async else ("host",)):
            blob, _ = pipe.acquire()
            if mode == "host":
                sizes = enc.export_tile_maps(blob)
                pipe.submit(int(sizes[14]), sizes)
            else:
                enc.export_tile_maps_async(blob, pipe.meta_tensor(), torch.cuda.current_stream(dev).cuda_stream)
                pipe.submit()
            res = pipe.flush()[-1]
            ref_sizes = enc.export_tile_maps(blob)                               # blob is free again after the flush
            mine = torch.tensor([checksum(blob[: int(ref_sizes[14])]), int(ref_sizes[14])], dtype=torch.int64, device=comm_dev)
            allsums = [torch.zeros_like(mine) for _ in range(world)]
            dist.all_gather(allsums, mine)
            if rank == 0:
                for r, (sz, payload) in enumerate(res):
                    want = allsums[r].cpu().tolist()
                    if int(sz[14]) != want[1] or checksum(payload.to(dev)) != want[0]:
                        bad.append(f"{mode}: rank {r}")
        gather_check = "ok" if not bad else "MISMATCH " + ", ".join(bad)

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
    traffic, traffic_src = None, None
    tpath = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "traffic.json")
    if os.path.exists(tpath) and W == 8192 and not args.mode3 and K <= 2 and not BF:
        with open(tpath) as f:
            tj = json.load(f)
        if tj.get("kernel") == "yk_encode2_kernel":
            traffic, traffic_src = int(tj["hbm_traffic_bytes"]), tj.get("source")
    roofline = {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic, "traffic_source": traffic_src,
                "kernel": "yk_encode2_kernel", "kernel_ms": round(kms["encode"], 4), "algorithmic_bytes": int(alg_bytes),
                "other_kernels_ms": {"alpha (memset+yk_alpha_kernel+yk_alpha_bbox_kernel)": round(kms["alpha"], 4), "scan+pack (2 kernels)": round(kms["pack"], 4)}}
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
        ora.mip_prefilter()
        obm = [ora.fitting_quad_smooth(sx, sy)[1] for sx, sy in pyoracle.PASSES]
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
            result["parity"] = "bit-exact vs oracle at full size (7 bitmaps, 3x tile defs, 3x nibble streams)" if ok else "MISMATCH"
            if not ok:
                print(json.dumps(result))
                return 1
    print(json.dumps(result))
    if world > 1:
        dist.destroy_process_group()
    return 0


if __name__ == "__main__":
    sys.exit(main())
