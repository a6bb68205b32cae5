#!/bin/bash
# VALU instructions per strip of the fused kernel by phase: ablation 3 = staging + outputs only, 1 = + gradient passes, 0 = everything
TAG=$1; cp $2 yaik_amd/libyaik_hip.so
mkdir -p gpurun_out/r02
for cls in frame mild ramp noise; do for abl in 3 1 0; do
  timeout -k 10 150 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY --output-format csv -d gpurun_out/r02/pmca_${TAG}_${cls}_$abl -- python3 tools/gpu_class_pmc.py $cls $abl > gpurun_out/r02/pmca_${TAG}_${cls}_$abl.log 2>&1
done; done
python3 - "$TAG" <<'PY' | tee gpurun_out/r02/pmca_$TAG.txt
import csv, glob, collections, sys
tag = sys.argv[1]
for cls in ("frame", "mild", "ramp", "noise"):
  for abl in (3, 1, 0):
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    for f in glob.glob(f"gpurun_out/r02/pmca_{tag}_{cls}_{abl}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "yk_encode2" in r["Kernel_Name"]:
                acc[r["Dispatch_Id"]][r["Counter_Name"]] += float(r["Counter_Value"])
    if acc:
        d = list(acc.values())[-1]
        w = d.get("SQ_WAVES", 65536.0)
        t = open(f"gpurun_out/r02/pmca_{tag}_{cls}_{abl}.log").read()
        ms = [l for l in t.splitlines() if "fused kernel" in l]
        print(tag, cls, "ablate", abl, {k: round(v / w, 1) for k, v in d.items()}, ms[-1] if ms else "")
PY
