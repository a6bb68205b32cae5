#!/bin/bash
# instruction counters of the fused kernel per content class for one library build: tools/r02_pmc.sh <tag> <lib>
TAG=$1; cp $2 yaik_amd/libyaik_hip.so
mkdir -p gpurun_out/r02
for cls in frame mild ramp noise; do
  timeout -k 10 150 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY --output-format csv -d gpurun_out/r02/pmc_${TAG}_$cls -- python3 tools/gpu_class_pmc.py $cls 0 > gpurun_out/r02/pmc_${TAG}_$cls.log 2>&1
done
python3 - "$TAG" <<'PY' | tee gpurun_out/r02/pmc_$TAG.txt
import csv, glob, collections, sys
tag = sys.argv[1]
for cls in ("frame", "mild", "ramp", "noise"):
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    for f in glob.glob(f"gpurun_out/r02/pmc_{tag}_{cls}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "yk_encode2" in r["Kernel_Name"]:
                acc[r["Dispatch_Id"]][r["Counter_Name"]] += float(r["Counter_Value"])
    if acc:
        d = list(acc.values())[-1]
        w = d.get("SQ_WAVES", 65536.0)
        print(tag, cls, {k: round(v / w, 1) for k, v in d.items()})
PY
