#!/bin/bash
# instruction counters of the fused kernel per content class for library builds selected through YK_LIB: tools/pmc_classes.sh <tag> lib...
TAG=$1; shift
mkdir -p gpurun_out/r04
O=gpurun_out/r04/pmc_classes_$TAG.txt
: > $O
for v in "$@"; do
  export YK_LIB=$PWD/$v
  n=$(basename $v .so)
  for cls in ${CLASSES:-frame mild ramp noise}; do
    timeout -k 10 150 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY --output-format csv -d gpurun_out/r04/pmcc_${TAG}_${n}_$cls -- python3 tools/gpu_class_pmc.py $cls 0 > gpurun_out/r04/pmcc_${TAG}_${n}_$cls.log 2>&1 || exit 1
    python3 - "$TAG" "$n" "$cls" <<'PY' >> $O
import csv, glob, collections, sys
tag, n, cls = sys.argv[1:4]
acc = collections.defaultdict(lambda: collections.defaultdict(float))
for f in glob.glob(f"gpurun_out/r04/pmcc_{tag}_{n}_{cls}/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "yk_encode2" in r["Kernel_Name"]:
            acc[r["Dispatch_Id"]][r["Counter_Name"]] += float(r["Counter_Value"])
if acc:
    d = list(acc.values())[-1]
    w = d.get("SQ_WAVES", 65536.0)
    print(n, cls, {k: round(v / w, 1) for k, v in d.items() if k != "SQ_WAVES"})
PY
    rm -rf gpurun_out/r04/pmcc_${TAG}_${n}_$cls gpurun_out/r04/pmcc_${TAG}_${n}_$cls.log
  done
done
cat $O
