#!/bin/bash
# arbitrary counters of the fused kernel on the bench frame: tools/pmc_any.sh <tag> <lib.so> "<CTR CTR ...>" ["<CTR ...>" ...]   (one rocprofv3 --pmc pass per group)
TAG=$1; LIB=$2; shift; shift
mkdir -p gpurun_out/r04
export YK_LIB=$PWD/$LIB
O=gpurun_out/r04/pmc_any_$TAG.txt
: > $O
i=0
for grp in "$@"; do
  i=$((i+1))
  timeout -k 10 150 rocprofv3 --pmc $grp --output-format csv -d gpurun_out/r04/pmca_${TAG}_$i -- python3 tools/gpu_class_pmc.py frame 0 > gpurun_out/r04/pmca_${TAG}_$i.log 2>&1 || { echo "rocprofv3 pass $i ($grp) FAILED:"; tail -25 gpurun_out/r04/pmca_${TAG}_$i.log; exit 1; }
  python3 - "$TAG" "$i" <<'PY' >> $O
import csv, glob, collections, sys
tag, i = sys.argv[1:3]
acc = collections.defaultdict(lambda: collections.defaultdict(float))
for f in glob.glob(f"gpurun_out/r04/pmca_{tag}_{i}/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "yk_encode2" in r["Kernel_Name"]:
            acc[r["Dispatch_Id"]][r["Counter_Name"]] += float(r["Counter_Value"])
if acc:
    d = list(acc.values())[-1]
    print({k: round(v, 1) for k, v in d.items()})
PY
  rm -rf gpurun_out/r04/pmca_${TAG}_$i gpurun_out/r04/pmca_${TAG}_$i.log
done
cat $O
