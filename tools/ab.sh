#!/bin/bash
# Same-box A/B of library builds, selected through YK_LIB (the product library in the tree is never overwritten):
#   tools/ab.sh <tag> [--pmc] libA.so libB.so ...
# per build: fused-kernel time per content class, the bench frame alone, the bench line (two frames in flight); with --pmc also the
# instruction counters of the fused kernel on the bench frame (rocprofv3 --pmc, its own pass).  Output: gpurun_out/r04/ab_<tag>.log
TAG=$1; shift
PMC=0; if [ "$1" = "--pmc" ]; then PMC=1; shift; fi
mkdir -p gpurun_out/r04
O=gpurun_out/r04/ab_$TAG.log
: > $O
for rep in 1 2; do
for v in "$@"; do
  export YK_LIB=$PWD/$v
  echo "== $v (rep $rep)" >> $O
  timeout -k 10 120 python tools/gpu_class_cost.py 2>&1 | grep "mode3=0" | cut -c1-40 >> $O || exit 1
  timeout -k 10 120 python tools/gpu_class_pmc.py frame 0 2>&1 | grep "fused kernel" >> $O || exit 1
  timeout -k 10 200 python bench.py --steps 40 --warmup 5 --no-cpu --no-parity 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print('bench Gpix/s', round(d['value']/1e3,1), 'ms/step', d['ms_per_step'], 'kernel_ms', d['roofline']['kernel_ms'])" >> $O || exit 1
done
done
if [ $PMC = 1 ]; then
for v in "$@"; do
  export YK_LIB=$PWD/$v
  n=$(basename $v .so)
  timeout -k 10 150 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY --output-format csv -d gpurun_out/r04/pmc_${TAG}_$n -- python3 tools/gpu_class_pmc.py frame 0 > gpurun_out/r04/pmc_${TAG}_$n.log 2>&1 || exit 1
  python3 - "$TAG" "$n" <<'PY' >> $O
import csv, glob, collections, sys
tag, n = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(lambda: collections.defaultdict(float))
for f in glob.glob(f"gpurun_out/r04/pmc_{tag}_{n}/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "yk_encode2" in r["Kernel_Name"]:
            acc[r["Dispatch_Id"]][r["Counter_Name"]] += float(r["Counter_Value"])
if acc:
    d = list(acc.values())[-1]
    w = d.get("SQ_WAVES", 65536.0)
    print("pmc", n, "waves", int(w), {k: round(v / w, 1) for k, v in d.items() if k != "SQ_WAVES"}, "VALU per launch (M)", round(d.get("SQ_INSTS_VALU", 0) / 1e6, 1))
PY
  rm -rf gpurun_out/r04/pmc_${TAG}_$n
done
fi
cat $O
