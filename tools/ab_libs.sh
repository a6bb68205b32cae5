#!/bin/bash
# same-box A/B of library builds: tools/ab_libs.sh <tag> libA libB ...   (per-class fused-kernel times, the bench frame, the bench line)
TAG=$1; shift
mkdir -p gpurun_out/r02
O=gpurun_out/r02/ab_$TAG.log
: > $O
for rep in 1 2; do
for v in "$@"; do
  cp $v yaik_amd/libyaik_hip.so
  echo "== $v" >> $O
  timeout -k 10 120 python tools/gpu_class_cost.py 2>&1 | grep "mode3=0" >> $O
  timeout -k 10 120 python tools/gpu_class_pmc.py frame 0 2>&1 | grep "fused kernel" >> $O
  timeout -k 10 200 python bench.py --steps 40 --warmup 5 --no-cpu --no-parity 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print('bench', d['value'], d['ms_per_step'], d['roofline']['kernel_ms'])" >> $O
done
done
cat $O
