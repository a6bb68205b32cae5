#!/bin/bash
# whole bench line for several library builds and bench flags: tools/r02_bench_ab.sh <tag> "<flags>" lib...
TAG=$1; FLAGS=$2; shift; shift
mkdir -p gpurun_out/r02
O=gpurun_out/r02/bench_ab_$TAG.log
: > $O
for rep in 1 2; do for v in "$@"; do
  cp $v yaik_amd/libyaik_hip.so
  timeout -k 10 200 python bench.py --steps 40 --warmup 5 --no-cpu --no-parity $FLAGS 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print('$v', d['value'], d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['other_kernels_ms'])" >> $O
done; done
cat $O
