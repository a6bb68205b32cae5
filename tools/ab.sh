#!/bin/bash
# A/B two builds of the library on the same box, alternating: tools/ab.sh libA libB   (box-to-box variance is ~5 %)
for i in 1 2; do
  for v in "$1" "$2"; do
    cp "$v" yaik_amd/libyaik_hip.so
    echo "== $v"; timeout -k 10 100 python tools/gpu_ablate.py 2>&1 | grep "v2 ablate= 0\|v2 mode3"; timeout -k 10 100 python tools/gpu_class_cost.py 2>&1 | grep "mode3=0"
  done
done
