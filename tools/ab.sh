#!/bin/bash
# A/B two builds of the library on the same box, alternating: tools/ab.sh libA libB
for i in 1 2 3; do
  for v in "$1" "$2"; do
    cp "$v" yaik_amd/libyaik_hip.so
    echo "== $v"; timeout -k 10 100 python tools/gpu_ablate.py 2>&1 | grep "v2 ablate= 0\|mode3"
  done
done
