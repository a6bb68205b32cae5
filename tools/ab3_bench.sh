#!/bin/bash
# three builds on the same box: whole bench step and the fused kernel alone
for i in 1 2; do for v in build/libA.so build/libB.so build/libC.so; do cp $v yaik_amd/libyaik_hip.so; echo "== $v"; timeout -k 10 200 python bench.py --steps 100 --warmup 5 --no-cpu --no-parity 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step'], d['roofline']['kernel_ms'])"; timeout -k 10 100 python tools/gpu_ablate.py 2>&1 | grep "v2 ablate= 0"; done; done
cp build/libA.so yaik_amd/libyaik_hip.so
