#!/bin/bash
# round 2, run 3: parity of the restructured fused kernel (scalar tile masks, two-stream screening, branch-free staging), A/B timing
mkdir -p gpurun_out/r02
timeout -k 10 300 tools/ubench/op_rate > gpurun_out/r02/op_rate.txt 2>&1
tail -14 gpurun_out/r02/op_rate.txt | cut -c1-120
cp build/libN1.so yaik_amd/libyaik_hip.so
timeout -k 10 1500 python -m pytest tests -x -q -m gpu -v > gpurun_out/r02/pytest_full_n1.log 2>&1; echo "pytest rc=$?"; tail -5 gpurun_out/r02/pytest_full_n1.log
O=gpurun_out/r02/exp3.log
: > $O
for v in build/libBase.so build/libN1.so build/libBase.so build/libN1.so; do
  cp $v yaik_amd/libyaik_hip.so
  echo "== $v" >> $O
  timeout -k 10 120 python tools/gpu_class_cost.py 2>&1 | grep "mode3=0" >> $O
  timeout -k 10 120 python tools/gpu_class_pmc.py frame 0 2>&1 | grep "fused kernel" >> $O
  timeout -k 10 200 python bench.py --steps 40 --warmup 5 --no-cpu --no-parity 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print('bench', d['value'], d['ms_per_step'], d['roofline']['kernel_ms'])" >> $O
done
cp build/libN1.so yaik_amd/libyaik_hip.so
cat $O
