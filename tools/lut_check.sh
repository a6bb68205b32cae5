#!/bin/bash
bash tools/gpu_suite.sh tests/test_gpu_lut3d.py tests/test_gpu_host_mirror.py || exit 1
for sz in 2048 8192; do
  timeout -k 10 300 python bench.py --stage lut3d --size $sz --steps 3 --warmup 1 > gpurun_out/r02/bench_stage_lut3d_$sz.json 2> gpurun_out/r02/bench_stage_lut3d_$sz.err; echo "lut3d $sz rc=$?"; python3 -c "
import json; d=json.loads(open('gpurun_out/r02/bench_stage_lut3d_$sz.json').readline()); print(d['value'], d['ms_per_step'], d['roofline']['kernel_ms'])"
done
