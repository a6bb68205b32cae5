#!/bin/bash
# N = 2 rehearsal of both multi-rank layouts on the ONE GPU of a dev box: gloo, both ranks on GPU 0 (plumbing only, not a scaling number)
mkdir -p gpurun_out/final
for layout in frames stripes; do
  YK_BENCH_BACKEND=gloo timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --layout $layout --steps 6 --warmup 2 --no-cpu > gpurun_out/final/bench_${layout}_gloo2.json 2> gpurun_out/final/bench_${layout}_gloo2.err; echo "$layout gloo2 rc=$?"; tail -1 gpurun_out/final/bench_${layout}_gloo2.json | cut -c1-700
done
