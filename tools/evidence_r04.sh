#!/bin/bash
# Round-4 evidence set (run on the GPU box through gpurun): the default bench line, its rocprofv3 kernel stats and PMC passes
# (tools/profile_round.sh), the stage benches, the gloo rehearsals of the N > 1 layouts, instruction counters per content class.
# Everything lands under gpurun_out/r04f/ (copied into profiles/r04/f_* afterwards).
set -e
O=gpurun_out/r04f
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python3 bench.py --steps 100 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err
tools/profile_round.sh r04f > $O/profile_round.log 2>&1
python3 bench.py --stage all --steps 40 --warmup 5 > $O/bench_stage_all.json 2>/dev/null
python3 bench.py --stage corners --steps 20 --warmup 3 > $O/bench_stage_corners.json 2>/dev/null
python3 bench.py --stage range1d --steps 20 --warmup 3 > $O/bench_stage_range1d.json 2>/dev/null
python3 bench.py --stage decode --steps 10 --warmup 2 > $O/bench_stage_decode.json 2>/dev/null
python3 bench.py --stage decode --device-streams --steps 20 --warmup 3 > $O/bench_stage_decode_device.json 2>/dev/null
python3 bench.py --in-flight 1 --steps 40 --warmup 5 --no-cpu --no-parity > $O/bench_one_in_flight.json 2>/dev/null
python3 bench.py --mode3 --steps 40 --warmup 5 --no-cpu --no-parity > $O/bench_mode3.json 2>/dev/null
python3 bench.py --size 2048 --batch 32 --steps 20 --warmup 3 --no-cpu --no-parity > $O/bench_2048_batch32.json 2>/dev/null
YK_BENCH_BACKEND=gloo python3 bench.py --gpus 2 --steps 6 --warmup 2 --no-cpu > $O/bench_frames_gloo2_rehearsal.json 2>/dev/null
YK_BENCH_BACKEND=gloo python3 bench.py --gpus 2 --steps 6 --warmup 2 --no-cpu --layout stripes > $O/bench_stripes_gloo2_rehearsal.json 2>/dev/null
tools/kstats.sh ks_final python3 $PWD/bench.py --steps 10 --warmup 3 --no-cpu --no-parity --in-flight 1 > $O/kernel_stats_one_in_flight.txt 2>&1
cp gpurun_out/r04/ks_final_kernel_stats.csv $O/kernel_stats_one_in_flight.csv
tools/kstats.sh ks_final_dec python3 $PWD/bench.py --stage decode --device-streams --steps 10 --warmup 2 > $O/decode_kernel_stats.txt 2>&1
ls -la $O
