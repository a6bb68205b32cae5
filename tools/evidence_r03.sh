#!/bin/bash
# Round-3 evidence set (run on the GPU box through gpurun): the default bench line, its rocprofv3 kernel stats and PMC passes
# (tools/profile_round.sh), the stage benches, the gloo rehearsals of the N > 1 layouts, instruction counters per content class,
# the wave timeline is collected separately (it needs the -DYK2_TIMING build).  Everything lands under gpurun_out/r03e/.
set -e
O=gpurun_out/r03h
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python3 bench.py --steps 100 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err
tools/profile_round.sh r03h > $O/profile_round.log 2>&1
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
tools/pmc_classes.sh r03h yaik_amd/libyaik_hip.so > /dev/null 2>&1 && cp gpurun_out/r03/pmc_classes_r03h.txt $O/valu_per_strip_by_class.txt
ls -la $O
