#!/bin/bash
# Collects the rocprofv3 evidence bench.py's roofline object is checked against (run on the GPU box via gpurun):
#   pass 1  --kernel-trace --stats            -> per-kernel average duration
#   pass 2+ --pmc ... (own runs, no tracing)  -> SQ counters, FETCH_SIZE, WRITE_SIZE (separate passes, see MI355X_MICROARCH.md)
# usage: tools/profile_round.sh <tag>     (outputs under gpurun_out/prof_<tag>/ and the summary gpurun_out/prof_<tag>_summary.json)
set -e
TAG=${1:-r01}
OUT=gpurun_out/prof_$TAG
mkdir -p gpurun_out
ARGS="--steps 10 --warmup 2 --no-cpu --no-parity"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 bench.py $ARGS > $OUT.bench.json 2> $OUT.stats.err
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VALU --output-format csv -d $OUT/pmc_sq -- python3 bench.py --steps 3 --warmup 1 --no-cpu --no-parity > /dev/null 2> $OUT.pmc1.err
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 bench.py --steps 3 --warmup 1 --no-cpu --no-parity > /dev/null 2> $OUT.pmc2.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 bench.py --steps 3 --warmup 1 --no-cpu --no-parity > /dev/null 2> $OUT.pmc3.err
python3 tools/summarize_prof.py $OUT > ${OUT}_summary.json
cat ${OUT}_summary.json
# bench.py reports roofline.traffic from this file only while the kernel source it was measured on is the one in the tree
python3 - "$TAG" "${OUT}_summary.json" <<'PY' > gpurun_out/traffic_$TAG.json
import hashlib, json, sys
tag, path = sys.argv[1], sys.argv[2]
d = json.load(open(path))["yk_encode2_kernel"]
print(json.dumps({"kernel": "yk_encode2_kernel", "hbm_traffic_bytes": d["hbm_traffic_bytes"], "fetch_bytes_corrected": d["fetch_bytes_corrected"],
                  "write_bytes_corrected": d["write_bytes_corrected"], "avg_ns_rocprof": d["avg_ns"],
                  "kernel_source_sha256": hashlib.sha256(open("yaik_amd/csrc/yk_encode2.hip", "rb").read()).hexdigest(),
                  "workload": "8192x8192 RGBA, bench.py defaults (two frames in flight, fused kernels ordered)",
                  "valu_wave_instructions": int(d["SQ_INSTS_VALU"]),
                  "valu_source": "SQ_INSTS_VALU of the same PMC run; 4.15 cycles per packed-16 / 32-bit integer wave-instruction per SIMD (profiles/r02/op_rate_gfx950.txt), 1024 SIMDs, 2.2 GHz shader clock under this kernel (profiles/r01/j_wave_timeline.txt)",
                  "source": f"profiles/{tag[:3]}/{tag[3:]}_pmc_and_stats_summary.json (tools/profile_round.sh: separate --pmc FETCH_SIZE / WRITE_SIZE passes; FETCH_SIZE x2 on gfx950, KB -> bytes)"}, indent=1))
PY
cat gpurun_out/traffic_$TAG.json
