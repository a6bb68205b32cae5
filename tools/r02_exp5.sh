#!/bin/bash
# full GPU suite on the working-tree library (verbose, unbuffered, per-test timeout so that a hang names itself), then same-box A/B of
# the committed kernel (N5) vs the working tree (N10)
mkdir -p gpurun_out/r02
cp build/libN10.so yaik_amd/libyaik_hip.so
export PYTHONUNBUFFERED=1
timeout -k 10 900 python -u -m pytest tests -m gpu -x -v --timeout 150 --timeout-method thread > gpurun_out/r02/pytest_full_n10.log 2>&1; RC=$?
echo "pytest rc=$RC"; tail -5 gpurun_out/r02/pytest_full_n10.log
[ $RC -eq 0 ] || exit $RC
bash tools/r02_ab.sh n10 build/libN5.so build/libN10.so
cp build/libN10.so yaik_amd/libyaik_hip.so
