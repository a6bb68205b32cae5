#!/bin/bash
# the whole -m gpu suite on the working-tree library: verbose, unbuffered, per-test timeout so that a hang names itself in the log
mkdir -p gpurun_out/r02
export PYTHONUNBUFFERED=1
timeout -k 10 900 python -u -m pytest tests -m gpu -x -v --timeout 150 --timeout-method thread "$@" > gpurun_out/r02/pytest_full.log 2>&1; RC=$?
echo "pytest rc=$RC"; grep -n "^E  \|FAILED\|passed\|failed" gpurun_out/r02/pytest_full.log | tail -15
exit $RC
