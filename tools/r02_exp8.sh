#!/bin/bash
cp build/libN12.so yaik_amd/libyaik_hip.so
bash tools/gpu_suite.sh tests/test_gpu_encode_parity.py tests/test_gpu_fuzz_parity.py tests/test_gpu_baseline_configs.py || exit 1
bash tools/r02_ab.sh n12 build/libN11.so build/libN12.so
cp build/libN12.so yaik_amd/libyaik_hip.so
