#!/bin/bash
# phase timeline of the fused kernel's waves (GPU box): build/libT.so = the library compiled with -DYK2_TIMING
cp yaik_amd/libyaik_hip.so build/_keep.so && cp build/libT.so yaik_amd/libyaik_hip.so && timeout -k 10 300 python tools/wave_timeline.py; rc=$?
cp build/_keep.so yaik_amd/libyaik_hip.so; exit $rc
