#!/bin/bash
# TEST INFRASTRUCTURE: build the host simulation of the C ABI (planner tests only, no GPU).
set -e
cd "$(dirname "$0")"
# -mfma (when the CPU has it) turns fmaf() into one instruction instead of a libm call; -ffp-contract=off keeps every other
# expression exactly as written (the backend is compared BIT FOR BIT with the GPU kernels)
FMA=""; grep -q -m1 " fma " /proc/cpuinfo && FMA="-mfma"
g++ -O2 $FMA -ffp-contract=off -std=c++17 -fPIC -shared -o libi2v_hostsim.so \
    ../../image-to-video-i2v-attack_amd/csrc/i2v_engine.cpp hostsim_backend.cpp
