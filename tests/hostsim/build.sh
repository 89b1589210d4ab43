#!/bin/bash
# TEST INFRASTRUCTURE: build the host simulation of the C ABI (planner tests only, no GPU).
set -e
cd "$(dirname "$0")"
g++ -O2 -std=c++17 -fPIC -shared -o libi2v_hostsim.so \
    ../../image-to-video-i2v-attack_amd/csrc/i2v_engine.cpp hostsim_backend.cpp
