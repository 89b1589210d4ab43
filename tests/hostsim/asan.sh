#!/bin/bash
# TEST INFRASTRUCTURE: run the planner tests against an ASAN+UBSAN build of the engine + host backend
# (GPU sanitizers are not available on the MI355X pool; the planner/executor C++ is the same source).
set -e
cd "$(dirname "$0")"
cp libi2v_hostsim.so /tmp/libi2v_hostsim.keep 2>/dev/null || true
g++ -O1 -g -ffp-contract=off -std=c++17 -fPIC -shared -fsanitize=address,undefined -fno-omit-frame-pointer -o libi2v_hostsim.so \
    ../../image-to-video-i2v-attack_amd/csrc/i2v_engine.cpp hostsim_backend.cpp
cd ../..
LD_PRELOAD=$(gcc -print-file-name=libasan.so):$(gcc -print-file-name=libubsan.so) ASAN_OPTIONS=detect_leaks=0 \
    python -m pytest tests/test_planner_hostsim.py tests/test_video_hostsim.py tests/test_video_ilaf.py tests/test_native_classifier.py tests/test_video_attacks.py tests/test_pil_resample.py tests/test_sign_family.py -x -q -m "not gpu"
cp /tmp/libi2v_hostsim.keep tests/hostsim/libi2v_hostsim.so 2>/dev/null || tests/hostsim/build.sh
