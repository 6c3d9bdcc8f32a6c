#!/bin/bash
# The library's own host layer (csrc/vdyn_capi.hip, compiled as plain C++) + the test-only HIP runtime / launcher
# stand-ins, under AddressSanitizer + UndefinedBehaviorSanitizer.  Output: tests/hipstub/_build/libvdyn_capi_asan.so
set -e
cd "$(dirname "$0")"
mkdir -p _build
g++ -std=c++17 -O1 -g -fPIC -shared -pthread -Wall -Wextra -Wno-unused-parameter \
    -fsanitize=address,undefined -fno-sanitize-recover=undefined -fno-omit-frame-pointer \
    -I. -DVDYN_BUILD_ID='"hipstub"' -x c++ ../../python-motionplanning_amd/csrc/vdyn_capi.hip hip_stub.cpp \
    -o _build/libvdyn_capi_asan.so
# ThreadSanitizer build of the same two files (the staging copies run on worker threads: CopyPool, vdyn_capi.hip)
if [ "$1" = "tsan" ]; then
    g++ -std=c++17 -O1 -g -fPIC -shared -pthread -fsanitize=thread -fno-omit-frame-pointer \
        -I. -DVDYN_BUILD_ID='"hipstub"' -x c++ ../../python-motionplanning_amd/csrc/vdyn_capi.hip hip_stub.cpp \
        -o _build/libvdyn_capi_tsan.so
fi
