#!/bin/bash
# Level 2 of the host-layer sanitizer build: csrc/vdyn_capi.hip AND the four kernel translation units, host halves only
# (hipcc --cuda-host-only: the real HIP headers, no device code), + the stub runtime, under ASan + UBSan.
# Output: tests/hipstub/_build/libvdyn_host_asan.so   (LD_PRELOAD clang's libclang_rt.asan-x86_64.so to load it)
set -e
cd "$(dirname "$0")"
mkdir -p _build
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
CSRC=../../python-motionplanning_amd/csrc
F="--offload-arch=gfx950 --cuda-host-only -O1 -g -std=c++17 -fPIC -fno-fast-math -Wno-unused-command-line-argument -fsanitize=address,undefined -fno-sanitize-recover=undefined -fno-omit-frame-pointer -shared-libasan"
for part in f32_rollout f32_rest f64_rollout f64_rest; do
    $HIPCC $F -c $CSRC/vdyn_kernels_$part.hip -o _build/k_$part.o &
done
$HIPCC $F -DVDYN_BUILD_ID='"hipstub2"' -c $CSRC/vdyn_capi.hip -o _build/capi2.o &
$HIPCC $F -DHIPSTUB_REAL_LAUNCHERS -x hip -c hip_stub.cpp -o _build/stub2.o &
wait
# the host halves reference the device code objects (`__hip_fatbin_<id>`, one per translation unit with kernels) that a
# host-only compile does not produce: define whatever names the objects ask for as empty blobs
nm _build/k_f32_rollout.o _build/k_f32_rest.o _build/k_f64_rollout.o _build/k_f64_rest.o _build/capi2.o _build/stub2.o | awk '$1 == "U" && $2 ~ /^__hip_fatbin_/ {print $2}' | sort -u \
    | awk '{print "const unsigned char " $1 "[8] = {0};"}' > _build/fatbins.c
gcc -c -fPIC _build/fatbins.c -o _build/fatbins.o
/opt/rocm/lib/llvm/bin/clang++ -shared -fPIC -pthread -fsanitize=address,undefined -shared-libasan \
    _build/k_f32_rollout.o _build/k_f32_rest.o _build/k_f64_rollout.o _build/k_f64_rest.o _build/capi2.o _build/stub2.o _build/fatbins.o \
    -o _build/libvdyn_host_asan.so
