#!/bin/bash
# What the deferred, randomly ordered HIP stand-in is worth: each ordering dependency of the host layer's pipelines is
# removed from a COPY of csrc/vdyn_capi.hip (sed), the copy is built against the stub under ASan + UBSan, and the driver is
# run under four seeds.  Every removal must make the driver fail (wrong data or a deadlock report); the unmodified copy
# must pass.  usage: tests/hipstub/negative_experiments.sh   (about two minutes; writes nothing into the repo)
set -e
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
W=$(mktemp -d)
LIBASAN=$(gcc -print-file-name=libasan.so)
cp "$ROOT/python-motionplanning_amd/csrc/vdyn_internal.hpp" "$W/"
declare -A EDIT
EDIT[unmodified]='s|^$||'
EDIT[upload:host_overwrites_pinned_chunk_before_its_upload_ran]='s|if (c >= 2) VDYN_PIPE(hipEventSynchronize(h->ev_h2d\[b\]));|;|'
EDIT[upload:device_chunk_refilled_before_its_kernel_ran]='s|if (c >= 2) VDYN_PIPE(hipStreamWaitEvent(h->stage_stream, h->ev_kernel\[b\], 0));|;|'
EDIT[upload:kernel_does_not_wait_for_its_chunk]='s|VDYN_PIPE(hipStreamWaitEvent(h->stream, h->ev_h2d\[b\], 0));|;|'
EDIT[download:copy_does_not_wait_for_the_kernel]='s|VDYN_HIP(h, hipStreamWaitEvent(h->down_stream, h->ev_kernel\[b\], 0));|;|'
EDIT[download:host_copies_out_before_the_download_ran]='s|VDYN_HIP(h, hipEventSynchronize(h->ev_d2h\[b\]));|;|'
EDIT[closed_loop:first_chunk_ignores_the_phase]='s|hn = c == 0 ? hc - (a.phase % a.ctrl_every + a.ctrl_every) % a.ctrl_every : hc;|hn = hc;|'
for name in unmodified "upload:host_overwrites_pinned_chunk_before_its_upload_ran" "upload:device_chunk_refilled_before_its_kernel_ran" \
            "upload:kernel_does_not_wait_for_its_chunk" "download:copy_does_not_wait_for_the_kernel" \
            "download:host_copies_out_before_the_download_ran" "closed_loop:first_chunk_ignores_the_phase"; do
    cp "$ROOT/python-motionplanning_amd/csrc/vdyn_capi.hip" "$W/capi.hip"
    sed -i "${EDIT[$name]}" "$W/capi.hip"
    if [ "$name" != unmodified ] && cmp -s "$W/capi.hip" "$ROOT/python-motionplanning_amd/csrc/vdyn_capi.hip"; then
        echo "$name: the edit did not apply (the source line changed): fix this script"; exit 2
    fi
    g++ -std=c++17 -O1 -g -fPIC -shared -pthread -fsanitize=address,undefined -fno-sanitize-recover=undefined \
        -I"$ROOT/tests/hipstub" -I"$ROOT/python-motionplanning_amd/csrc" -DVDYN_BUILD_ID='"hipstub"' -x c++ "$W/capi.hip" \
        "$ROOT/tests/hipstub/hip_stub.cpp" -o "$W/lib.so"
    fails=0
    for seed in 0 1 2 3; do
        env LD_PRELOAD="$LIBASAN" HIPSTUB_SEED=$seed ASAN_OPTIONS=detect_leaks=0 python3 "$ROOT/tests/_host_layer_driver.py" "$W/lib.so" \
            > "$W/out.log" 2>&1 || fails=$((fails + 1))
    done
    echo "$name: the driver failed under $fails of 4 seeds"
done
rm -rf "$W"
