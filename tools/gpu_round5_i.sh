#!/bin/bash
# round 5: shared-table trajectories through the host ABI (tests, A/B against round 4's library), then the final records of
# this library: the -m gpu suite, the rocprofv3 collections, the default bench line
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -q -k "host_abi or two_handles" -p no:cacheprovider > gpurun_out/r05_gpu_tests_i.log 2>&1
echo "host abi tests rc=$?"; tail -3 gpurun_out/r05_gpu_tests_i.log
VDYN_LIB_PATH=$PWD/tools/isa/_variants/libvdyn_r04.so timeout -k 10 300 python tools/host_abi_logs.py > gpurun_out/r05_host_abi_logs_r04lib.json 2> gpurun_out/r05_host_abi_logs_r04lib.err; cat gpurun_out/r05_host_abi_logs_r04lib.json
timeout -k 10 300 python tools/host_abi_logs.py > gpurun_out/r05_host_abi_logs.json 2> gpurun_out/r05_host_abi_logs.err; cat gpurun_out/r05_host_abi_logs.json
bash tools/gpu_round5_f.sh
