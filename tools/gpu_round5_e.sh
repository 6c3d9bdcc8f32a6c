#!/bin/bash
# round 5, fifth GPU call: the streamed host-ABI outputs (tests, then A/B of the large-output calls against round 4's
# library on one box), then the whole -m gpu suite
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -q -k "host_abi or closed_loop or datalog" -p no:cacheprovider > gpurun_out/r05_gpu_tests_e.log 2>&1
echo "host abi tests rc=$?"; tail -3 gpurun_out/r05_gpu_tests_e.log
VDYN_LIB_PATH=$PWD/tools/isa/_variants/libvdyn_r04.so timeout -k 10 300 python tools/host_abi_logs.py > gpurun_out/r05_host_abi_logs_r04lib.json 2> gpurun_out/r05_host_abi_logs_r04lib.err; cat gpurun_out/r05_host_abi_logs_r04lib.json
timeout -k 10 300 python tools/host_abi_logs.py > gpurun_out/r05_host_abi_logs.json 2> gpurun_out/r05_host_abi_logs.err; cat gpurun_out/r05_host_abi_logs.json
timeout -k 10 900 python -m pytest tests -m gpu -q -p no:cacheprovider > gpurun_out/r05_gpu_tests_e_all.log 2>&1
echo "all gpu tests rc=$?"; tail -3 gpurun_out/r05_gpu_tests_e_all.log
