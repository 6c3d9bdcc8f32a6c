#!/bin/bash
# the -m gpu suite only (after a change that does not touch csrc/)
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -q -x -p no:cacheprovider > gpurun_out/r05_gpu_tests.log 2>&1
rc=$?; echo "rc=$rc" >> gpurun_out/r05_gpu_tests.log; tail -3 gpurun_out/r05_gpu_tests.log
exit $rc
