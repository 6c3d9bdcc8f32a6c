#!/bin/bash
# round 5, fourth GPU call: host-ABI tests + numbers on the library with the ramped pipeline, the five-rank command's
# wall time, then the rocprofv3 evidence of the final build (headline: collect.sh; the other kernels: collect_kernels.sh)
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -q -k "host_abi or config3_full or general_chain or safe_redo" -p no:cacheprovider > gpurun_out/r05_gpu_tests_d.log 2>&1
echo "tests rc=$?"; tail -2 gpurun_out/r05_gpu_tests_d.log
timeout -k 10 300 python tools/host_abi_threads.py 8 > gpurun_out/r05_host_abi_threads_ramped.txt 2>&1; cat gpurun_out/r05_host_abi_threads_ramped.txt
t0=$(date +%s%N)
timeout -k 10 590 python bench.py --gpus 5 --device-map 0,0,0,0,0 --dist-backend gloo --steps 20 --warmup 5 > gpurun_out/r05_five_ranks_one_gpu_bench.json 2> gpurun_out/r05_five_ranks_one_gpu_bench.err
rc=$?
t1=$(date +%s%N)
echo "python bench.py --gpus 5 --device-map 0,0,0,0,0 --dist-backend gloo --steps 20 --warmup 5: rc=$rc, wall $(( (t1 - t0) / 1000000 )) ms (budget of the command: 600 s)" | tee gpurun_out/r05_five_ranks_one_gpu_wall.txt
timeout -k 10 400 python bench.py > gpurun_out/r05_bench.json 2> gpurun_out/r05_bench.err
echo "default bench rc=$?"
bash profiles/collect.sh r05 2>&1 | tail -8
bash profiles/collect_kernels.sh r05 2>&1 | tail -14
