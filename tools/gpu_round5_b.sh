#!/bin/bash
# round 5, second GPU call: the bench multi-rank tests again, A/B of round 4's library and this one on ONE box,
# the five-rank rehearsal (six ranks were refused by the box: "7 processes had the GPU open (limit 6)"), the staging-thread sweep of the host ABI
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_multishard.py -q -s -p no:cacheprovider > gpurun_out/r05_gpu_multishard_b.log 2>&1
echo "multishard rc=$?"; tail -3 gpurun_out/r05_gpu_multishard_b.log
VDYN_LIB_PATH=$PWD/tools/isa/_variants/libvdyn_r04.so timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r05_ab_r04lib_bench.json 2> gpurun_out/r05_ab_r04lib_bench.err
echo "A (r04 lib) rc=$?"
timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r05_ab_newlib_bench.json 2> gpurun_out/r05_ab_newlib_bench.err
echo "B (new lib) rc=$?"
t0=$(date +%s.%N)
timeout -k 10 590 python bench.py --gpus 5 --device-map 0,0,0,0,0 --dist-backend gloo --steps 20 --warmup 5 > gpurun_out/r05_five_ranks_one_gpu_bench.json 2> gpurun_out/r05_five_ranks_one_gpu_bench.err
rc=$?
t1=$(date +%s.%N)
echo "five ranks rc=$rc wall $(echo "$t1 - $t0" | bc) s" | tee gpurun_out/r05_five_ranks_one_gpu_wall.txt
tail -3 gpurun_out/r05_five_ranks_one_gpu_bench.err
timeout -k 10 600 python tools/host_abi_threads.py > gpurun_out/r05_host_abi_threads.txt 2>&1
cat gpurun_out/r05_host_abi_threads.txt
