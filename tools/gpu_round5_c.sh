#!/bin/bash
# round 5, third GPU call: the whole -m gpu suite on the final library, then tools/gpu_round5_b.sh's measurements
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -q -p no:cacheprovider > gpurun_out/r05_gpu_tests_c.log 2>&1
rc=$?
echo "rc=$rc" >> gpurun_out/r05_gpu_tests_c.log
tail -4 gpurun_out/r05_gpu_tests_c.log
[ $rc -eq 0 ] || [ $rc -eq 1 ] || exit $rc
VDYN_LIB_PATH=$PWD/tools/isa/_variants/libvdyn_r04.so timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r05_ab_r04lib_bench.json 2> gpurun_out/r05_ab_r04lib_bench.err
echo "A (r04 lib) rc=$?"
timeout -k 10 300 python bench.py --steps 20 --warmup 5 > gpurun_out/r05_bench_k20.json 2> gpurun_out/r05_bench_k20.err
echo "B (new lib) rc=$?"
t0=$(date +%s.%N)
timeout -k 10 590 python bench.py --gpus 5 --device-map 0,0,0,0,0 --dist-backend gloo --steps 20 --warmup 5 > gpurun_out/r05_five_ranks_one_gpu_bench.json 2> gpurun_out/r05_five_ranks_one_gpu_bench.err
rc=$?
t1=$(date +%s.%N)
echo "five ranks rc=$rc wall $(echo "$t1 - $t0" | bc) s" | tee gpurun_out/r05_five_ranks_one_gpu_wall.txt
timeout -k 10 600 python tools/host_abi_threads.py 4 8 12 16 > gpurun_out/r05_host_abi_threads.txt 2>&1
cat gpurun_out/r05_host_abi_threads.txt
