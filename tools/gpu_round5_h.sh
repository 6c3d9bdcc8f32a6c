#!/bin/bash
# round 5: records of the final code -- the default bench line, the same-box pair (round 4's library / this one), the
# five-rank command with its wall time
mkdir -p gpurun_out
timeout -k 10 400 python bench.py > gpurun_out/r05_bench.json 2> gpurun_out/r05_bench.err
echo "default bench rc=$?"; python tools/fmt_bench.py < gpurun_out/r05_bench.json
VDYN_LIB_PATH=$PWD/tools/isa/_variants/libvdyn_r04.so timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r05_same_box_round4_library_bench_k20.json 2> gpurun_out/r05_ab_r04.err
echo "A rc=$?"
timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r05_same_box_this_library_bench_k20.json 2> gpurun_out/r05_ab_new.err
echo "B rc=$?"
t0=$(date +%s%N)
timeout -k 10 590 python bench.py --gpus 5 --device-map 0,0,0,0,0 --dist-backend gloo --steps 20 --warmup 5 > gpurun_out/r05_five_ranks_one_gpu_bench.json 2> gpurun_out/r05_five_ranks_one_gpu_bench.err
rc=$?
t1=$(date +%s%N)
echo "python bench.py --gpus 5 --device-map 0,0,0,0,0 --dist-backend gloo --steps 20 --warmup 5: rc=$rc, wall $(( (t1 - t0) / 1000000 )) ms (budget of the command: 600 s)" | tee gpurun_out/r05_five_ranks_one_gpu_wall.txt
