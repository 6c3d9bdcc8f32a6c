#!/bin/bash
# the -m gpu suite and the five-rank command (after a change that does not touch csrc/)
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -q -x -p no:cacheprovider > gpurun_out/r05_gpu_tests.log 2>&1
rc=$?; echo "rc=$rc" >> gpurun_out/r05_gpu_tests.log; tail -3 gpurun_out/r05_gpu_tests.log
[ $rc -eq 0 ] || exit $rc
t0=$(date +%s%N)
timeout -k 10 590 python bench.py --gpus 5 --device-map 0,0,0,0,0 --dist-backend gloo --steps 20 --warmup 5 > gpurun_out/r05_five_ranks_one_gpu_bench.json 2> gpurun_out/r05_five_ranks_one_gpu_bench.err
rc=$?
t1=$(date +%s%N)
echo "python bench.py --gpus 5 --device-map 0,0,0,0,0 --dist-backend gloo --steps 20 --warmup 5: rc=$rc, wall $(( (t1 - t0) / 1000000 )) ms (budget of the command: 600 s)" | tee gpurun_out/r05_five_ranks_one_gpu_wall.txt
python -c "
import json
d=json.loads([l for l in open('gpurun_out/r05_five_ranks_one_gpu_bench.json') if l.startswith('{')][0])
print(d['value'], d['attempt'], d['exchange']['kind'], d['exchange']['verified'], d['exchange_ab']['p2p']['verified'], d['strong']['lane']['verified'], 'cpu_baseline' in d)"
