#!/bin/bash
# round 5, first GPU call: (1) the NEW test of the general chain with SAFE lanes against round 4's library (expected to
# fail in the fp64 k = 12 shared-table instance the ISA audit flagged), (2) the whole -m gpu suite on the new library,
# (3) a K = 20 bench line
mkdir -p gpurun_out
VDYN_LIB_PATH=$PWD/tools/isa/_variants/libvdyn_r04.so timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -q -k "general_chain" -p no:cacheprovider > gpurun_out/r05_r04lib_general_chain.log 2>&1
echo "rc=$?" >> gpurun_out/r05_r04lib_general_chain.log
timeout -k 10 900 python -m pytest tests -m gpu -q -p no:cacheprovider > gpurun_out/r05_gpu_tests_a.log 2>&1
rc=$?
echo "rc=$rc" >> gpurun_out/r05_gpu_tests_a.log
tail -5 gpurun_out/r05_gpu_tests_a.log
[ $rc -eq 0 ] || [ $rc -eq 1 ] || exit $rc
timeout -k 10 300 python bench.py --steps 20 --warmup 5 > gpurun_out/r05_bench_k20_a.json 2> gpurun_out/r05_bench_k20_a.err
echo "bench rc=$?"
python tools/fmt_bench.py < gpurun_out/r05_bench_k20_a.json
