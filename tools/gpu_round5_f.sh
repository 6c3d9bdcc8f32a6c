#!/bin/bash
# round 5, final GPU call of the build: the whole -m gpu suite, smoke, the default bench line, then the rocprofv3
# evidence of THIS library (headline: collect.sh; the other kernels: collect_kernels.sh) so that every summary carries its id
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -q -p no:cacheprovider > gpurun_out/r05_gpu_tests.log 2>&1
rc=$?; echo "rc=$rc" >> gpurun_out/r05_gpu_tests.log; tail -3 gpurun_out/r05_gpu_tests.log
[ $rc -eq 0 ] || exit $rc
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
bash profiles/collect.sh r05 2>&1 | tail -4
bash profiles/collect_kernels.sh r05 2>&1 | tail -3
timeout -k 10 400 python bench.py > gpurun_out/r05_bench.json 2> gpurun_out/r05_bench.err
echo "default bench rc=$?"; python tools/fmt_bench.py < gpurun_out/r05_bench.json
timeout -k 10 300 python bench.py --steps 20 --warmup 5 > gpurun_out/r05_bench_k20.json 2> gpurun_out/r05_bench_k20.err
echo "k20 bench rc=$?"; python tools/fmt_bench.py < gpurun_out/r05_bench_k20.json
