#!/bin/bash
# what the driver does at round end, on the final tree: the -m gpu suite, smoke(), the default bench line
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -q -x -p no:cacheprovider > gpurun_out/r05_gpu_tests.log 2>&1
rc=$?; echo "rc=$rc" >> gpurun_out/r05_gpu_tests.log; tail -3 gpurun_out/r05_gpu_tests.log
[ $rc -eq 0 ] || exit $rc
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout -k 10 400 python bench.py > gpurun_out/r05_bench.json 2> gpurun_out/r05_bench.err
echo "default bench rc=$?"; python tools/fmt_bench.py < gpurun_out/r05_bench.json
