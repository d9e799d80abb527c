#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_eq.py tests/test_gpu_opus_block.py tests/test_gpu_chain.py tests/test_gpu_group.py -m gpu -x -q > gpurun_out/r2e_tests.log 2>&1 && \
timeout -k 10 300 python bench.py --workload c3 --no-cpu > gpurun_out/r2e_c3.json 2> gpurun_out/r2e_c3.err && \
timeout -k 10 300 python tools/bench_eq.py > gpurun_out/r2e_bench_eq.txt 2>&1
rc=$?
tail -5 gpurun_out/r2e_tests.log
echo "check E rc=$rc"
exit $rc
