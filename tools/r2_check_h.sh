#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_eq.py tests/test_gpu_opus_block.py tests/test_gpu_chain.py -m gpu -x -q > gpurun_out/r2h_tests.log 2>&1 && \
timeout -k 10 300 python tools/ab_eq_gain.py > gpurun_out/r2h_ab.txt 2>&1
rc=$?
tail -3 gpurun_out/r2h_tests.log; cat gpurun_out/r2h_ab.txt
echo "check H rc=$rc"
exit $rc
