#!/bin/bash
# EQ kernel iteration: parity tests, kernel times, per-role stamps
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_eq.py tests/test_gpu_opus_block.py tests/test_gpu_chain.py -m gpu -x -q > gpurun_out/r2g_tests.log 2>&1 && \
timeout -k 10 300 python tools/bench_eq.py > gpurun_out/r2g_bench_eq.txt 2>&1 && \
make -s -C libcoolmic-dsp_amd stamps > gpurun_out/r2g_make.log 2>&1 && \
COOLMIC_HIP_LIB=$PWD/libcoolmic-dsp_amd/lib/libcoolmic-dsp-hip-stamps.so timeout -k 10 300 python tools/eq_stamps.py > gpurun_out/r2g_stamps.txt 2>&1 && \
timeout -k 10 120 python tools/bench_chain.py > gpurun_out/r2g_chain.txt 2>&1
rc=$?
tail -3 gpurun_out/r2g_tests.log; head -4 gpurun_out/r2g_bench_eq.txt; cat gpurun_out/r2g_stamps.txt gpurun_out/r2g_chain.txt
echo "check G rc=$rc"
exit $rc
