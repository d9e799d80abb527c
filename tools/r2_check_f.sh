#!/bin/bash
mkdir -p gpurun_out
make -s -C libcoolmic-dsp_amd stamps > gpurun_out/r2f_make.log 2>&1 && \
COOLMIC_HIP_LIB=$PWD/libcoolmic-dsp_amd/lib/libcoolmic-dsp-hip-stamps.so timeout -k 10 300 python tools/eq_stamps.py > gpurun_out/r2f_stamps.txt 2>&1 && \
timeout -k 10 900 bash tools/eq_pmc.sh > gpurun_out/r2f_eq_pmc.txt 2>&1
rc=$?
cat gpurun_out/r2f_stamps.txt
echo "check F rc=$rc"
exit $rc
