#!/bin/bash
mkdir -p gpurun_out
python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "node" > gpurun_out/r2c_tests.log 2>&1 && \
CMHIP_ROWS_RPT=32 python tools/bench_generic.py > gpurun_out/r2c_generic_rpt32.txt 2>&1 && \
CMHIP_ROWS_RPT=64 python tools/bench_generic.py > gpurun_out/r2c_generic_rpt64.txt 2>&1 && \
python bench.py --workload c4 --no-extras --no-cpu > gpurun_out/r2c_c4.json 2> gpurun_out/r2c_c4.err && \
COOLMIC_BENCH_FORCE_NODE=1 python bench.py --workload c5 --no-extras --no-cpu > gpurun_out/r2c_c5_force.json 2> gpurun_out/r2c_c5_force.err && \
python bench.py --workload c4 --no-extras --no-cpu > gpurun_out/r2c_c4b.json 2> gpurun_out/r2c_c4b.err && \
COOLMIC_BENCH_FORCE_NODE=1 python bench.py --workload c5 --no-extras --no-cpu > gpurun_out/r2c_c5b_force.json 2> gpurun_out/r2c_c5b_force.err
rc=$?
tail -3 gpurun_out/r2c_tests.log
echo "check C rc=$rc"
exit $rc
