#!/bin/bash
mkdir -p gpurun_out
python -m pytest tests/test_gpu_parity.py tests/test_gpu_opus_block.py -m gpu -x -q > gpurun_out/r2d_tests.log 2>&1 && \
python tools/bench_generic.py > gpurun_out/r2d_generic.txt 2>&1 && \
python bench.py --workload c4 --no-extras --no-cpu > gpurun_out/r2d_c4.json 2> gpurun_out/r2d_c4.err && \
COOLMIC_BENCH_FORCE_NODE=1 python bench.py --workload c5 --no-extras --no-cpu > gpurun_out/r2d_c5_force.json 2> gpurun_out/r2d_c5_force.err && \
python bench.py --workload c4 --no-extras --no-cpu > gpurun_out/r2d_c4b.json 2> gpurun_out/r2d_c4b.err && \
COOLMIC_BENCH_FORCE_NODE=1 python bench.py --workload c5 --no-extras --no-cpu > gpurun_out/r2d_c5b_force.json 2> gpurun_out/r2d_c5b_force.err
rc=$?
tail -3 gpurun_out/r2d_tests.log
echo "check D rc=$rc"
exit $rc
