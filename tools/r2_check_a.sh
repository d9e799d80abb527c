#!/bin/bash
# round-2 check A: GPU tests, then the bench lines of the new launch / node paths (joined with &&)
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q > gpurun_out/r2a_tests.log 2>&1 && \
python bench.py --steps 20 --warmup 5 > gpurun_out/r2a_bench_short.json 2> gpurun_out/r2a_bench_short.err && \
python bench.py --workload c4 --no-extras --no-cpu > gpurun_out/r2a_c4.json 2> gpurun_out/r2a_c4.err && \
COOLMIC_BENCH_FORCE_NODE=1 python bench.py --workload c5 --no-extras --no-cpu > gpurun_out/r2a_c5_force.json 2> gpurun_out/r2a_c5_force.err && \
COOLMIC_BENCH_REHEARSAL=1 python bench.py --gpus 2 --workload c5 --steps 100 --warmup 20 --no-extras --no-cpu > gpurun_out/r2a_reh_c5_2.json 2> gpurun_out/r2a_reh_c5_2.err && \
COOLMIC_BENCH_REHEARSAL=1 python bench.py --gpus 2 --steps 100 --warmup 20 --no-extras --no-cpu > gpurun_out/r2a_reh_c2_2.json 2> gpurun_out/r2a_reh_c2_2.err
rc=$?
[ $rc -eq 0 ] && python tools/bench_generic.py > gpurun_out/r2a_generic.txt 2>&1
tail -5 gpurun_out/r2a_tests.log
echo "check A rc=$rc"
exit $rc
