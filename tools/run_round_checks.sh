#!/bin/bash
# Round-end checks on the GPU box: GPU tests, smoke, the default bench line, rocprofv3 kernel stats of
# configs 3 and 2 and of the six-channel workload x6, HBM counters of the three (tools/hbm_pmc.sh).  Everything lands under gpurun_out/.
set -e
R=$PWD
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1 || { tail -30 gpurun_out/pytest_gpu.log; exit 1; }
tail -3 gpurun_out/pytest_gpu.log
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()"
timeout -k 10 600 python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err
cat gpurun_out/bench_default.json
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof_c3 $R/gpurun_out/prof_c2
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_c3 -- python3 $R/bench.py --workload c3 --steps 100 --warmup 100 --no-cpu --no-extras > $R/gpurun_out/c3_rocprof.json 2> $R/gpurun_out/c3_rocprof.err
cat $R/gpurun_out/c3_rocprof.json
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_c2 -- python3 $R/bench.py --steps 100 --warmup 100 --no-cpu --no-extras > $R/gpurun_out/c2_rocprof.json 2> $R/gpurun_out/c2_rocprof.err
cat $R/gpurun_out/c2_rocprof.json
rm -rf $R/gpurun_out/prof_x6
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_x6 -- python3 $R/bench.py --workload x6 --steps 100 --warmup 100 --no-cpu --no-extras > $R/gpurun_out/x6_rocprof.json 2> $R/gpurun_out/x6_rocprof.err
cat $R/gpurun_out/x6_rocprof.json
cd $R
timeout -k 10 300 bash tools/hbm_pmc.sh c2 > gpurun_out/hbm_c2.log 2>&1 && tail -8 gpurun_out/hbm_c2.log
timeout -k 10 300 bash tools/hbm_pmc.sh c3 > gpurun_out/hbm_c3.log 2>&1 && tail -8 gpurun_out/hbm_c3.log
timeout -k 10 300 bash tools/hbm_pmc.sh x6 > gpurun_out/hbm_x6.log 2>&1 && tail -8 gpurun_out/hbm_x6.log
# the N = 1 lines of the other workloads (config 5 with a one-rank RCCL communicator: all a 1-GPU box holds),
# the 2-rank rehearsal of the self-launching bench on one GPU, and the tables DESIGN.md quotes
timeout -k 10 300 python bench.py --workload c4 --no-extras --no-cpu > gpurun_out/n1_c4.json 2> gpurun_out/n1_c4.err
timeout -k 10 300 env COOLMIC_BENCH_FORCE_NODE=1 python bench.py --workload c5 --no-extras --no-cpu > gpurun_out/n1_c5.json 2> gpurun_out/n1_c5.err
timeout -k 10 300 python bench.py --workload c3 --no-cpu > gpurun_out/n1_c3.json 2> gpurun_out/n1_c3.err
timeout -k 10 300 env COOLMIC_BENCH_REHEARSAL=1 python bench.py --gpus 2 --workload c5 --steps 100 --warmup 20 --no-extras --no-cpu > gpurun_out/rehearsal_c5_2ranks.json 2> gpurun_out/rehearsal_c5_2ranks.err
timeout -k 10 300 python tools/bench_generic.py > gpurun_out/table_many_channels.txt 2>&1
timeout -k 10 300 python tools/bench_eq.py > gpurun_out/table_eq.txt 2>&1
timeout -k 10 300 python tools/bench_eq_sections.py > gpurun_out/table_eq_sections.txt 2>&1
timeout -k 10 120 python tools/bench_chain.py > gpurun_out/table_chain.txt 2>&1
timeout -k 10 900 bash tools/eq_pmc.sh > gpurun_out/eq_sq_counters.txt 2>&1
tail -25 gpurun_out/eq_sq_counters.txt
