#!/bin/bash
# Round-end checks on the GPU box: GPU tests, smoke, the default bench line, rocprofv3 kernel stats of
# configs 3 and 2, HBM counters of both (tools/hbm_pmc.sh).  Everything lands under gpurun_out/.
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
cd $R
timeout -k 10 300 bash tools/hbm_pmc.sh c2 > gpurun_out/hbm_c2.log 2>&1 && tail -8 gpurun_out/hbm_c2.log
timeout -k 10 300 bash tools/hbm_pmc.sh c3 > gpurun_out/hbm_c3.log 2>&1 && tail -8 gpurun_out/hbm_c3.log
