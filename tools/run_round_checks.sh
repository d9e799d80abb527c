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
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_c3 -- python3 $R/bench.py --workload c3 --steps 20 --warmup 3 --no-cpu --no-extras > $R/gpurun_out/c3_rocprof.json 2> $R/gpurun_out/c3_rocprof.err
cat $R/gpurun_out/c3_rocprof.json
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_c2 -- python3 $R/bench.py --steps 20 --warmup 3 --no-cpu --no-extras > $R/gpurun_out/c2_rocprof.json 2> $R/gpurun_out/c2_rocprof.err
cat $R/gpurun_out/c2_rocprof.json
