#!/bin/bash
# Round-end checks on the GPU box (one gpurun call): GPU tests, smoke, the default bench line, rocprofv3 kernel
# stats of configs 2 and 3, of the read-only leg c2ro in its three gain forms and of the six-channel workload x6,
# HBM counters of them (tools/hbm_pmc.sh), the
# N = 1 lines of the other workloads, the two-rank rehearsal, the tables DESIGN.md quotes.  Everything lands
# under gpurun_out/; tools/collect_profiles.sh copies what is kept into profiles/.
set -e
R=$PWD
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1 || { tail -30 gpurun_out/pytest_gpu.log; exit 1; }
tail -3 gpurun_out/pytest_gpu.log
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()"
timeout -k 10 600 python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err
cat gpurun_out/bench_default.json
cd /tmp && export TMPDIR=/tmp
prof() {    # name, bench arguments...
    local name=$1; shift
    rm -rf $R/gpurun_out/prof_$name
    timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$name -- \
        python3 $R/bench.py "$@" --steps 200 --warmup 20 --no-cpu --no-extras > $R/gpurun_out/${name}_rocprof.json 2> $R/gpurun_out/${name}_rocprof.err
    cat $R/gpurun_out/${name}_rocprof.json
    python3 $R/tools/trace_summary.py $R/gpurun_out/prof_$name $R/gpurun_out/${name}_rocprof.json $R/gpurun_out/${name}_trace_summary.json
}
# (--no-extras: every launch of the kernel in a trace is a step of the bench; the batch's arrays lie where hipMalloc
# first puts them -- the library's default, and what the default line's `value` and `roofline` are measured on)
prof c2
# the north star's literal leg: VU only, 2 B read per sample, the three arithmetic forms of k_run_fast_ro
prof c2ro --workload c2ro
prof c2ro_below --workload c2ro --gain below
prof c2ro_off --workload c2ro --gain off
prof c3 --workload c3
prof x6 --workload x6
cd $R
for w in c2 c2ro c3 x6; do
    timeout -k 10 300 bash tools/hbm_pmc.sh $w > gpurun_out/hbm_$w.log 2>&1 && tail -8 gpurun_out/hbm_$w.log
done
for g in below off; do
    timeout -k 10 300 bash tools/hbm_pmc.sh c2ro $g > gpurun_out/hbm_c2ro_$g.log 2>&1 && tail -8 gpurun_out/hbm_c2ro_$g.log
done
timeout -k 10 300 python bench.py --workload c2ro --no-cpu > gpurun_out/n1_c2ro.json 2> gpurun_out/n1_c2ro.err
timeout -k 10 300 python bench.py --workload c4 --no-extras --no-cpu > gpurun_out/n1_c4.json 2> gpurun_out/n1_c4.err
timeout -k 10 300 env COOLMIC_BENCH_FORCE_NODE=1 python bench.py --workload c5 --no-extras --no-cpu > gpurun_out/n1_c5.json 2> gpurun_out/n1_c5.err
timeout -k 10 300 python bench.py --workload c3 --no-cpu > gpurun_out/n1_c3.json 2> gpurun_out/n1_c3.err
# the driver's multi-GPU command on the one GPU: two ranks share device 0 (not a scaling number); the line carries
# the config-4 / config-5 legs (node_vu)
timeout -k 10 400 env COOLMIC_BENCH_REHEARSAL=1 python bench.py --gpus 2 --steps 100 --warmup 20 --no-extras --no-cpu > gpurun_out/rehearsal_2ranks.json 2> gpurun_out/rehearsal_2ranks.err
cat gpurun_out/rehearsal_2ranks.json
timeout -k 10 300 python tools/bench_generic.py > gpurun_out/table_many_channels.txt 2>&1
timeout -k 10 300 python tools/bench_eq.py > gpurun_out/table_eq.txt 2>&1
gcc -std=gnu11 -O2 -I include examples/product_chain.c -L libcoolmic-dsp_amd/lib -lcoolmic-dsp-hip -lpthread \
    -Wl,-rpath,$R/libcoolmic-dsp_amd/lib -o /tmp/product_chain
timeout -k 10 120 /tmp/product_chain 8000 > gpurun_out/table_chain.txt 2>&1
cat gpurun_out/table_chain.txt
