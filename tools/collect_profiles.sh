#!/bin/bash
# copies what tools/run_round_checks.sh left under gpurun_out/ into profiles/ (names of round $1, default r01)
R=${1:-r02}
set -e
cp gpurun_out/bench_default.json profiles/${R}_bench_default.json
cp gpurun_out/c2_rocprof.json profiles/${R}_c2_bench_under_rocprof.json
cp gpurun_out/c3_rocprof.json profiles/${R}_c3_bench_under_rocprof.json
cp gpurun_out/x6_rocprof.json profiles/${R}_x6_bench_under_rocprof.json
# (gpurun merges into gpurun_out/, so older runs' files may still be there: take the newest)
cp "$(ls -t gpurun_out/prof_c2/*/*_kernel_stats.csv | head -1)" profiles/${R}_c2_kernel_stats.csv
cp "$(ls -t gpurun_out/prof_c3/*/*_kernel_stats.csv | head -1)" profiles/${R}_c3_kernel_stats.csv
cp "$(ls -t gpurun_out/prof_x6/*/*_kernel_stats.csv | head -1)" profiles/${R}_x6_kernel_stats.csv
for w in c2 c3 x6; do
    cp gpurun_out/pmc_$w.json profiles/${R}_${w}_pmc.json
    cp gpurun_out/pmc_$w.json profiles/pmc_$w.json
done
cp gpurun_out/pmc_c2.json profiles/pmc_latest.json
cp gpurun_out/bench_default.json profiles/n1_c2.json
for w in c3 c4 c5; do
    cp gpurun_out/n1_$w.json profiles/n1_$w.json
    cp gpurun_out/n1_$w.json profiles/${R}_${w}_bench_n1.json
done
cp gpurun_out/rehearsal_c5_2ranks.json profiles/${R}_rehearsal_c5_2ranks_one_gpu.json
for t in many_channels eq eq_sections chain; do cp gpurun_out/table_$t.txt profiles/${R}_table_$t.txt; done
tail -25 gpurun_out/eq_sq_counters.txt > profiles/${R}_c3_sq_counters.txt
