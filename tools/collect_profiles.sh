#!/bin/bash
# copies what tools/run_round_checks.sh left under gpurun_out/ into profiles/ (names of round $1)
R=${1:-r04}
set -e
cp gpurun_out/bench_default.json profiles/${R}_bench_default.json
cp gpurun_out/bench_default.json profiles/n1_c2.json
for w in c2 c2ro c2ro_below c2ro_off c3 x6; do
    cp gpurun_out/${w}_rocprof.json profiles/${R}_${w}_bench_under_rocprof.json
    # (gpurun merges into gpurun_out/, so older runs' files may still be there: take the newest)
    cp "$(ls -t gpurun_out/prof_$w/*/*_kernel_stats.csv | head -1)" profiles/${R}_${w}_kernel_stats.csv
    cp gpurun_out/${w}_trace_summary.json profiles/${R}_${w}_trace_summary.json
done
for w in c2 c2ro c3 x6 c2ro_below c2ro_off; do
    cp gpurun_out/pmc_$w.json profiles/${R}_${w}_pmc.json
    cp gpurun_out/pmc_$w.json profiles/pmc_$w.json
done
cp gpurun_out/pmc_c2.json profiles/pmc_latest.json
for w in c2ro c3 c4 c5; do
    cp gpurun_out/n1_$w.json profiles/n1_$w.json
    cp gpurun_out/n1_$w.json profiles/${R}_${w}_bench_n1.json
done
cp gpurun_out/rehearsal_2ranks.json profiles/${R}_rehearsal_2ranks_one_gpu.json
for t in many_channels eq chain; do cp gpurun_out/table_$t.txt profiles/${R}_table_$t.txt; done
