#!/bin/bash
# The placement measurements DESIGN.md 4.1 quotes, on the GPU box: gpurun_out/placement_*.txt
# (copy them to profiles/ with the round's prefix to keep them).
set -e
mkdir -p gpurun_out
timeout -k 10 400 python tools/placement_slab.py 200 2048 0 > gpurun_out/placement_slab.txt 2>&1
timeout -k 10 600 python tools/placement_map.py > gpurun_out/placement_map.txt 2>&1
timeout -k 10 400 python tools/placement_matrix.py 6 > gpurun_out/placement_matrix.txt 2>&1
timeout -k 10 400 python tools/placement_forms.py 6 > gpurun_out/placement_forms.txt 2>&1
timeout -k 10 400 python tools/placement_forms.py 3 ro > gpurun_out/placement_forms_ro.txt 2>&1
for nw in 4 8 1; do
    echo "CMHIP_FAST_NW=$nw, placement search on"; CMHIP_PLACE=2 CMHIP_FAST_NW=$nw CMHIP_PLACE_DEBUG=1 timeout -k 10 300 python tools/placement_batches.py 6 2 2>&1 | grep -v "candidate [0-9] at"
    echo "CMHIP_FAST_NW=$nw, CMHIP_PLACE=0"; CMHIP_PLACE=0 CMHIP_FAST_NW=$nw timeout -k 10 300 python tools/placement_batches.py 6 2 2>&1
done > gpurun_out/placement_batches.txt
{   # fresh processes in a row, one batch each: the search as a program meets it
    echo "placement search on (default)"; for i in 1 2 3 4 5 6 7 8; do timeout -k 10 120 python tools/placement_batches.py 1 1 2>&1 | tr "\n" " "; echo; done
    echo "CMHIP_PLACE=0"; for i in 1 2 3 4; do CMHIP_PLACE=0 timeout -k 10 120 python tools/placement_batches.py 1 1 2>&1 | tr "\n" " "; echo; done
} > gpurun_out/placement_processes.txt
timeout -k 10 300 python tools/step_overhead.py > gpurun_out/step_overhead.txt 2>&1
tail -n 4 gpurun_out/placement_forms.txt; tail -n 4 gpurun_out/step_overhead.txt
