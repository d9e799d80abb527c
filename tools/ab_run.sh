#!/bin/bash
# A/B of library builds in one gpurun call: tools/ab_run.sh "<ab_eq.py args>" NAME...   (default = the product build)
# two rounds, so that drift of the box shows
ARGS=$1; shift
L=$PWD/libcoolmic-dsp_amd/lib
for rnd in 1 2; do
    for n in "$@"; do
        if [ "$n" = default ]; then unset COOLMIC_HIP_LIB; else export COOLMIC_HIP_LIB=$L/libcoolmic-dsp-hip-$n.so; fi
        timeout -k 10 120 python3 tools/ab_eq.py $ARGS || exit 1
    done
done
