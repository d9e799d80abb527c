#!/bin/bash
# round-2 check B: block-kernel parity tests, then the many-channel kernel table
mkdir -p gpurun_out
python -m pytest tests/test_gpu_parity.py tests/test_gpu_opus_block.py tests/test_gpu_group.py -m gpu -x -q > gpurun_out/r2b_tests.log 2>&1 && \
python tools/bench_generic.py > gpurun_out/r2b_generic.txt 2>&1 && \
python tools/bench_ident_many.py > gpurun_out/r2b_ident.txt 2>&1
rc=$?
tail -5 gpurun_out/r2b_tests.log
echo "check B rc=$rc"
exit $rc
