#!/bin/bash
# Where the GPU path starts to pay behind the operator API: examples/group_server.c (N sine pipelines on one
# coolmic_group_t: pump a block, read every stream's PCM through its coolmic_iohandle_t, one host thread) by
# stream count and block length, beside one pipeline through coolmic_transform_t -> coolmic_vumeter_t
# (examples/product_chain.c).  The CPU pull chain of the same host is in the bench line (cpu_baseline:
# pull_chain_1024B_one_thread_Msamples_s); this script does not touch the oracle.  -> gpurun_out/group_crossover.txt (profiles/r04_group_crossover.txt)
set -e
R=$PWD
O=gpurun_out/group_crossover.txt
CC="gcc -std=gnu11 -O2 -I include -L libcoolmic-dsp_amd/lib -Wl,-rpath,$R/libcoolmic-dsp_amd/lib"
$CC examples/group_server.c -lcoolmic-dsp-hip -lpthread -o /tmp/group_server
$CC examples/product_chain.c -lcoolmic-dsp-hip -lpthread -o /tmp/product_chain
{
echo "# tools/group_crossover.sh: Msamples/s through the operator API on one host thread, mono 48 kHz sine sources, gain 1000/1000"
echo "# one pipeline, coolmic_transform_t -> coolmic_vumeter_t, 1024-byte pulls (examples/product_chain.c):"
timeout -k 10 120 /tmp/product_chain 8000 | grep "direct gain on"
echo "# N pipelines on one coolmic_group_t (examples/group_server.c N BLOCK ROUNDS 1):"
for block in 512 4096; do
  for n in 1 2 4 8 16 32 64 128 256 1024 4096; do
    rounds=$(( 200000 / n + 16 )); [ $rounds -gt 2000 ] && rounds=2000
    timeout -k 10 120 /tmp/group_server $n $block $rounds 1 | head -1
  done
done
echo "# the CPU pull chain of this host, one thread, 1024-byte pulls: cpu_baseline.pull_chain_1024B_one_thread_Msamples_s of the"
echo "# bench line of the same round (profiles/r04_bench_default.json: 305 Msamples/s on the EPYC 9575F; 16 threads block-at-once: 6 074)"
} > $O 2>&1
cat $O
