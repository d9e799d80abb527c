# The round's build against round 3's on the SAME array pairs (tools/placement_forms.py with another build named).
# The other build is not in the tree; make it first, in the container:
#   git worktree add /tmp/r03 ee2608b && make -C /tmp/r03/libcoolmic-dsp_amd -j8 &&
#   cp /tmp/r03/libcoolmic-dsp_amd/lib/libcoolmic-dsp-hip.so libcoolmic-dsp_amd/lib/libcoolmic-dsp-hip-r03.so
# (built libraries travel to the GPU box; the copy is git-ignored).  -> gpurun_out/r04_same_arrays_r03_vs_r04.txt
set -e
L=libcoolmic-dsp_amd/lib
O=gpurun_out/r04_same_arrays_r03_vs_r04.txt
rm -f $O
echo "== config 2 shape, PCM + VU, same 16 array pairs (last line: round-3 build)" >> $O
timeout -k 10 300 python tools/placement_forms.py 4 pcm $L/libcoolmic-dsp-hip-r03.so >> $O 2>&1
echo "== config 4 shape (8192 mono), PCM + VU" >> $O
PF_SHAPE=8192,1,65536 timeout -k 10 300 python tools/placement_forms.py 4 pcm $L/libcoolmic-dsp-hip-r03.so >> $O 2>&1
echo "== config 2 shape, VU only (read-only), general gain {750,1250}/1000 + swap" >> $O
timeout -k 10 300 python tools/placement_forms.py 4 ro $L/libcoolmic-dsp-hip-r03.so >> $O 2>&1
echo "== 8192 mono, VU only, gain 900/1000 (below scale)" >> $O
PF_SHAPE=8192,1,65536 timeout -k 10 300 python tools/placement_forms.py 4 ro $L/libcoolmic-dsp-hip-r03.so >> $O 2>&1
cat $O
