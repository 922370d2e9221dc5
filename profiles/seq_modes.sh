# measurement aid: the other search modes with pictures in flight (profiles/seq_probe.py), 1080p
mkdir -p gpurun_out/r4j
export GPU_MAX_HW_QUEUES=24
for m in ${MODES:-epzs}; do for d in ${DEPTHS:-1 2 4 8}; do timeout 200 python profiles/seq_probe.py ${NPIC:-32} $d 0 $m 2>&1 | grep -E "depth|launches"; done; done > gpurun_out/r4j/seq_modes3.txt
cat gpurun_out/r4j/seq_modes3.txt
