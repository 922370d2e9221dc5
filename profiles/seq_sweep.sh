# measurement aid: pictures in flight x workgroups per picture (profiles/seq_probe.py), 1080p configs[1]
mkdir -p gpurun_out/r4b
export GPU_MAX_HW_QUEUES=${HWQ:-24}
for cfg in "8 32" "10 25" "12 21" "16 16" "12 20" "16 15"; do set -- $cfg; timeout 120 python profiles/seq_probe.py ${NPIC:-64} $1 $2 2>&1 | grep depth; done > gpurun_out/r4b/seq_sweep3.txt
cat gpurun_out/r4b/seq_sweep3.txt
