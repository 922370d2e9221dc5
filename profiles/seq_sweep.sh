# measurement aid: pictures in flight x workgroups per picture (profiles/seq_probe.py), 1080p configs[1]
mkdir -p gpurun_out/r4b
export GPU_MAX_HW_QUEUES=16
for cfg in "8 35" "8 32" "7 41" "6 48" "6 44" "5 60" "5 52" "4 80" "4 64"; do set -- $cfg; timeout 120 python profiles/seq_probe.py ${NPIC:-44} $1 $2 2>&1 | grep depth; done > gpurun_out/r4b/seq_sweep2.txt
cat gpurun_out/r4b/seq_sweep2.txt
