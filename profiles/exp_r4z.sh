# measurement aid (round 4): the four-wave EPZS form, pictures in flight x workgroups per picture (profiles/seq_probe.py), 1080p
export GPU_MAX_HW_QUEUES=${QUEUES:-32}
mkdir -p gpurun_out/r4z
SWEEP=${SWEEP:-16:16 20:12 20:13 24:10 32:8}
( for dw in $SWEEP; do timeout 200 python profiles/seq_probe.py ${NPIC:-96} ${dw%:*} ${dw#*:} epzs 2>&1 | grep -E "depth|Error|error"; done
) > gpurun_out/r4z/probe3.txt 2>&1
cat gpurun_out/r4z/probe3.txt
