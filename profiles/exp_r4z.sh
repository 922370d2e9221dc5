# measurement aid (round 4): EPZS with five references in flight, both forms (two four-wave workgroups' LDS: 81 120 B each); one reference again; the sequence tests; smoke
mkdir -p gpurun_out/r4z
export GPU_MAX_HW_QUEUES=24
( timeout 200 python profiles/seq_probe.py 64 16 0 epzs5; JMHIP_EPZS_WAVES=8 timeout 200 python profiles/seq_probe.py 64 8 0 epzs5; timeout 200 python profiles/seq_probe.py 96 16 0 epzs ) 2>&1 | grep -E "depth|rror" > gpurun_out/r4z/probe_epzs5.txt; cat gpurun_out/r4z/probe_epzs5.txt
unset GPU_MAX_HW_QUEUES
timeout 600 python -m pytest tests/test_gpu_seq.py -x -q -m gpu > gpurun_out/r4z/pytest_seq.txt 2>&1; tail -3 gpurun_out/r4z/pytest_seq.txt
timeout 300 python __graft_entry__.py --smoke 2>&1 | tail -2
