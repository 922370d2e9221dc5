# round 4: the epzs8 probe (configs[2]'s kernel, both forms), the sequence tests (pictures of several slices in flight), the drop-in with SliceMode 1
mkdir -p gpurun_out/r4z
export GPU_MAX_HW_QUEUES=24
( timeout 200 python profiles/seq_probe.py 96 16 0 epzs8; JMHIP_EPZS_WAVES=8 timeout 200 python profiles/seq_probe.py 96 8 0 epzs8 ) 2>&1 | grep -E "depth|rror" > gpurun_out/r4z/probe_epzs8.txt; cat gpurun_out/r4z/probe_epzs8.txt
unset GPU_MAX_HW_QUEUES
timeout 600 python -m pytest tests/test_gpu_seq.py -x -q -m gpu > gpurun_out/r4z/pytest_seq.txt 2>&1; tail -5 gpurun_out/r4z/pytest_seq.txt
timeout 900 python -m pytest tests/test_lencod_dropin.py -x -q -m gpu -k "2160 or slice or flight or configs3" > gpurun_out/r4z/pytest_dropin.txt 2>&1; tail -5 gpurun_out/r4z/pytest_dropin.txt
