# round 4, after the four-wave EPZS form: the collection (profiles/collect5.sh), a fuzz run against the oracle, fuzz of the drop-in, the epzs8 probe
bash profiles/collect5.sh r4v6 > gpurun_out/r4v6_collect.log 2>&1
mkdir -p gpurun_out/r4z
timeout 400 python tests/fuzz_mbenc.py 240 850000 > gpurun_out/r4z/fuzz.txt 2>&1; tail -2 gpurun_out/r4z/fuzz.txt
export GPU_MAX_HW_QUEUES=24
( timeout 200 python profiles/seq_probe.py 64 16 0 epzs8; JMHIP_EPZS_WAVES=8 timeout 200 python profiles/seq_probe.py 64 8 0 epzs8 ) 2>&1 | grep -E "depth|rror" > gpurun_out/r4z/probe_epzs8.txt; cat gpurun_out/r4z/probe_epzs8.txt
