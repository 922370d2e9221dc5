# round 4, final verification of the library: the collection (profiles/collect5.sh), the GPU suite, fuzz runs against the oracle and against CPU JM
bash profiles/collect5.sh r4v6 > gpurun_out/r4v6_collect.log 2>&1
mkdir -p gpurun_out/r4z
timeout 1200 python -m pytest tests -x -q -m gpu > gpurun_out/r4z/pytest_gpu.txt 2>&1; tail -3 gpurun_out/r4z/pytest_gpu.txt
timeout 400 python tests/fuzz_mbenc.py 200 860000 > gpurun_out/r4z/fuzz.txt 2>&1; tail -1 gpurun_out/r4z/fuzz.txt
timeout 300 python tests/fuzz_dropin.py 150 870000 > gpurun_out/r4z/fuzz_dropin.txt 2>&1; tail -1 gpurun_out/r4z/fuzz_dropin.txt
