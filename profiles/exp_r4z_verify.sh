# round 4, final: the collection (profiles/collect5.sh), the GPU suite, smoke, fuzz runs against the oracle and against CPU JM
bash profiles/collect5.sh r4v7 > gpurun_out/r4v7_collect.log 2>&1
mkdir -p gpurun_out/r4z
timeout 1500 python -m pytest tests -x -q -m gpu > gpurun_out/r4z/pytest_gpu.txt 2>&1; tail -3 gpurun_out/r4z/pytest_gpu.txt
timeout 300 python __graft_entry__.py --smoke 2>&1 | tail -1
timeout 400 python tests/fuzz_mbenc.py 240 880000 > gpurun_out/r4z/fuzz.txt 2>&1; tail -1 gpurun_out/r4z/fuzz.txt
timeout 300 python tests/fuzz_dropin.py 180 890000 > gpurun_out/r4z/fuzz_dropin.txt 2>&1; tail -1 gpurun_out/r4z/fuzz_dropin.txt
