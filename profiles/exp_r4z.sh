mkdir -p gpurun_out/r4z
timeout 400 python tests/fuzz_mbenc.py 300 900000 > gpurun_out/r4z/fuzz2.txt 2>&1; tail -1 gpurun_out/r4z/fuzz2.txt
timeout 400 python tests/fuzz_dropin.py 270 910000 > gpurun_out/r4z/fuzz_dropin2.txt 2>&1; tail -1 gpurun_out/r4z/fuzz_dropin2.txt
