# round 6: the fuzzers on the round's last library (new seeds)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_fuzz; mkdir -p $O
timeout 400 python tests/fuzz_mbenc.py 300 20000 2>&1 | tail -3 | tee $O/fuzz_mbenc.txt
timeout 340 python tests/fuzz_bslice.py 240 2300000 2>&1 | tail -3 | tee $O/fuzz_bslice.txt
timeout 400 python tests/fuzz_dropin.py 300 700000 2>&1 | tail -3 | tee $O/fuzz_dropin1.txt
timeout 340 python tests/fuzz_dropin.py 240 3002000 2>&1 | tail -3 | tee $O/fuzz_dropin2.txt
