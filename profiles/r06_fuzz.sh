# round 6: the fuzzers on the round's last library (new seeds; after the Vp fix)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_fuzz2; mkdir -p $O
timeout 700 python tests/fuzz_dropin.py 600 710000 2>&1 | tail -4 | tee $O/fuzz_dropin1.txt
timeout 400 python tests/fuzz_mbenc.py 300 23000 2>&1 | tail -3 | tee $O/fuzz_mbenc.txt
timeout 300 python tests/fuzz_bslice.py 200 2310000 2>&1 | tail -3 | tee $O/fuzz_bslice.txt
timeout 300 python tests/fuzz_dropin.py 200 3003000 2>&1 | tail -3 | tee $O/fuzz_dropin2.txt
