cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_fuzz3; mkdir -p $O
timeout 260 python tests/fuzz_bslice.py 180 2320000 2>&1 | tail -2 | tee $O/fuzz_bslice.txt
timeout 300 python tests/fuzz_dropin.py 220 3004000 2>&1 | tail -2 | tee $O/fuzz_dropin_b.txt
timeout 200 python tests/fuzz_dropin.py 120 730000 2>&1 | tail -2 | tee $O/fuzz_dropin.txt
