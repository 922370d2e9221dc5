# round 6: the library with the 16x16 mode coded ahead -- the whole GPU suite, smoke, the fuzzers (new seeds)
set -x
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$1; rm -rf $O; mkdir -p $O
cd $R
timeout 1500 python -m pytest tests -q -m gpu > $O/pytest_gpu.txt 2>&1
tail -3 $O/pytest_gpu.txt
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.txt 2>&1; tail -1 $O/smoke.txt
timeout 300 python tests/fuzz_mbenc.py 200 26000 2>&1 | tail -2 | tee $O/fuzz_mbenc.txt
timeout 300 python tests/fuzz_dropin.py 200 720000 2>&1 | tail -3 | tee $O/fuzz_dropin.txt
