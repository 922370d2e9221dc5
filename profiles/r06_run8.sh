set -x
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$1; rm -rf $O; mkdir -p $O
cd $R
timeout 1500 python -m pytest tests/test_gpu_bslice.py tests/test_gpu_seq.py -x -q -m gpu -k "in_flight or stress or starved or one_launch or entries" > $O/pytest.txt 2>&1
tail -5 $O/pytest.txt
bash profiles/r06_init_prof2.sh $1/init2 2>&1 | grep "Frame\|0000\|flight_launch\|entered\|left" | head -30
