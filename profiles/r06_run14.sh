# round 6: the whole GPU suite and smoke() on the library of the day
set -x
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$1; rm -rf $O; mkdir -p $O
cd $R
timeout 2400 python -m pytest tests -x -q -m gpu > $O/pytest_gpu.txt 2>&1
tail -3 $O/pytest_gpu.txt
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.txt 2>&1; tail -1 $O/smoke.txt
