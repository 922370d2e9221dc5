# round 6: the six-wave form (k_mb_pipe6, two workgroups per compute unit) against the eight-wave form
set -x
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$1; rm -rf $O; mkdir -p $O
cd $R
python bench.py --steps 20 --no-cpu-baseline --no-end-to-end --streams 0 > $O/bench_20_six.json 2> $O/bench_20_six.err
JMHIP_FS_WAVES=8 python bench.py --steps 20 --no-cpu-baseline --no-end-to-end --streams 0 > $O/bench_20_eight.json 2> $O/bench_20_eight.err
python bench.py --steps 40 --no-cpu-baseline --no-end-to-end --streams 0 > $O/bench_40_six.json 2> $O/bench_40_six.err
python profiles/batch_prof.py 21 fs 1 > $O/batch_prof_six.txt 2>&1
timeout 900 python -m pytest tests/test_gpu_seq.py -x -q -m gpu > $O/pytest_seq.txt 2>&1
rocprofv3 --list-avail > $O/avail.txt 2>&1
tail -3 $O/pytest_seq.txt; cat $O/batch_prof_six.txt; for f in $O/bench_*.json; do echo $f; head -c 300 $f; echo; done
