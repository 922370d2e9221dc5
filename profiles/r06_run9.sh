# round 6: what the 168-register budget costs the EIGHT-wave kernel alone on its compute unit (no second workgroup): spills against issue contention
set -x
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$1; rm -rf $O; mkdir -p $O
cd $R
python bench.py --steps 20 --no-cpu-baseline --no-end-to-end --streams 0 > $O/bench_20.json 2> $O/bench_20.err
python profiles/batch_prof.py 21 fs 1 > $O/batch_prof.txt 2>&1
cat $O/batch_prof.txt; head -c 250 $O/bench_20.json
