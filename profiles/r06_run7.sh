# round 6: the six-wave form once more, the roles 4..7 balanced over its two waves, on the round's leaner searches
set -x
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$1; rm -rf $O; mkdir -p $O
cd $R
JMHIP_FS_WAVES=6 python bench.py --steps 20 --no-cpu-baseline --no-end-to-end --streams 0 > $O/bench_20_six.json 2> $O/bench_20_six.err
JMHIP_FS_WAVES=6 python bench.py --steps 40 --no-cpu-baseline --no-end-to-end --streams 0 > $O/bench_40_six.json 2> $O/bench_40_six.err
python bench.py --steps 40 --no-cpu-baseline --no-end-to-end --streams 0 > $O/bench_40_eight.json 2> $O/bench_40_eight.err
JMHIP_FS_WAVES=6 python profiles/batch_prof.py 21 fs 1 > $O/batch_prof_six.txt 2>&1
cat $O/batch_prof_six.txt; for f in $O/bench_*.json; do echo $f; head -c 250 $f; echo; done
