# round 6 (as collect5.sh; the profiled runs without bench.py's own rocprofv3 passes), the bench as it is now (the timed P pictures in ONE launch): bench.py without and with rocprofv3, plus the two PMC passes; usage: bash profiles/collect5.sh <tag>
set -x
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$1; rm -rf $O; mkdir -p $O
cd $R
python bench.py > $O/bench.json 2> $O/bench.err
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-end-to-end --streams 0 > $O/bench_20.json 2> $O/bench_20.err
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o t -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-end-to-end --streams 0 --no-traffic > $O/bench_prof.json 2> $O/prof.err
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -o t -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-end-to-end --streams 0 --no-traffic > /dev/null 2> $O/pmc1.err
timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -o t -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-end-to-end --streams 0 --no-traffic > /dev/null 2> $O/pmc2.err
ls -R $O | head -30; tail -c 600 $O/bench.json
