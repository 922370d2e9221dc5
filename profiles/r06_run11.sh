# round 6: the second 4x4 block of every 8x8 block searched ahead by the 8x8 sub-mode's wave (search_phase<SPEC>) against the plain chain (JMHIP_MB_NO_SPEC=1)
set -x
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$1; rm -rf $O; mkdir -p $O
cd $R
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-end-to-end --streams 0 > $O/bench_20_spec.json 2> $O/bench_20_spec.err
JMHIP_MB_NO_SPEC=1 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-end-to-end --streams 0 > $O/bench_20_nospec.json 2> $O/bench_20_nospec.err
python bench.py --steps 40 --no-cpu-baseline --no-end-to-end --streams 0 > $O/bench_40_spec.json 2> $O/bench_40_spec.err
python profiles/batch_prof.py 21 fs 1 > $O/batch_prof_spec.txt 2>&1
timeout 1200 python -m pytest tests/test_gpu_seq.py tests/test_gpu_mbenc.py -x -q -m gpu > $O/pytest_subset.txt 2>&1
tail -3 $O/pytest_subset.txt; cat $O/batch_prof_spec.txt; for f in $O/bench_*.json; do echo $f; head -c 300 $f; echo; tail -c 600 $f | head -c 300; echo; done; tail -3 $O/*.err
