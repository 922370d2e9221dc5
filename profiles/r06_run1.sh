# round 6, first GPU call: the round-5 schedule with the searches' sums interleaved (fence_regs / sad_block / rows_at_once): bench at the driver's 20 steps, where the time goes, tests
set -x
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$1; rm -rf $O; mkdir -p $O
cd $R
python bench.py --steps 20 --no-cpu-baseline --no-end-to-end --streams 0 > $O/bench_20.json 2> $O/bench_20.err
python profiles/batch_prof.py 21 fs 1 > $O/batch_prof.txt 2>&1
for m in 1 6 5 7 8; do python profiles/prof_mbpipe.py $m 2>&1 | grep -v amdgpu.ids; done > $O/prof_mbpipe.txt
timeout 900 python -m pytest tests/test_gpu_seq.py tests/test_gpu_mbenc.py -x -q -m gpu > $O/pytest_seq_mbenc.txt 2>&1
tail -3 $O/pytest_seq_mbenc.txt; cat $O/batch_prof.txt; head -c 1500 $O/bench_20.json
