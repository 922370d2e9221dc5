# round 6: k_mb_pipe without the time stamps' tests, reductions with fused DPP steps, the small blocks' searches leave after step 0 when nothing else can win
set -x
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$1; rm -rf $O; mkdir -p $O
cd $R
timeout 240 python bench.py --steps 20 --no-cpu-baseline --no-end-to-end --streams 0 > $O/bench_20.json 2> $O/bench_20.err
timeout 240 python bench.py --steps 20 --no-cpu-baseline --no-end-to-end --streams 0 > $O/bench_20b.json 2> $O/bench_20b.err
timeout 180 python profiles/batch_prof.py 21 fs 1 > $O/batch_prof.txt 2>&1
(timeout 120 python profiles/prof_mbpipe.py 1) 2>&1 | grep -v amdgpu.ids > $O/prof_mbpipe.txt
timeout 1200 python -m pytest tests/test_gpu_seq.py tests/test_gpu_mbenc.py tests/test_gpu_bslice.py -x -q -m gpu > $O/pytest.txt 2>&1
tail -3 $O/pytest.txt; cat $O/batch_prof.txt; grep "first 4x4\|phase 0 per" -A1 $O/prof_mbpipe.txt | tail -8; for f in $O/bench_20*.json; do head -c 250 $f; echo; done
