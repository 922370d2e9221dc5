# round 6: the 16x16 mode coded ahead (PRE) and the next windows staged ahead (ES) against PRE alone (JMHIP_MB_NO_ES=1)
set -x
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$1; rm -rf $O; mkdir -p $O
cd $R
for n in 20 40; do
python bench.py --steps $n --warmup 5 --no-cpu-baseline --no-end-to-end --streams 0 --no-traffic > $O/bench_${n}_es.json 2> $O/bench_${n}_es.err
JMHIP_MB_NO_ES=1 python bench.py --steps $n --warmup 5 --no-cpu-baseline --no-end-to-end --streams 0 --no-traffic > $O/bench_${n}_noes.json 2> $O/bench_${n}_noes.err
done
python profiles/batch_prof.py 21 fs 1 > $O/batch_prof_es.txt 2>&1
JMHIP_MB_NO_ES=1 python profiles/batch_prof.py 21 fs 1 > $O/batch_prof_noes.txt 2>&1
timeout 1200 python -m pytest tests/test_gpu_seq.py tests/test_gpu_mbenc.py -x -q -m gpu > $O/pytest_subset.txt 2>&1
set +x
tail -3 $O/pytest_subset.txt; for f in $O/batch_prof*.txt; do echo $f; grep -v amdgpu.ids $f | head -3; grep "31->16\|8->12\|8->13\|8->14" $f; done
for f in $O/bench_*.json; do echo $f; python -c "
import json,sys
d=json.loads(open('$f').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['roofline']['avg_kernel_ms'], d['config']['records_equal_jm'], d['config']['records_equal_picture_after_picture'], d['config']['mb_types_pskip_16x16_16x8_8x16_p8x8_i4_i16'])"; done
