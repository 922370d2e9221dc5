# round 6: search_phase<SPEC> (the second 4x4 block searched ahead by the Intra4x4 wave) against the plain chain (JMHIP_MB_NO_SPEC=1): stamps with means, longer launches
set -x
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$1; rm -rf $O; mkdir -p $O
cd $R
python profiles/batch_prof.py 21 fs 1 > $O/batch_prof_spec.txt 2>&1
JMHIP_MB_NO_SPEC=1 python profiles/batch_prof.py 21 fs 1 > $O/batch_prof_nospec.txt 2>&1
python profiles/batch_prof.py 41 fs 1 > $O/batch_prof_spec_40.txt 2>&1
JMHIP_MB_NO_SPEC=1 python profiles/batch_prof.py 41 fs 1 > $O/batch_prof_nospec_40.txt 2>&1
for n in 40 80; do
python bench.py --steps $n --no-cpu-baseline --no-end-to-end --streams 0 > $O/bench_${n}_spec.json 2> $O/bench_${n}_spec.err
JMHIP_MB_NO_SPEC=1 python bench.py --steps $n --no-cpu-baseline --no-end-to-end --streams 0 > $O/bench_${n}_nospec.json 2> $O/bench_${n}_nospec.err
done
set +x
for f in $O/batch_prof*.txt; do echo $f; grep -v amdgpu.ids $f; done; for f in $O/bench_*.json; do echo $f; head -c 220 $f | tail -c 120; echo; done
