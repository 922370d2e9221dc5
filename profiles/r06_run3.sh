# round 6: do two six-wave workgroups really share a compute unit?
set -x
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$1; rm -rf $O; mkdir -p $O
cd $R
hipcc --offload-arch=gfx950 -O3 -w profiles/microbench/residency.hip -o /tmp/residency && /tmp/residency > $O/residency.txt 2>&1
JMHIP_DEBUG_GRID=1 JMHIP_FS_WAVES=6 python bench.py --steps 20 --no-cpu-baseline --no-end-to-end --streams 0 > $O/bench_20_six.json 2> $O/bench_20_six.err
cat $O/residency.txt; grep encode_slice_launch $O/bench_20_six.err | sort | uniq -c; head -c 300 $O/bench_20_six.json
