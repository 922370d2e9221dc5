# round 5: where the first pictures' time goes in the drop-in encoder (i_frame_ms 334, two pictures 0.6 s)
# usage (GPU box): bash profiles/r05_init_prof.sh <tag>
# 1. lencod_hip.exe on two 1080p pictures, the adapter's timeline; 2. the same under rocprofv3 --hip-trace --stats (HIP API totals: hipMalloc, hipHostMalloc, module load in the first launches)
O=$PWD/gpurun_out/${1:-init}; mkdir -p $O
T=$(mktemp -d); cd $T
python - <<E
import sys; sys.path.insert(0, "$GRAFT_REPO_ROOT")
import bench
bench.write_yuv("syn1080p.yuv", 2)
E
cp $GRAFT_REPO_ROOT/tests/golden/q_offset.cfg .
FLAGS="-p InputFile=syn1080p.yuv -p SourceWidth=1920 -p SourceHeight=1080 -p OutputWidth=1920 -p OutputHeight=1080 -p SearchMode=-1 -p SearchRange=32 -p NumberReferenceFrames=1 -p LevelIDC=51 -p RDOptimization=0 -p AdaptiveRounding=0 -p OutputFile=o.264 -p ReconFile=o_rec.yuv -p TraceFile=/dev/null -p FramesToBeEncoded=2"
EXE=$GRAFT_REPO_ROOT/oracle/_ref/lencod_hip.exe
for k in 1 2 3; do
  sleep 1        # (the driver is still clearing the previous process away for ~130 ms after it has gone: a run started then waits for it inside hipGetDeviceCount)
  t0=$(date +%s.%N)
  JMHIP_ADAPTER_TIMELINE=1 JMHIP_INIT_PROF=1 $EXE -d $GRAFT_REPO_ROOT/tests/golden/jm_baseline.cfg $FLAGS > $O/run$k.out 2> $O/run$k.err
  python -c "import sys,time; print(\"wall %.3f s\" % (time.time() - float(sys.argv[1])))" $t0 >> $O/run$k.err
  md5sum o.264 >> $O/run$k.err
done
export TMPDIR=/tmp
rocprofv3 --hip-trace --stats --output-format csv -d $O/hiptrace -o t -- $EXE -d $GRAFT_REPO_ROOT/tests/golden/jm_baseline.cfg $FLAGS > $O/prof.out 2> $O/prof.err
find $O/hiptrace -name '*hip_api_stats*' -exec cp {} $O/hip_api_stats.csv \;
find $O/hiptrace -name '*hip_api_trace*' -exec sh -c 'head -4000 {} > '$O'/hip_api_trace_head.csv' \;
rm -rf $O/hiptrace
grep -h "jmhip\|wall\|^0000\|Total" $O/run3.err $O/run3.out | cut -c1-400 | head -40
head -25 $O/hip_api_stats.csv
