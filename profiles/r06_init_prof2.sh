# round 6: where configs[2]'s FIRST P picture spends its time in the drop-in encoder (p_frame_ms_hip[0] 135 - 232 ms against 78 for the second)
# usage (GPU box): bash profiles/r06_init_prof2.sh <tag>
O=$PWD/gpurun_out/${1:-init2}; mkdir -p $O
T=$(mktemp -d); cd $T
python - <<E
import sys; sys.path.insert(0, "$GRAFT_REPO_ROOT")
import bench
bench.write_yuv("syn1080p.yuv", 3)
E
cp $GRAFT_REPO_ROOT/tests/golden/q_offset.cfg .
FLAGS="-p InputFile=syn1080p.yuv -p SourceWidth=1920 -p SourceHeight=1080 -p OutputWidth=1920 -p OutputHeight=1080 -p SearchMode=3 -p SearchRange=32 -p NumberReferenceFrames=5 -p LevelIDC=51 -p RDOptimization=0 -p AdaptiveRounding=0 -p SymbolMode=1 -p ProfileIDC=100 -p Transform8x8Mode=1 -p OutputFile=o.264 -p ReconFile=o_rec.yuv -p TraceFile=/dev/null -p FramesToBeEncoded=3"
EXE=$GRAFT_REPO_ROOT/oracle/_ref/lencod_hip.exe
for k in 1 2 3; do
  sleep 1
  JMHIP_ADAPTER_TIMELINE=1 JMHIP_INIT_PROF=1 $EXE -d $GRAFT_REPO_ROOT/tests/golden/jm_baseline.cfg $FLAGS > $O/run$k.out 2> $O/run$k.err
  md5sum o.264 >> $O/run$k.err
done
export TMPDIR=/tmp
rocprofv3 --hip-trace --kernel-trace --output-format csv -d $O/hiptrace -o t -- $EXE -d $GRAFT_REPO_ROOT/tests/golden/jm_baseline.cfg $FLAGS > $O/prof.out 2> $O/prof.err
python - <<E
import csv, glob
rows = []
for f in glob.glob("$O/hiptrace/**/*hip_api_trace.csv", recursive=True):
    rows += [r for r in csv.DictReader(open(f))]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
t0 = int(rows[0]["Start_Timestamp"])
out = open("$O/hip_api_long_calls.txt", "w")
for r in rows:
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
    if d > 1.0:
        out.write("%9.1f ms  +%7.1f ms  %s\n" % ((int(r["Start_Timestamp"]) - t0) / 1e6, d, r["Function"]))
kr = []
for f in glob.glob("$O/hiptrace/**/*kernel_trace.csv", recursive=True):
    kr += [r for r in csv.DictReader(open(f))]
kr.sort(key=lambda r: int(r["Start_Timestamp"]))
for r in kr:
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
    if d > 1.0:
        out.write("kernel %9.1f ms  +%7.1f ms  %s\n" % ((int(r["Start_Timestamp"]) - t0) / 1e6, d, r["Kernel_Name"][:60]))
E
rm -rf $O/hiptrace
grep -h "Frame\|jmhip\|picture\|^0000\|encode_slice_launch" $O/run3.err $O/run3.out | cut -c1-300 | head -60
cat $O/hip_api_long_calls.txt | head -80
