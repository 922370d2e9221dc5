#!/bin/bash
# where the host's wall time goes at 2160p (configs[3], RDO off): user / system time, page faults, system calls
ROOT=$(cd "$(dirname "$0")/.." && pwd)
T=$(mktemp -d); cd $T
python - <<PY
import sys; sys.path.insert(0, "$ROOT/tests/golden"); import synclip; synclip.syn2160p("syn2160p.yuv", int("${FRAMES:-4}"))
PY
ARGS=""; for kv in InputFile=syn2160p.yuv SourceWidth=3840 SourceHeight=2160 OutputWidth=3840 OutputHeight=2160 SearchMode=-1 SearchRange=32 NumberReferenceFrames=1 LevelIDC=51 RDOptimization=0 AdaptiveRounding=0 SliceMode=1 SliceArgument=4080 OutputFile=o.264 ReconFile=o_rec.yuv TraceFile=/dev/null FramesToBeEncoded=${FRAMES:-4}; do ARGS="$ARGS -p $kv"; done
which perf strace ltrace gdb valgrind 2>&1 | head
df -h . | tail -1
/usr/bin/time -v $ROOT/oracle/_ref/lencod_hip.exe -d $ROOT/tests/golden/jm_baseline.cfg $ARGS 2> time.txt | grep -E "^\s*[0-9]+\(" | tail -4
grep -E "Elapsed|User time|System time|Minor|Major|Maximum resident|Voluntary|Involuntary" time.txt
if which strace > /dev/null; then strace -c -f -o strace.txt $ROOT/oracle/_ref/lencod_hip.exe -d $ROOT/tests/golden/jm_baseline.cfg $ARGS > /dev/null 2>&1; head -25 strace.txt; fi
