#!/bin/bash
# gprof of the host side of lencod_hip (oracle/_ref/lencod_hip_pg.exe) on configs[3] with RDO off (G4r: 3840x2160, 8 slices of 4080 macroblocks), ${FRAMES:-4} pictures
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
T=$(mktemp -d); cd $T
python - <<PY
import sys; sys.path.insert(0, "$ROOT/tests/golden"); import synclip; synclip.syn2160p("syn2160p.yuv", int("${FRAMES:-4}"))
PY
ARGS=""; for kv in InputFile=syn2160p.yuv SourceWidth=3840 SourceHeight=2160 OutputWidth=3840 OutputHeight=2160 SearchMode=-1 SearchRange=32 NumberReferenceFrames=1 LevelIDC=51 RDOptimization=0 AdaptiveRounding=0 SliceMode=1 SliceArgument=4080 OutputFile=o.264 ReconFile=o_rec.yuv TraceFile=/dev/null FramesToBeEncoded=${FRAMES:-4}; do ARGS="$ARGS -p $kv"; done
JMHIP_ADAPTER_TIMELINE=1 $ROOT/oracle/_ref/lencod_hip_pg.exe -d $ROOT/tests/golden/jm_baseline.cfg $ARGS 2> err.txt | grep -E "^\s*[0-9]+\(" | tail -4
grep "jmhip adapter" err.txt | cut -c1-900
gprof -b -p $ROOT/oracle/_ref/lencod_hip_pg.exe gmon.out | head -40
