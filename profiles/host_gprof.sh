#!/bin/bash
# gprof of the host side of lencod_hip (oracle/_ref/lencod_hip_pg.exe: the reference objects built with -pg) on configs[1], RDO off, 6 pictures
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
T=$(mktemp -d); cd $T
python - <<PY
import sys; sys.path.insert(0, "$ROOT"); import bench; bench.write_yuv("syn1080p.yuv", int("${FRAMES:-6}"))
PY
ARGS=""; for kv in InputFile=syn1080p.yuv SourceWidth=1920 SourceHeight=1080 OutputWidth=1920 OutputHeight=1080 SearchMode=-1 SearchRange=32 NumberReferenceFrames=1 LevelIDC=51 RDOptimization=0 AdaptiveRounding=0 OutputFile=o.264 ReconFile=o_rec.yuv TraceFile=/dev/null FramesToBeEncoded=${FRAMES:-6}; do ARGS="$ARGS -p $kv"; done
$ROOT/oracle/_ref/lencod_hip_pg.exe -d $ROOT/tests/golden/jm_baseline.cfg $ARGS | grep -E "^\s*[0-9]+\(" | tail -4
gprof -b -p $ROOT/oracle/_ref/lencod_hip_pg.exe gmon.out | head -45
