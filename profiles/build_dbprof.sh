#!/bin/sh
# builds profiles/microbench/libjmhip_dbprof.so: the product objects with deblock_rows.hip recompiled with -DDB_PROF (phase profiler)
set -e
cd "$(dirname "$0")/.."
python -m jm_amd.build >/dev/null
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=default -DDB_PROF -c jm_amd/csrc/deblock_rows.hip -o /tmp/db_prof.o
hipcc --offload-arch=gfx950 -shared -fPIC -o profiles/microbench/libjmhip_dbprof.so $(ls jm_amd/build/*.o | grep -v deblock_rows) /tmp/db_prof.o
echo profiles/microbench/libjmhip_dbprof.so
