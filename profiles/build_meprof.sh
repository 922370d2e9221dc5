#!/bin/sh
# builds profiles/microbench/libjmhip_meprof.so: the product objects with me_fast.hip recompiled with -DME_PROF (phase profiler)
set -e
cd "$(dirname "$0")/.."
python -m jm_amd.build >/dev/null
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=default -DME_PROF -c jm_amd/csrc/me_fast.hip -o /tmp/me_fast_prof.o
hipcc --offload-arch=gfx950 -shared -fPIC -o profiles/microbench/libjmhip_meprof.so $(ls jm_amd/build/*.o | grep -v me_fast) /tmp/me_fast_prof.o
echo profiles/microbench/libjmhip_meprof.so
