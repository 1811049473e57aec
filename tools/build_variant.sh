#!/bin/bash
# A library VARIANT for tools/ab_builds.sh: csrc/mcrt_hip.hip recompiled with extra flags, linked with the current objects of the
# other translation units into tools/_build/lib<NAME>.so.   tools/build_variant.sh NAME [-DFOO=1 ...]
cd "$(dirname "$0")/.." || exit 1
NAME=$1; shift
C=monte-carlo-ray-tracer_amd/csrc; B=tools/_build; mkdir -p $B
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC"
if [[ " $* " == *" -ffp-contract=fast "* ]]; then FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC"; fi
hipcc $FLAGS "$@" -c $C/mcrt_hip.hip -o $B/mcrt_hip_$NAME.o 2> $B/$NAME.err || { tail -20 $B/$NAME.err; exit 1; }
OTHERS=$(ls $C/_obj/*.o | grep -v "mcrt_hip\.hip\.")
hipcc --offload-arch=gfx950 -shared -fPIC -o $B/lib$NAME.so $B/mcrt_hip_$NAME.o $OTHERS && echo "built $B/lib$NAME.so"
