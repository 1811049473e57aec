#!/bin/bash
# A/B of library BUILDS on one GPU box: tools/_build/lib<X>.so for X in "$@" are copied over csrc/libmcrt_hip.so in turn and the
# same ab_probe line is run with each (fresh boxes differ by ~1 %, so builds are compared within one call).
#   WORKLOAD=c3 SQRTSPP=8 [EMISSIONS=1e7] tools/ab_builds.sh A B A B
cd "$(dirname "$0")/.."
LIB=monte-carlo-ray-tracer_amd/csrc/libmcrt_hip.so
cp $LIB /tmp/lib_orig.so
for x in "$@"; do
  cp tools/_build/lib$x.so $LIB
  echo "build $x: $(timeout 300 python tools/ab_probe.py ${WORKLOAD:-c3} --sqrtspp ${SQRTSPP:-8} --steps ${STEPS:-2} ${EMISSIONS:+--emissions $EMISSIONS} "base:" 2>&1 | tail -1 | cut -c1-200)"
done
cp /tmp/lib_orig.so $LIB
