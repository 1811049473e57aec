#!/bin/bash
# round 5, call 21: library builds on one box.
#   touch:    -DMCRT_TRACE_TOUCH (the lean visit asks for the first word of the block it continues with at its end) - C3, C4 probes
#   unroll50: -mllvm -unroll-threshold=50 (renderKernelPM 11 % fewer static instructions), O2: -O2 - C5 probe (1e7 emissions), C3
mkdir -p gpurun_out/r05
L=gpurun_out/r05/ab_builds_touch_flags.log
: > $L
WORKLOAD=c3 SQRTSPP=8 tools/ab_builds.sh base touch unroll50 O2 base touch 2>&1 | sed "s/^/c3 /" | tee -a $L
WORKLOAD=c4 SQRTSPP=4 tools/ab_builds.sh base touch base touch 2>&1 | sed "s/^/c4 /" | tee -a $L
WORKLOAD=c5 SQRTSPP=8 EMISSIONS=1e7 tools/ab_builds.sh base unroll50 O2 base unroll50 O2 2>&1 | sed "s/^/c5 /" | tee -a $L
WORKLOAD=c2 STEPS=3 SQRTSPP=16 tools/ab_builds.sh base unroll50 O2 base 2>&1 | sed "s/^/c2 /" | tee -a $L
