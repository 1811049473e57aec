#!/bin/bash
# round 5, call 22: library builds on one box, second batch.
#   nobreak:  the two compaction loops of histSelectK without their early exit (they were the loops hipcc reports as "not unrolled")
#   maxilp / maxmem / iterminreg: -mllvm --amdgpu-sched-strategy=max-ilp / max-memory-clause / iterative-minreg
#   unroll1500: -mllvm -unroll-threshold=1500
mkdir -p gpurun_out/r05
L=gpurun_out/r05/ab_builds_flags2.log
: > $L
WORKLOAD=c5 SQRTSPP=8 EMISSIONS=1e7 tools/ab_builds.sh base nobreak maxilp maxmem iterminreg unroll1500 base nobreak 2>&1 | sed "s/^/c5 /" | tee -a $L
WORKLOAD=c3 SQRTSPP=8 tools/ab_builds.sh base maxilp maxmem iterminreg base 2>&1 | sed "s/^/c3 /" | tee -a $L
WORKLOAD=c2 STEPS=3 SQRTSPP=16 tools/ab_builds.sh base maxilp maxmem iterminreg base 2>&1 | sed "s/^/c2 /" | tee -a $L
WORKLOAD=pm STEPS=3 tools/ab_builds.sh base nobreak base nobreak 2>&1 | sed "s/^/pm /" | tee -a $L
