#!/bin/bash
# round 6, call 2: (a) C2 / C2-GGX with the flat megakernel's bounce cut at the shadow ray (pathTracerBounceSplit: the Interaction dead
# before the second intersection) against the uncut form, library builds swapped on one box; (b) where a C5 frame's time goes when it
# runs through the wavefront pipeline (MCRT_KERNEL=wf: shade + trace + kNN launches) instead of renderKernelPM: kernel trace.
mkdir -p gpurun_out/r06
L=gpurun_out/r06/ab_c2_split.log
: > $L
WORKLOAD=c2 STEPS=3 tools/ab_builds.sh nosplit split nosplit split 2>&1 | sed "s/^/c2 /" | tee -a $L
WORKLOAD=c2_ggx STEPS=3 tools/ab_builds.sh nosplit split nosplit split 2>&1 | sed "s/^/c2_ggx /" | tee -a $L
R=$PWD; O=$R/gpurun_out/r06/c5_wf; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
for k in default wf; do
  if [ $k = wf ]; then export MCRT_KERNEL=wf; else unset MCRT_KERNEL; fi
  timeout 900 rocprofv3 --kernel-trace --stats -d $O/$k/kt -- python $R/bench.py --child-frame --workload c5 --sqrtspp 8 --emissions 1e7 > $O/$k.kt.log 2>&1
  python $R/tools/summarize_rocprof.py $O/$k > $O/$k.md 2>&1
  grep -h child_frame $O/$k.kt.log | tail -1 >> $O/$k.md
  rm -rf $O/$k
  head -30 $O/$k.md | cut -c1-200
done
