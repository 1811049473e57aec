#!/bin/bash
# round 6, call 1: the review's first probe, no new code. C3 (64 spp) and C4 (16 spp) with the pool capped at 256 K / 512 K / 1 M / 4 M
# slots against the default rule: per-kernel time (kernel trace) and fabric bytes (TCC read classes + WRITE_SIZE) per variant,
# so that shade ms per slot-bounce and bytes per ray can be read against the 16 M-class pool.
R=$PWD; O=$R/gpurun_out/r06/pool_probe; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
run() {  # tag workload sqrtspp slots
  local tag=$1 wl=$2 spp=$3 slots=$4
  if [ "$slots" != "default" ]; then export MCRT_WF_SLOTS=$slots; else unset MCRT_WF_SLOTS; fi
  timeout 500 rocprofv3 --kernel-trace --stats -d $O/$tag/kt -- python $R/bench.py --child-frame --workload $wl --sqrtspp $spp > $O/$tag.kt.log 2>&1
  timeout 500 rocprofv3 --kernel-trace --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum SQ_WAVE_CYCLES SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY -d $O/$tag/rd -- python $R/bench.py --child-frame --workload $wl --sqrtspp $spp > $O/$tag.rd.log 2>&1
  timeout 500 rocprofv3 --kernel-trace --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum -d $O/$tag/wr -- python $R/bench.py --child-frame --workload $wl --sqrtspp $spp > $O/$tag.wr.log 2>&1
  python $R/tools/summarize_rocprof.py $O/$tag > $O/$tag.md 2>&1
  grep -h child_frame $O/$tag.kt.log | tail -1 >> $O/$tag.md
  # unprofiled timing of the same variant (two frames)
  python $R/tools/ab_probe.py $wl --sqrtspp $spp --steps 2 "$tag:" 2>&1 | tail -1 >> $O/$tag.md
  rm -rf $O/$tag
  tail -3 $O/$tag.md | cut -c1-300
}
for s in default 4194304 1048576 524288 262144; do run c3_$s c3 8 $s; done
for s in default 1048576 262144; do run c4_$s c4 4 $s; done
