#!/bin/bash
# Per-kernel time and SQ / TCC counters of ONE frame of a bench workload under given environment settings:
#   tools/pmc_probe.sh <tag> <workload> <sqrtspp> [ENV=VAL ...]
# Output: gpurun_out/pmc_probe/<tag>.md (kernel trace + counter sums per kernel, tools/summarize_rocprof.py).
TAG=$1; WL=$2; SPP=$3; shift 3
R=$PWD; O=$R/gpurun_out/pmc_probe/$TAG; mkdir -p $O
for kv in "$@"; do export "$kv"; done
cd /tmp; export TMPDIR=/tmp
SQ="SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_INSTS_SALU"
SQ2="SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM SQ_WAVES"
timeout 600 rocprofv3 --kernel-trace --stats -d $O/kt -- python $R/bench.py --child-frame --workload $WL --sqrtspp $SPP > $O/kt.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc $SQ -d $O/sq -- python $R/bench.py --child-frame --workload $WL --sqrtspp $SPP > $O/sq.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc $SQ2 -d $O/sq2 -- python $R/bench.py --child-frame --workload $WL --sqrtspp $SPP > $O/sq2.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_128B_sum TCC_HIT_sum TCC_MISS_sum -d $O/tcc -- python $R/bench.py --child-frame --workload $WL --sqrtspp $SPP > $O/tcc.log 2>&1
cd $R
python tools/summarize_rocprof.py $O > $R/gpurun_out/pmc_probe/$TAG.md 2>&1
grep child_frame $O/kt.log | tail -1 >> $R/gpurun_out/pmc_probe/$TAG.md
rm -rf $O
