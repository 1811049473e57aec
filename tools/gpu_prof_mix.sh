#!/bin/bash
# Instruction mix of one frame of a workload (rocprofv3 PMC passes, SQ counters only). Usage: gpu_prof_mix.sh <workload> <tag> [extra bench args]
WL=${1:-pm}; TAG=${2:-r02_mix_$WL}; shift; shift
O=$PWD/gpurun_out/$TAG
mkdir -p $O
export TMPDIR=/tmp
R=$PWD
cd /tmp
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES -d $O/passA -- python $R/bench.py --child-frame --workload $WL "$@" > $O/passA.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_THREAD_CYCLES_VALU GRBM_GUI_ACTIVE -d $O/passB -- python $R/bench.py --child-frame --workload $WL "$@" > $O/passB.log 2>&1
cd $R
python tools/summarize_rocprof.py $O > $O/summary.md 2>&1
find $O -name "*.db" -size +20M -delete
cat $O/summary.md | head -80
tail -2 $O/passA.log
