#!/bin/bash
# C2-only profile pass (headline kernel): kernel trace + separate FETCH/WRITE/SQ counter passes; summaries only.
TAG=${1:-r01d}
R=/root/repo
O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp; export TMPDIR=/tmp
prof() {
    local name=$1; shift
    local flags=(); while [ "$1" != "--" ]; do flags+=("$1"); shift; done; shift
    timeout 300 rocprofv3 "${flags[@]}" -d $O/$name -- python $R/bench.py "$@" --no-cpu > $O/$name.json 2> $O/$name.err
    tail -c 300 $O/$name.json | head -c 200; echo
}
SQ="SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_ANY SQ_INSTS_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY"
prof kt_c2 --kernel-trace --stats -- --workload c2 --steps 3 --warmup 1
prof pmc_c2_fetch --kernel-trace --pmc FETCH_SIZE -- --workload c2 --steps 1 --warmup 0
prof pmc_c2_write --kernel-trace --pmc WRITE_SIZE -- --workload c2 --steps 1 --warmup 0
prof pmc_c2_sq --kernel-trace --pmc $SQ -- --workload c2 --steps 1 --warmup 0
python $R/tools/summarize_rocprof.py $O > $O/summary.md
find $O -name "*.db" -delete
find $O -type d -empty -delete
