#!/bin/bash
# Round-end profile pass on the GPU box: rocprofv3 kernel-trace/--stats and separate PMC passes (HBM traffic, SQ) of
# the bench workloads. Output under gpurun_out/<tag>/; summarise with tools/summarize_rocprof.py and commit under profiles/.
TAG=${1:-r01c}
R=/root/repo
O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp; export TMPDIR=/tmp
prof() {  # name, rocprof flags..., -- , bench args
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
prof kt_c3 --kernel-trace --stats -- --workload c3 --steps 1 --warmup 0
prof pmc_c3_fetch --kernel-trace --pmc FETCH_SIZE -- --workload c3 --steps 1 --warmup 0
prof pmc_c3_write --kernel-trace --pmc WRITE_SIZE -- --workload c3 --steps 1 --warmup 0
prof pmc_c3_sq --kernel-trace --pmc $SQ -- --workload c3 --steps 1 --warmup 0
prof kt_pm --kernel-trace --stats -- --workload pm --steps 3 --warmup 1
prof pmc_pm_fetch --kernel-trace --pmc FETCH_SIZE -- --workload pm --steps 1 --warmup 0
prof pmc_pm_write --kernel-trace --pmc WRITE_SIZE -- --workload pm --steps 1 --warmup 0
prof pmc_pm_sq --kernel-trace --pmc $SQ -- --workload pm --steps 1 --warmup 0
# keep the summaries only: the rocpd databases are tens of MB each and gpurun_out/ is capped at 64 MiB
python $R/tools/summarize_rocprof.py $O > $O/summary.md
find $O -name "*.db" -delete
find $O -type d -empty -delete
