#!/bin/bash
# round 5, call 3: the whole GPU tier again (bench.py's line now leaves through a private copy of stdout: RCCL's banner, flushed from C
# stdio at exit, followed the line in call 2), then A/B: packed multiply-adds in the lean visit (MCRT_WF_PK=1), fewer stack rows in LDS
# for more tree blocks (MCRT_TRACE_STACK=12 / 10)
mkdir -p gpurun_out/r05
date
timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -15 | tee gpurun_out/r05/pytest_call3.log
for spec in "c3 8" "c4 4"; do
  set -- $spec
  timeout 400 python tools/ab_probe.py $1 --sqrtspp $2 --steps 2 "lean:" "pk:MCRT_WF_PK=1" "stack12:MCRT_TRACE_STACK=12" "stack10:MCRT_TRACE_STACK=10" "pk_stack12:MCRT_WF_PK=1,MCRT_TRACE_STACK=12" "lean:" "pk:MCRT_WF_PK=1" "stack12:MCRT_TRACE_STACK=12" "pk_stack12:MCRT_WF_PK=1,MCRT_TRACE_STACK=12" 2>&1 | grep '^{' | cut -c1-200 | sed "s/^/$1 /" | tee -a gpurun_out/r05/ab_trace_pk_stack.log
done
date
