#!/bin/bash
# path samples per pool slot of the pipeline (iterations of a small frame = samples per slot x bounces)
O=gpurun_out/r04o; mkdir -p $O
run() { # workload sqrtspp
  timeout 300 python tools/ab_probe.py $1 --sqrtspp $2 --steps 3 "p64:MCRT_KERNEL=wf" "p32:MCRT_KERNEL=wf,MCRT_WF_SLOT_PATHS=32" "p16:MCRT_KERNEL=wf,MCRT_WF_SLOT_PATHS=16" "p8:MCRT_KERNEL=wf,MCRT_WF_SLOT_PATHS=8" "p4:MCRT_KERNEL=wf,MCRT_WF_SLOT_PATHS=4" "sm:MCRT_KERNEL=sm" 2>&1 | grep '^{' | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('$1 sqrtspp $2', d['variant'], d['ms_best'], d['Mray_s'], d['kernel_id'], d['same_bits_as_first'])" | tee -a $O/slot_paths.log
}
run spaceship 1; run spaceship 2; run spaceship 4; run spaceship 8; run c3 2; run c3 4; run c3 8
