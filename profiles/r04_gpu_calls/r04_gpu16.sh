#!/bin/bash
# the slot-count rule on small / medium / full frames (default options), against the megakernel and the old rule
O=gpurun_out/r04p; mkdir -p $O
run() { # workload sqrtspp steps
  timeout 600 python tools/ab_probe.py $1 --sqrtspp $2 --steps $3 "auto:" "wf:MCRT_KERNEL=wf" "wf64:MCRT_KERNEL=wf,MCRT_WF_SLOT_PATHS=64" 2>&1 | grep '^{' | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('$1 sqrtspp $2', d['variant'], d['ms_best'], d['Mray_s'], d['kernel_id'], d['same_bits_as_first'])" | tee -a $O/slot_rule.log
}
run spaceship 2 3; run spaceship 4 3; run spaceship 8 3; run c3 2 3; run c3 4 3; run c3 8 3; run c3 16 2; run c4 8 2
timeout 600 python tools/ab_probe.py c3 --steps 1 "auto:" "wf128:MCRT_WF_SLOT_PATHS=128" 2>&1 | grep '^{' | cut -c1-120 | tee -a $O/slot_rule.log
