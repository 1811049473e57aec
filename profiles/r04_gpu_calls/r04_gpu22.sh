#!/bin/bash
# trace kernel: workgroups without rays leave before staging. Pipeline tests, then small / mid / large frames against the megakernel
O=gpurun_out/r04x; mkdir -p $O
timeout 300 python -m pytest tests/test_gpu_parity.py tests/test_deep_tree.py tests/test_gpu_large_scene.py -m gpu -q -x -k "wavefront or pipeline or optional or intersect or deep or tree or c3 or spaceship" 2>&1 | tail -2
for s in 1 2 3 4 8; do
  timeout 200 python tools/ab_probe.py spaceship --sqrtspp $s --steps 3 "wf:MCRT_KERNEL=wf" "sm:MCRT_KERNEL=sm" 2>&1 | grep '^{' | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('spaceship sqrtspp $s', d['variant'], d['ms_best'], d['Mray_s'], d['same_bits_as_first'])" | tee -a $O/early_exit.log
done
for s in 1 2 8; do
  timeout 200 python tools/ab_probe.py c3 --sqrtspp $s --steps 2 "wf:" "sm:MCRT_KERNEL=sm" 2>&1 | grep '^{' | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('c3 sqrtspp $s', d['variant'], d['ms_best'], d['Mray_s'], d['same_bits_as_first'])" | tee -a $O/early_exit.log
done
