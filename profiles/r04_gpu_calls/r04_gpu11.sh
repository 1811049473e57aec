#!/bin/bash
# k > 128 on the wave search (tests), the pipeline against the megakernels on the mid-size tree at small and large frames, C5 through the pipeline
O=gpurun_out/r04k; mkdir -p $O
timeout 900 python -m pytest tests/test_knn_large_k.py tests/test_gpu_parity.py -m gpu -q -rA -k "knn or pm or photon" > $O/pytest_knn.log 2>&1; tail -3 $O/pytest_knn.log; grep -E "max rel|FAILED|Error" $O/pytest_knn.log | head -20
for s in 1 2 4; do
  timeout 300 python tools/ab_probe.py spaceship --sqrtspp $s --steps 3 "sm:" "wf:MCRT_KERNEL=wf" 2>&1 | grep '^{' | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('spaceship sqrtspp $s', d['variant'], d['ms_best'], d['Mray_s'], d['same_bits_as_first'])" | tee -a $O/spaceship_sizes.log
done
timeout 600 python tools/ab_probe.py c5 --sqrtspp 8 --emissions 1e7 --steps 2 "pm:" "wf:MCRT_KERNEL=wf" 2>&1 | grep '^{' | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('c5 64spp', d.get('variant'), d.get('ms_best'), d.get('Mray_s'), d.get('kernel_id'), d.get('error'))" | tee $O/c5_wf.log
