#!/bin/bash
# large-k tests again (2 LDS refraction entries), the path-count rule for the pipeline, C3 small frames through both kernels
O=gpurun_out/r04l; mkdir -p $O
timeout 900 python -m pytest tests/test_knn_large_k.py tests/test_gpu_large_scene.py -m gpu -q -rA -k "large_k or spaceship" > $O/pytest.log 2>&1; tail -3 $O/pytest.log; grep -E "max rel|FAILED|Error|pipeline" $O/pytest.log | head -20
timeout 300 python tools/ab_probe.py spaceship --steps 3 "auto:" "sm:MCRT_KERNEL=sm" 2>&1 | grep '^{' | cut -c1-150 | tee $O/spaceship_auto.log
for s in 1 2 4; do
  timeout 300 python tools/ab_probe.py c3 --sqrtspp $s --steps 3 "wf:" "sm:MCRT_KERNEL=sm" 2>&1 | grep '^{' | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('c3 sqrtspp $s', d['variant'], d['ms_best'], d['Mray_s'], d['kernel_id'], d['same_bits_as_first'])" | tee -a $O/c3_sizes.log
done
